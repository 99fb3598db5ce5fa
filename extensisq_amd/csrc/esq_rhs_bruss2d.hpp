// esq_rhs_bruss2d.hpp -- the pointwise functor of the built-in 2-D Brusselator plugin,
// shared by its translation units (esq_rhs_bruss2d.hip: the sweeps; esq_rhs_bruss2d_chain*.hip:
// the chain sweeps, one unit per depth range, so that their template instantiations --
// the longest compile of the library -- run in parallel).
#pragma once
#include "esq_rhs_common.hpp"

using namespace esq_rhs;

namespace {

// The plugin's pointwise functor: centres and five-point Laplacians of (u, v) ->
// (du, dv), same operation order as k_bruss2d.  Everything else -- the one-stage
// sweep with its epilogues, the marching chain sweeps, the launch geometry -- is
// esq_stencil2d.hpp, shared with the heat plugin and with user plugins.
struct BrussFn {
    double d, A, B;
    __device__ __forceinline__ void eval(const double2 (&c)[2], const double2 (&lap)[2],
                                         double2 (&f)[2]) const {
        const double uuvx = c[0].x * c[0].x * c[1].x, uuvy = c[0].y * c[0].y * c[1].y;
        f[0].x = ((A + uuvx) - (B + 1.0) * c[0].x) + d * lap[0].x;
        f[0].y = ((A + uuvy) - (B + 1.0) * c[0].y) + d * lap[0].y;
        f[1].x = (B * c[0].x - uuvx) + d * lap[1].x;
        f[1].y = (B * c[0].y - uuvy) + d * lap[1].y;
    }
    // one field only (split chain sweeps: a wave per field), same operations
    __device__ __forceinline__ double2 eval_one(int field, const double2 (&c)[2],
                                                double2 lap) const {
        const double uuvx = c[0].x * c[0].x * c[1].x, uuvy = c[0].y * c[0].y * c[1].y;
        if (field == 0)
            return make_double2(((A + uuvx) - (B + 1.0) * c[0].x) + d * lap.x,
                                ((A + uuvy) - (B + 1.0) * c[0].y) + d * lap.y);
        return make_double2((B * c[0].x - uuvx) + d * lap.x, (B * c[0].y - uuvy) + d * lap.y);
    }
};

// one field per wave in the chain sweeps (both fields in one wave -- round 3's first
// version, narrower register caps, Pr8 0.93 against 0.69 ms/step -- was retired in
// round 6 together with its switch: half of this plugin's compile time)
using BrussSplit = esq::Stencil2D<2, true, BrussFn, true>;
inline BrussFn fn_of(const Rhs *r) {
    return BrussFn{r->alpha * ((double)r->N * (double)r->N), r->a, r->b};
}


// the chain sweeps of depth LO..HI (a depth outside: ESQ_ENOTSUP)
template <int LO, int HI>
inline int bruss2d_chain_range(Rhs *r, const double *y_in, const esq_chain *chain,
                               void *stream, void *start_event, void *stop_event) {
    return BrussSplit::chain<LO, HI>(fn_of(r), r->N, y_in, chain, stream, start_event,
                                     stop_event, /*tall_tiles=*/false, /*min_rows=*/0,
                                     &r->tune);
}

}  // namespace

namespace esq_rhs {
// one per translation unit (esq_rhs_bruss2d_chain23.hip, ..4.hip, ..56.hip)
int bruss2d_chain_d23(Rhs *, const double *, const esq_chain *, void *, void *, void *);
int bruss2d_chain_d4(Rhs *, const double *, const esq_chain *, void *, void *, void *);
int bruss2d_chain_d56(Rhs *, const double *, const esq_chain *, void *, void *, void *);
}  // namespace esq_rhs
