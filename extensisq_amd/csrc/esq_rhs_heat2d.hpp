// esq_rhs_heat2d.hpp -- the pointwise functor of the built-in 2-D heat plugin, shared by
// its translation units (esq_rhs_heat2d.hip: the sweeps and the Chebyshev chains;
// esq_rhs_heat2d_chain*.hip: the chain sweeps of the explicit pairs, one unit per depth
// range, so that their template instantiations compile in parallel).
#pragma once
#include "esq_rhs_common.hpp"

using namespace esq_rhs;

namespace {

// the plugin's pointwise functor: f = c * laplacian (everything else:
// esq_stencil2d.hpp)
struct HeatFn {
    double c;
    __device__ __forceinline__ void eval(const double2 (&)[1], const double2 (&lap)[1],
                                         double2 (&f)[1]) const {
        f[0].x = c * lap[0].x;
        f[0].y = c * lap[0].y;
    }
};

using Heat = esq::Stencil2D<1, false, HeatFn>;
inline HeatFn fn_of(const Rhs *r) { return HeatFn{(double)(r->N + 1) * (double)(r->N + 1)}; }


// the chain sweeps of depth LO..HI (a depth outside: ESQ_ENOTSUP).  The one-field rows
// are light: tall tiles where one wave per SIMD fills the chip, tiles down to `depth`
// rows (esq_stencil2d.hpp, geo_chain)
template <int LO, int HI>
inline int heat2d_chain_range(Rhs *r, const double *y_in, const esq_chain *chain,
                              void *stream, void *start_event, void *stop_event) {
    return Heat::chain<LO, HI>(fn_of(r), r->N, y_in, chain, stream, start_event, stop_event,
                               /*tall_tiles=*/true, /*min_rows=*/-1, &r->tune);
}

}  // namespace

namespace esq_rhs {
// one per translation unit (esq_rhs_heat2d_chain23.hip, ..4.hip, ..56.hip)
int heat2d_chain_d23(Rhs *, const double *, const esq_chain *, void *, void *, void *);
int heat2d_chain_d4(Rhs *, const double *, const esq_chain *, void *, void *, void *);
int heat2d_chain_d56(Rhs *, const double *, const esq_chain *, void *, void *, void *);
}  // namespace esq_rhs
