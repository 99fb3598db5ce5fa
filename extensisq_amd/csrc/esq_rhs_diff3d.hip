// esq_rhs_diff3d.hip -- 3-D diffusion, 7-point Laplacian, Dirichlet 0
// (BASELINE.json configs[3], the SSV2stab workload): the first instantiation of
// esq_stencil3d.hpp.  Every sweep, epilogue, chain and tile geometry lives there;
// this unit is the pointwise functor and the C entry points.
#include "esq_rhs_common.hpp"
#include "esq_stencil3d.hpp"

using namespace esq_rhs;

namespace {

// u_t = (N + 1)^2 * (sum of the six neighbours - 6 u): the expression of the NumPy twin
// (oracle/problems.py: diff3d_rhs), operation for operation
struct Diff3dFn {
    static constexpr bool kZeroOutside = true, kAutonomous = true;
    double c;
    __device__ __forceinline__ double ghost(int, int, int, int, int, double, double) const {
        return 0.0;
    }
    __device__ __forceinline__ void eval(const esq::Nb3 (&nb)[1], int, int, int, double,
                                         double (&f)[1]) const {
        f[0] = c * ((((nb[0].below + nb[0].above) + (nb[0].up + nb[0].dn)) +
                     (nb[0].lf + nb[0].rt)) - 6.0 * nb[0].c);
    }
};
using Diff3d = esq::Stencil3D<1, Diff3dFn>;

inline Diff3dFn fn_of(const Rhs *r) {
    return Diff3dFn{(double)(r->N + 1) * (double)(r->N + 1)};
}
inline esq::Stencil3dTuning tuning_of(const Rhs *r) {
    esq::Stencil3dTuning t;
    t.force = r->rkc_force; t.planes = r->rkc_planes; t.jt = r->rkc_jt; t.nw = r->rkc_nw;
    t.march_r = r->diff3d_r;
    return t;
}

}  // namespace

extern "C" {

// D Chebyshev stages per launch (esq_rhs_rkc_chain_fn)
int esq_rhs_diff3d_rkc_chain(void *user, const esq_rkc_chain *ch, size_t n, void *stream,
                             void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !ch) return ESQ_EINVAL;
    return Diff3d::rkc_chain(fn_of(r), r->N, ch, stream, start_event, stop_event, tuning_of(r));
}

// D consecutive Runge-Kutta stages per launch (esq_rhs_chain_fn, esq_chain3d.hpp)
int esq_rhs_diff3d_chain(void *user, const double *y_in, const esq_chain *chain, size_t n,
                         void *stream, void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !chain) return ESQ_EINVAL;
    return Diff3d::chain(fn_of(r), r->N, y_in, chain, stream, start_event, stop_event,
                         tuning_of(r));
}

int esq_rhs_diff3d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = DIFF3D; r.N = N; r.n = (size_t)N * N * N;
    const esq::Stencil3dTuning t = esq::stencil3d_tuning_from_env();
    r.rkc_force = t.force; r.rkc_planes = t.planes; r.rkc_jt = t.jt; r.rkc_nw = t.nw;
    r.diff3d_r = t.march_r;
    return make(user_out, r);
}

int esq_rhs_diff3d_rkc(void *user, double t, const double *yjm1, const double *yjm2,
                       const double *yn, const double *fn, double mu, double nu,
                       double omn, double hmus, double ajm1, double *y_out,
                       size_t n, void *stream, void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    return Diff3d::rkc(fn_of(r), r->N, t, yjm1, yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out,
                       stream, start_event, stop_event, tuning_of(r));
}
// fused entry: the Runge-Kutta arithmetic that follows a stage evaluation (and
// the end of a Chebyshev step) inside the sweep, every epilogue kind; the
// on-the-fly first-stage input (ESQ_FUSE_SRC) is not offered
int esq_rhs_diff3d_fused(void *user, double t, const double *y_in, double *f,
                         const esq_epilogue *epi, size_t n, void *stream,
                         void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !epi) return ESQ_EINVAL;
    return Diff3d::fused(fn_of(r), r->N, t, y_in, f, epi, stream, start_event, stop_event,
                         tuning_of(r));
}
int esq_rhs_diff3d(void *user, double t, const double *y, double *f, size_t n,
                   void *stream) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    return Diff3d::rhs(fn_of(r), r->N, t, y, f, stream, tuning_of(r), rhs_variant() == 1);
}

}  // extern "C"
