// esq_rhs_diff3d.hip -- 3-D diffusion, 7-point Laplacian, Dirichlet 0
// (BASELINE.json configs[3], the SSV2stab workload): the first instantiation of
// esq_stencil3d.hpp.  Every sweep, epilogue, chain and tile geometry lives there;
// this unit is the pointwise functor and the C entry points.
#include "esq_rhs_diff3d.hpp"


extern "C" {

int esq_rhs_diff3d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = DIFF3D; r.N = N; r.n = (size_t)N * N * N;
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
    return Diff3d::rhs(fn_of(r), r->N, t, y, f, stream, tuning_of(r), false);
}

}  // extern "C"
