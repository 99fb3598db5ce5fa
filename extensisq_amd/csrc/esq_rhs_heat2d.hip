// esq_rhs_heat2d.hip -- 2-D heat equation, 5-point Laplacian, Dirichlet 0
// (BASELINE.json configs[1], configs[4]).
#include "esq_rhs_heat2d.hpp"

namespace {

// 2-D heat, Dirichlet 0:  c*((up + down) + (left + right) - 4*u)
__global__ __launch_bounds__(kBlock) void k_heat2d(const double *__restrict__ u,
                                                   double *__restrict__ f, int N,
                                                   double c, unsigned nblocks,
                                                   unsigned bpr) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const unsigned i = lb / bpr;
    const unsigned j = (lb % bpr) * kBlock + threadIdx.x;
    if (i >= (unsigned)N || j >= (unsigned)N) return;
    const size_t k = (size_t)i * N + j;
    const double uc = u[k];
    const double up = i > 0 ? u[k - N] : 0.0;
    const double dn = i + 1 < (unsigned)N ? u[k + N] : 0.0;
    const double lf = j > 0 ? u[k - 1] : 0.0;
    const double rt = j + 1 < (unsigned)N ? u[k + 1] : 0.0;
    f[k] = c * (((up + dn) + (lf + rt)) - 4.0 * uc);
}

}  // namespace

extern "C" {

int esq_rhs_heat2d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = HEAT2D; r.N = N; r.n = (size_t)N * N;
    return make(user_out, r);
}

int esq_rhs_heat2d(void *user, double t, const double *y, double *f, size_t n,
                   void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n) return ESQ_EINVAL;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    if (Heat::grid_ok(r->N))
        return Heat::rhs(fn_of(r), r->N, y, f, stream);
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    hipLaunchKernelGGL(k_heat2d, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       y, f, r->N, c, grid, bpr);
    return (int)hipGetLastError();
}
int esq_rhs_heat2d_fused(void *user, double t, const double *y_in, double *f,
                         const esq_epilogue *epi, size_t n, void *stream,
                         void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n || !epi) return ESQ_EINVAL;
    return Heat::fused(fn_of(r), r->N, y_in, f, epi, stream, start_event, stop_event);
}
int esq_rhs_heat2d_rkc(void *user, double t, const double *yjm1, const double *yjm2,
                       const double *yn, const double *fn, double mu, double nu,
                       double omn, double hmus, double ajm1, double *y_out,
                       size_t n, void *stream, void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n) return ESQ_EINVAL;
    return Heat::rkc(fn_of(r), r->N, yjm1, make_epi(yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out),
                     stream, start_event, stop_event);
}

// D Chebyshev stages per launch (esq_rhs_rkc_chain_fn, csrc/esq_rkc2d.hpp)
int esq_rhs_heat2d_rkc_chain(void *user, const esq_rkc_chain *ch, size_t n, void *stream,
                             void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n || !ch) return ESQ_EINVAL;
    return Heat::rkc_chain(fn_of(r), r->N, ch, stream, start_event, stop_event, &r->tune);
}

// (the depths live in three translation units: esq_rhs_heat2d.hpp)
int esq_rhs_heat2d_chain(void *user, const double *y_in, const esq_chain *chain,
                          size_t n, void *stream, void *start_event,
                          void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n || !chain) return ESQ_EINVAL;
    if (chain->depth <= 3)
        return heat2d_chain_d23(r, y_in, chain, stream, start_event, stop_event);
    if (chain->depth == 4)
        return heat2d_chain_d4(r, y_in, chain, stream, start_event, stop_event);
    return heat2d_chain_d56(r, y_in, chain, stream, start_event, stop_event);
}

}  // extern "C"
