// esq_rhs_bruss2d.hip -- 2-D Brusselator reaction-diffusion, periodic
// (BASELINE.json configs[2], the north-star workload).
#include "esq_rhs_bruss2d.hpp"

namespace {

// 2-D Brusselator, periodic.  y = [u.ravel(), v.ravel()]
//   du = (A + u*u*v - (B+1)*u) + d*lap(u);  dv = (B*u - u*u*v) + d*lap(v)
template <bool NTS>
__global__ __launch_bounds__(kBlock) void k_bruss2d(
    const double *__restrict__ y, double *__restrict__ f, int N, double d,
    double A, double B, unsigned nblocks, unsigned bpr) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const unsigned i = lb / bpr;
    const unsigned j = (lb % bpr) * kBlock + threadIdx.x;
    if (i >= (unsigned)N || j >= (unsigned)N) return;
    const size_t NN = (size_t)N * N;
    const double *__restrict__ u = y;
    const double *__restrict__ v = y + NN;
    const unsigned im = i == 0 ? N - 1 : i - 1, ip = i + 1 == (unsigned)N ? 0 : i + 1;
    const unsigned jm = j == 0 ? N - 1 : j - 1, jp = j + 1 == (unsigned)N ? 0 : j + 1;
    const size_t k = (size_t)i * N + j;
    const size_t kup = (size_t)im * N + j, kdn = (size_t)ip * N + j;
    const size_t klf = (size_t)i * N + jm, krt = (size_t)i * N + jp;
    const double uc = u[k], vc = v[k];
    const double lapu = ((u[kup] + u[kdn]) + (u[klf] + u[krt])) - 4.0 * uc;
    const double lapv = ((v[kup] + v[kdn]) + (v[klf] + v[krt])) - 4.0 * vc;
    const double uuv = uc * uc * vc;
    const double fu = ((A + uuv) - (B + 1.0) * uc) + d * lapu;
    const double fv = (B * uc - uuv) + d * lapv;
    if (NTS) {
        __builtin_nontemporal_store(fu, f + k);
        __builtin_nontemporal_store(fv, f + NN + k);
    } else {
        f[k] = fu;
        f[NN + k] = fv;
    }
}

}  // namespace

extern "C" {

int esq_rhs_bruss2d_create(void **user_out, int N, double alpha, double a,
                           double b) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = BRUSS2D; r.N = N; r.n = 2 * (size_t)N * N;
    r.alpha = alpha; r.a = a; r.b = b;
    return make(user_out, r);
}

int esq_rhs_bruss2d(void *user, double t, const double *y, double *f, size_t n,
                    void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n) return ESQ_EINVAL;
    const double d = r->alpha * ((double)r->N * (double)r->N);
    if (BrussSplit::grid_ok(r->N))
        return BrussSplit::rhs(fn_of(r), r->N, y, f, stream);
    // odd grids: the scalar kernel
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    hipLaunchKernelGGL(k_bruss2d<false>, dim3(grid), dim3(kBlock), 0,
                       (hipStream_t)stream, y, f, r->N, d, r->a, r->b, grid, bpr);
    return (int)hipGetLastError();
}

int esq_rhs_bruss2d_fused(void *user, double t, const double *y_in, double *f,
                          const esq_epilogue *epi, size_t n, void *stream,
                          void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n || !epi) return ESQ_EINVAL;
    return BrussSplit::fused(fn_of(r), r->N, y_in, f, epi, stream, start_event, stop_event);
}

int esq_rhs_bruss2d_chain(void *user, const double *y_in, const esq_chain *chain,
                          size_t n, void *stream, void *start_event,
                          void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n || !chain) return ESQ_EINVAL;
    // (the depths live in three translation units: esq_rhs_bruss2d.hpp)
    if (chain->depth <= 3)
        return bruss2d_chain_d23(r, y_in, chain, stream, start_event, stop_event);
    if (chain->depth == 4)
        return bruss2d_chain_d4(r, y_in, chain, stream, start_event, stop_event);
    return bruss2d_chain_d56(r, y_in, chain, stream, start_event, stop_event);
}

}  // extern "C"
