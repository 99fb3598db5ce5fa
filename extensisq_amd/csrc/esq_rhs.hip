// esq_rhs.hip -- built-in device RHS plugins for the synthetic workloads named
// in BASELINE.json `configs`.  They stand where the user's
// Python callable `fun(t, y)` stands in the reference (common.py:356); their
// NumPy twins, used by the tests, are in oracle/problems.py and use the same
// operation order (the library is built with -ffp-contract=off), so the two
// agree bit for bit.
//
// All kernels are stencil sweeps: one HBM read + one HBM write per element is
// the floor ("RHS-min" in BASELINE.md); neighbour reuse is served by L1/L2.
// Workgroups are dealt round-robin over the 8 XCDs, so block b is remapped to
// a contiguous band of rows per XCD (blockIdx % 8 = XCD label): the up/down
// neighbour rows then hit the SAME XCD's L2 instead of being fetched twice.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/extensisq_amd.h"

namespace {

constexpr int kBlock = 256;
constexpr int kXcd = 8;

enum Kind { DIAG = 1, HEAT2D = 2, BRUSS2D = 3, DIFF3D = 4 };

struct Rhs {
    int kind;
    int N;
    int device;
    double alpha, a, b;
    double amp;
    double *lam_dev;
    size_t n;
};

// band remap: logical block id such that XCD x (label blockIdx%8) sweeps the
// contiguous range [x*per, (x+1)*per)
__device__ __forceinline__ unsigned band_block(unsigned b, unsigned nblocks) {
    const unsigned per = (nblocks + kXcd - 1) / kXcd;
    return (b % kXcd) * per + b / kXcd;
}

// f = lam*y + amp*sin(t)
__global__ __launch_bounds__(kBlock) void k_diag(const double *__restrict__ y,
                                                 double *__restrict__ f,
                                                 const double *__restrict__ lam,
                                                 double forcing, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        f[i] = lam[i] * y[i] + forcing;
}

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

// 2-D Brusselator, periodic.  y = [u.ravel(), v.ravel()]
//   du = (A + u*u*v - (B+1)*u) + d*lap(u);  dv = (B*u - u*u*v) + d*lap(v)
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
    f[k] = ((A + uuv) - (B + 1.0) * uc) + d * lapu;
    f[NN + k] = (B * uc - uuv) + d * lapv;
}

// 3-D diffusion, Dirichlet 0, 7-point
__global__ __launch_bounds__(kBlock) void k_diff3d(const double *__restrict__ u,
                                                   double *__restrict__ f, int N,
                                                   double c, unsigned nblocks,
                                                   unsigned bpr) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const unsigned row = lb / bpr;   // row = i*N + j
    const unsigned l = (lb % bpr) * kBlock + threadIdx.x;
    if (row >= (unsigned)N * N || l >= (unsigned)N) return;
    const unsigned i = row / N, j = row % N;
    const size_t NN = (size_t)N * N;
    const size_t k = (size_t)row * N + l;
    const double uc = u[k];
    const double a0 = i > 0 ? u[k - NN] : 0.0;
    const double a1 = i + 1 < (unsigned)N ? u[k + NN] : 0.0;
    const double b0 = j > 0 ? u[k - N] : 0.0;
    const double b1 = j + 1 < (unsigned)N ? u[k + N] : 0.0;
    const double c0 = l > 0 ? u[k - 1] : 0.0;
    const double c1 = l + 1 < (unsigned)N ? u[k + 1] : 0.0;
    f[k] = c * ((((a0 + a1) + (b0 + b1)) + (c0 + c1)) - 6.0 * uc);
}

int make(void **out, Rhs proto) {
    if (!out) return ESQ_EINVAL;
    Rhs *r = (Rhs *)malloc(sizeof(Rhs));
    if (!r) return ESQ_ENOMEM;
    *r = proto;
    *out = r;
    return 0;
}

}  // namespace

extern "C" {

int esq_rhs_diag_create(void **user_out, int device, const double *lam_host,
                        size_t n, double forcing_amp) {
    if (!lam_host || n == 0) return ESQ_EINVAL;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    double *d = nullptr;
    e = hipMalloc(&d, n * sizeof(double));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(d, lam_host, n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return (int)e; }
    Rhs r{};
    r.kind = DIAG; r.n = n; r.lam_dev = d; r.amp = forcing_amp; r.device = device;
    return make(user_out, r);
}
int esq_rhs_heat2d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = HEAT2D; r.N = N; r.n = (size_t)N * N;
    return make(user_out, r);
}
int esq_rhs_bruss2d_create(void **user_out, int N, double alpha, double a,
                           double b) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = BRUSS2D; r.N = N; r.n = 2 * (size_t)N * N;
    r.alpha = alpha; r.a = a; r.b = b;
    return make(user_out, r);
}
int esq_rhs_diff3d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = DIFF3D; r.N = N; r.n = (size_t)N * N * N;
    return make(user_out, r);
}
int esq_rhs_free(void *user) {
    if (!user) return 0;
    Rhs *r = (Rhs *)user;
    if (r->kind == DIAG && r->lam_dev) {
        (void)hipSetDevice(r->device);
        (void)hipFree(r->lam_dev);
    }
    free(r);
    return 0;
}

int esq_rhs_diag(void *user, double t, const double *y, double *f, size_t n,
                 void *stream) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIAG || n != r->n) return ESQ_EINVAL;
    const double forcing = r->amp != 0.0 ? r->amp * sin(t) : 0.0;
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_diag, dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, y, f, r->lam_dev, forcing, n);
    return (int)hipGetLastError();
}
int esq_rhs_heat2d(void *user, double t, const double *y, double *f, size_t n,
                   void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n) return ESQ_EINVAL;
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipLaunchKernelGGL(k_heat2d, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       y, f, r->N, c, grid, bpr);
    return (int)hipGetLastError();
}
int esq_rhs_bruss2d(void *user, double t, const double *y, double *f, size_t n,
                    void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n) return ESQ_EINVAL;
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    const double d = r->alpha * ((double)r->N * (double)r->N);
    hipLaunchKernelGGL(k_bruss2d, dim3(grid), dim3(kBlock), 0,
                       (hipStream_t)stream, y, f, r->N, d, r->a, r->b, grid, bpr);
    return (int)hipGetLastError();
}
int esq_rhs_diff3d(void *user, double t, const double *y, double *f, size_t n,
                   void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipLaunchKernelGGL(k_diff3d, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       y, f, r->N, c, grid, bpr);
    return (int)hipGetLastError();
}

}  // extern "C"
