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

#include <hip/hip_ext.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_plugin.hpp"
#include "esq_terms.hpp"

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


// ---------------------------------------------------------------------------
// Vectorised 5-point sweeps (even N): one thread owns a column PAIR (16-byte
// accesses) and marches down R rows with a rolling (up, centre, down) register
// window, so every row is loaded once per row group instead of three times;
// the left/right neighbours come from the adjacent lanes (wave64 shuffles),
// only the lanes at a wave or row edge touch memory for them.  Arithmetic
// order is identical to the scalar kernels (and to oracle/problems.py).
// ---------------------------------------------------------------------------
using esq::v2d;
using RkcEpi = esq::EpiRkc;

// Where a sweep takes its input from: a vector in memory (SrcPlain), or -- for
// the FIRST stage of a step -- the stage argument formed on the fly from the
// state and the first stage derivative,
//     ys = y + h * (c * K0)            (common.py:355, stage 1: one term)
// with exactly the operations of k_lincomb / EpiStage (fma(c, K0, 0), then *h,
// then +y, each rounded), so the derivative is bit-identical.  The argument is
// then never written to nor read from memory: the end-point sweep of the
// previous step need not produce it, the first sweep reads y and K0 (which its
// epilogue needs anyway) instead of a third vector.  Unlike the general
// "stage argument inside the stencil sweep" (rejected: 12-term rows on halos)
// this costs one extra row window of ONE vector, served by L1/L2.
struct SrcPlain {
    const double *__restrict__ f;
    __device__ __forceinline__ double2 ld2(size_t e2) const {
        return reinterpret_cast<const double2 *>(f)[e2];
    }
    __device__ __forceinline__ double ld(size_t e) const { return f[e]; }
};
struct SrcAxpy {
    const double *__restrict__ y, *__restrict__ k;
    double c, h;
    __device__ __forceinline__ double one(double yy, double kk) const {
        return __dadd_rn(yy, __dmul_rn(h, fma(c, kk, 0.0)));
    }
    __device__ __forceinline__ double2 ld2(size_t e2) const {
        const double2 a = reinterpret_cast<const double2 *>(y)[e2];
        const double2 b = reinterpret_cast<const double2 *>(k)[e2];
        return make_double2(one(a.x, b.x), one(a.y, b.y));
    }
    __device__ __forceinline__ double ld(size_t e) const { return one(y[e], k[e]); }
};

template <bool PERIODIC, class Src>
struct RowWin {
    Src src;
    size_t base;                    // offset of the field inside the state (doubles)
    int N;
    unsigned pair, npairs;          // this thread's column pair
    bool live;                      // pair < npairs
    __device__ __forceinline__ double2 row(int i) const {
        // row i of the field at this thread's pair; rows outside are the
        // periodic image or zero (Dirichlet)
        if (PERIODIC) {
            i = i < 0 ? i + N : (i >= N ? i - N : i);
        } else if (i < 0 || i >= N) {
            return make_double2(0.0, 0.0);
        }
        if (!live) return make_double2(0.0, 0.0);
        return src.ld2((base + (size_t)i * N) / 2 + pair);       // N even
    }
    // left neighbour of .x and right neighbour of .y in row i (centre c given)
    __device__ __forceinline__ void sides(int i, double2 c, double &lf,
                                          double &rt) const {
        const int lane = threadIdx.x & 63;
        lf = __shfl_up(c.y, 1, 64);
        rt = __shfl_down(c.x, 1, 64);
        if (!live) return;
        const size_t r = base + (size_t)i * N;
        if (lane == 0 || pair == 0) {
            if (pair > 0) lf = src.ld(r + 2 * (size_t)pair - 1);
            else lf = PERIODIC ? src.ld(r + N - 1) : 0.0;
        }
        if (lane == 63 || pair + 1 >= npairs) {
            if (pair + 1 < npairs) rt = src.ld(r + 2 * (size_t)pair + 2);
            else rt = PERIODIC ? src.ld(r) : 0.0;
        }
    }
};

// ---------------------------------------------------------------------------
// SWEEPS.  One wave tile = 64 column pairs of ONE grid row; all three window
// rows are requested up front, together with the epilogue's operands, so every
// load of the thread is in flight before the first use.  `Epi` (esq_epilogue.hpp)
// says what happens to the fresh derivative: store only (EpiNone), next stage
// argument (EpiStage), blocked accumulation (EpiBlock), solution + error norm
// (EpiSolErr), FSAL error norm (EpiErrNorm), Chebyshev recursion (EpiRkc).
// The epilogues are pointwise: nothing is recomputed on halos.
// ---------------------------------------------------------------------------
template <class Epi, class Src>
__global__ __launch_bounds__(kBlock) void k_bruss2d_sweep(
    Src ys, double *__restrict__ f, Epi epi, int N,
    double d, double A, double B, unsigned nblocks, unsigned wpr) {
    const unsigned tile = band_block(blockIdx.x, nblocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int i = (int)(tile / wpr);
    double local = 0.0;
    if (i < N) {                                           // wave-uniform
        const size_t NN = (size_t)N * N;
        RowWin<true, Src> U, V;
        U.src = V.src = ys;
        U.base = 0; V.base = NN;
        U.N = V.N = N;
        U.npairs = V.npairs = (unsigned)N / 2;
        U.pair = V.pair = (tile % wpr) * 64 + (threadIdx.x & 63);
        U.live = V.live = U.pair < U.npairs;
        const size_t k2 = ((size_t)i * N) / 2 + (U.live ? U.pair : 0);   // N even
        const size_t v2 = NN / 2 + k2;
        typename Epi::In cu, cv;
        epi.load(cu, k2);
        epi.load(cv, v2);
        const double2 uu = U.row(i - 1), uc = U.row(i), ud = U.row(i + 1);
        const double2 vu = V.row(i - 1), vc = V.row(i), vd = V.row(i + 1);
        double ul, urt, vl, vrt;
        U.sides(i, uc, ul, urt);
        V.sides(i, vc, vl, vrt);
        double2 fu, fv;
        {
            const double lapx = ((uu.x + ud.x) + (ul + uc.y)) - 4.0 * uc.x;
            const double lapy = ((uu.y + ud.y) + (uc.x + urt)) - 4.0 * uc.y;
            const double lvx = ((vu.x + vd.x) + (vl + vc.y)) - 4.0 * vc.x;
            const double lvy = ((vu.y + vd.y) + (vc.x + vrt)) - 4.0 * vc.y;
            const double uuvx = uc.x * uc.x * vc.x, uuvy = uc.y * uc.y * vc.y;
            fu.x = ((A + uuvx) - (B + 1.0) * uc.x) + d * lapx;
            fu.y = ((A + uuvy) - (B + 1.0) * uc.y) + d * lapy;
            fv.x = (B * uc.x - uuvx) + d * lvx;
            fv.y = (B * uc.y - uuvy) + d * lvy;
        }
        if (U.live) {
            epi.store_f(f, k2, fu);
            epi.store_f(f, v2, fv);
            epi.finish(cu, fu, uc, k2, local);
            epi.finish(cv, fv, vc, v2, local);
        }
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}

template <class Epi, class Src>
__global__ __launch_bounds__(kBlock) void k_heat2d_sweep(
    Src ys, double *__restrict__ f, Epi epi, int N,
    double c, unsigned nblocks, unsigned wpr) {
    const unsigned tile = band_block(blockIdx.x, nblocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int i = (int)(tile / wpr);
    double local = 0.0;
    if (i < N) {
        RowWin<false, Src> U;
        U.src = ys;
        U.base = 0;
        U.N = N;
        U.npairs = (unsigned)N / 2;
        U.pair = (tile % wpr) * 64 + (threadIdx.x & 63);
        U.live = U.pair < U.npairs;
        const size_t k2 = ((size_t)i * N) / 2 + (U.live ? U.pair : 0);
        typename Epi::In cu;
        epi.load(cu, k2);
        const double2 uu = U.row(i - 1), uc = U.row(i), ud = U.row(i + 1);
        double ul, urt;
        U.sides(i, uc, ul, urt);
        double2 out;
        out.x = c * (((uu.x + ud.x) + (ul + uc.y)) - 4.0 * uc.x);
        out.y = c * (((uu.y + ud.y) + (uc.x + urt)) - 4.0 * uc.y);
        if (U.live) {
            epi.store_f(f, k2, out);
            epi.finish(cu, out, uc, k2, local);
        }
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}

// f = lam*y + forcing, pointwise: the same epilogues on a grid-stride loop
// (lam holds exactly n doubles; the state vectors are zero-padded to a multiple
// of 512, and the padding must stay zero)
template <class Epi, class Src>
__global__ __launch_bounds__(kBlock) void k_diag_sweep(
    Src y, double *__restrict__ f, Epi epi,
    const double *__restrict__ lam, double forcing, size_t n, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i2 = (size_t)blockIdx.x * kBlock + threadIdx.x; i2 < n2; i2 += stride) {
        typename Epi::In in;
        epi.load(in, i2);
        const double2 yc = y.ld2(i2);
        double2 fy = make_double2(0.0, 0.0);
        if (2 * i2 < n) fy.x = lam[2 * i2] * yc.x + forcing;
        if (2 * i2 + 1 < n) fy.y = lam[2 * i2 + 1] * yc.y + forcing;
        epi.store_f(f, i2, fy);
        epi.finish(in, fy, yc, i2, local);
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
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

// ESQ_RHS_VARIANT=1: scalar kernels instead of the vectorised sweeps (tests)
int rhs_variant() {
    static const int v = getenv("ESQ_RHS_VARIANT") ? atoi(getenv("ESQ_RHS_VARIANT")) : 0;
    return v;
}


// 3-D diffusion, marching version: a thread owns one (j, l) column of the grid
// (flattened plane index p) and walks R planes along i with a rolling
// (below, centre, above) window; the l-neighbours come from adjacent lanes,
// the j-neighbours are two coalesced loads of the centre plane.  3 loads per
// output instead of 7; arithmetic order identical to k_diff3d.
template <int R, bool RKC>
__global__ __launch_bounds__(kBlock) void k_diff3d_v2(
    const double *__restrict__ u, double *__restrict__ f, int N, double c,
    unsigned nblocks, unsigned bpp, RkcEpi epi) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const int i0 = (int)(lb / bpp) * R;
    const unsigned p = (lb % bpp) * kBlock + threadIdx.x;     // plane index
    const unsigned NN = (unsigned)N * (unsigned)N;
    if (i0 >= N) return;
    const bool live = p < NN;
    const unsigned j = live ? p / N : 0, l = live ? p % N : 0;
    const int lane = threadIdx.x & 63;
    auto at = [&](int i) -> double {
        return (live && i >= 0 && i < N) ? u[(size_t)i * NN + p] : 0.0;
    };
    double below = at(i0 - 1), centre = at(i0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = i0 + r;
        if (i >= N) break;
        const double above = at(i + 1);
        double c0 = __shfl_up(centre, 1, 64), c1 = __shfl_down(centre, 1, 64);
        if (live) {
            const double *pl = u + (size_t)i * NN;
            if (l == 0) c0 = 0.0; else if (lane == 0) c0 = pl[p - 1];
            if (l + 1 == (unsigned)N) c1 = 0.0;
            else if (lane == 63 || p + 1 >= NN) c1 = pl[p + 1];
            const double b0 = j > 0 ? pl[p - N] : 0.0;
            const double b1 = j + 1 < (unsigned)N ? pl[p + N] : 0.0;
            const double fy =
                c * ((((below + above) + (b0 + b1)) + (c0 + c1)) - 6.0 * centre);
            const size_t k = (size_t)i * NN + p;
            if (RKC)
                epi.out[k] = epi.one(centre, epi.yjm2[k], epi.yn[k], epi.fn[k], fy);
            else
                f[k] = fy;
        }
        below = centre;
        centre = above;
    }
}

int make(void **out, Rhs proto) {
    if (!out) return ESQ_EINVAL;
    Rhs *r = (Rhs *)malloc(sizeof(Rhs));
    if (!r) return ESQ_ENOMEM;
    *r = proto;
    *out = r;
    return 0;
}

// the on-the-fly first-stage input is instantiated for the epilogues a first
// stage can have: the second stage's argument with at most one row from memory
template <class E> constexpr bool kFirstStage = false;
template <> constexpr bool kFirstStage<esq::EpiStage<0>> = true;
template <> constexpr bool kFirstStage<esq::EpiStage<1>> = true;
bool first_stage_ok(const esq_epilogue *e) {
    return e->kind == ESQ_EPI_STAGE && e->nt <= 1 && e->in_base;
}
SrcAxpy axpy_of(const esq_epilogue *e) {
    return SrcAxpy{e->in_base, e->in_row, e->in_c, e->in_h};
}

// ---- launch geometry of the 2-D sweeps: one wave tile per 64 column pairs
struct Geo2d {
    unsigned wpr, grid;
};
Geo2d geo2d(int N) {
    Geo2d g;
    g.wpr = (N / 2 + 63) / 64;                                  // wave tiles per row
    const unsigned tiles = g.wpr * (unsigned)N;
    const unsigned nblocks = (tiles + kBlock / 64 - 1) / (kBlock / 64);
    g.grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    return g;
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
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    if (r->N % 2 == 0 && r->N >= 4 && rhs_variant() != 1) {
        const Geo2d g = geo2d(r->N);
        esq::EpiNone ep{};
        hipLaunchKernelGGL((k_heat2d_sweep<esq::EpiNone, SrcPlain>), dim3(g.grid),
                           dim3(kBlock), 0, (hipStream_t)stream, SrcPlain{y}, f, ep, r->N, c, g.grid, g.wpr);
        return (int)hipGetLastError();
    }
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    hipLaunchKernelGGL(k_heat2d, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       y, f, r->N, c, grid, bpr);
    return (int)hipGetLastError();
}
int esq_rhs_bruss2d(void *user, double t, const double *y, double *f, size_t n,
                    void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n) return ESQ_EINVAL;
    const double d = r->alpha * ((double)r->N * (double)r->N);
    if (r->N % 2 == 0 && r->N >= 4 && rhs_variant() != 1) {
        const Geo2d g = geo2d(r->N);
        esq::EpiNone ep{};
        hipLaunchKernelGGL((k_bruss2d_sweep<esq::EpiNone, SrcPlain>), dim3(g.grid),
                           dim3(kBlock), 0, (hipStream_t)stream, SrcPlain{y}, f, ep, r->N, d, r->a, r->b,
                           g.grid, g.wpr);
        return (int)hipGetLastError();
    }
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    static const bool nts = getenv("ESQ_RHS_STORE_NT") && atoi(getenv("ESQ_RHS_STORE_NT"));
    if (nts)
        hipLaunchKernelGGL(k_bruss2d<true>, dim3(grid), dim3(kBlock), 0,
                           (hipStream_t)stream, y, f, r->N, d, r->a, r->b, grid, bpr);
    else
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
    if (r->N % 2 != 0 || r->N < 4) return ESQ_ENOTSUP;
    const Geo2d g = geo2d(r->N);
    if (esq::epilogue_reduces(epi)) {
        if ((int)g.grid > epi->partials_cap) return ESQ_ENOTSUP;
        if (epi->partials_used) *epi->partials_used = (int)g.grid;
    }
    const double d = r->alpha * ((double)r->N * (double)r->N);
    if (epi->in_row && !first_stage_ok(epi)) return ESQ_ENOTSUP;
    const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
        using E = decltype(ep);
        if constexpr (kFirstStage<E>) {
            if (epi->in_row) {
                hipExtLaunchKernelGGL((k_bruss2d_sweep<E, SrcAxpy>), dim3(g.grid),
                                      dim3(kBlock), 0, (hipStream_t)stream,
                                      (hipEvent_t)start_event, (hipEvent_t)stop_event,
                                      0, axpy_of(epi), f, ep, r->N, d, r->a, r->b,
                                      g.grid, g.wpr);
                return;
            }
        }
        hipExtLaunchKernelGGL((k_bruss2d_sweep<E, SrcPlain>), dim3(g.grid),
                              dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                              SrcPlain{y_in}, f, ep, r->N, d, r->a, r->b, g.grid,
                              g.wpr);
    });
    return rc ? rc : (int)hipGetLastError();
}
int esq_rhs_heat2d_fused(void *user, double t, const double *y_in, double *f,
                         const esq_epilogue *epi, size_t n, void *stream,
                         void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n || !epi) return ESQ_EINVAL;
    if (r->N % 2 != 0 || r->N < 4) return ESQ_ENOTSUP;
    const Geo2d g = geo2d(r->N);
    if (esq::epilogue_reduces(epi)) {
        if ((int)g.grid > epi->partials_cap) return ESQ_ENOTSUP;
        if (epi->partials_used) *epi->partials_used = (int)g.grid;
    }
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    if (epi->in_row && !first_stage_ok(epi)) return ESQ_ENOTSUP;
    const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
        using E = decltype(ep);
        if constexpr (kFirstStage<E>) {
            if (epi->in_row) {
                hipExtLaunchKernelGGL((k_heat2d_sweep<E, SrcAxpy>), dim3(g.grid),
                                      dim3(kBlock), 0, (hipStream_t)stream,
                                      (hipEvent_t)start_event, (hipEvent_t)stop_event,
                                      0, axpy_of(epi), f, ep, r->N, c, g.grid, g.wpr);
                return;
            }
        }
        hipExtLaunchKernelGGL((k_heat2d_sweep<E, SrcPlain>), dim3(g.grid),
                              dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                              SrcPlain{y_in}, f, ep, r->N, c, g.grid, g.wpr);
    });
    return rc ? rc : (int)hipGetLastError();
}
int esq_rhs_diag_fused(void *user, double t, const double *y_in, double *f,
                       const esq_epilogue *epi, size_t n, void *stream,
                       void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIAG || n != r->n || !epi) return ESQ_EINVAL;
    const double forcing = r->amp != 0.0 ? r->amp * sin(t) : 0.0;
    // the state vectors are padded to a multiple of 512 doubles
    const size_t n_pad = ((n + 511) / 512) * 512, n2 = n_pad / 2;
    size_t blocks = (n2 + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    if (esq::epilogue_reduces(epi)) {
        if ((int)blocks > epi->partials_cap) return ESQ_ENOTSUP;
        if (epi->partials_used) *epi->partials_used = (int)blocks;
    }
    if (epi->in_row && !first_stage_ok(epi)) return ESQ_ENOTSUP;
    const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
        using E = decltype(ep);
        if constexpr (kFirstStage<E>) {
            if (epi->in_row) {
                hipExtLaunchKernelGGL((k_diag_sweep<E, SrcAxpy>), dim3((unsigned)blocks),
                                      dim3(kBlock), 0, (hipStream_t)stream,
                                      (hipEvent_t)start_event, (hipEvent_t)stop_event,
                                      0, axpy_of(epi), f, ep, r->lam_dev, forcing, n, n2);
                return;
            }
        }
        hipExtLaunchKernelGGL((k_diag_sweep<E, SrcPlain>), dim3((unsigned)blocks),
                              dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                              SrcPlain{y_in}, f, ep, r->lam_dev, forcing, n, n2);
    });
    return rc ? rc : (int)hipGetLastError();
}

static RkcEpi make_epi(const double *yjm2, const double *yn, const double *fn,
                       double mu, double nu, double omn, double hmus, double ajm1,
                       double *out) {
    RkcEpi e{};
    e.yjm2 = yjm2; e.yn = yn; e.fn = fn; e.out = out;
    e.mu = mu; e.nu = nu; e.omn = omn; e.hmus = hmus; e.ajm1 = ajm1;
    return e;
}
int esq_rhs_heat2d_rkc(void *user, double t, const double *yjm1, const double *yjm2,
                       const double *yn, const double *fn, double mu, double nu,
                       double omn, double hmus, double ajm1, double *y_out,
                       size_t n, void *stream, void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n) return ESQ_EINVAL;
    if (r->N % 2 != 0 || r->N < 4) return ESQ_ENOTSUP;
    const Geo2d g = geo2d(r->N);
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL((k_heat2d_sweep<esq::EpiRkc, SrcPlain>), dim3(g.grid),
                          dim3(kBlock), 0, (hipStream_t)stream, (hipEvent_t)start_event,
                          (hipEvent_t)stop_event, 0, SrcPlain{yjm1}, (double *)nullptr,
                          make_epi(yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out),
                          r->N, c, g.grid, g.wpr);
    return (int)hipGetLastError();
}
int esq_rhs_diff3d_rkc(void *user, double t, const double *yjm1, const double *yjm2,
                       const double *yn, const double *fn, double mu, double nu,
                       double omn, double hmus, double ajm1, double *y_out,
                       size_t n, void *stream, void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    if (r->N < 2) return ESQ_ENOTSUP;
    constexpr int R = 8;
    const unsigned NN = (unsigned)r->N * (unsigned)r->N;
    const unsigned bpp = (NN + kBlock - 1) / kBlock;
    const unsigned nb = bpp * (unsigned)((r->N + R - 1) / R);
    const unsigned grid = ((nb + kXcd - 1) / kXcd) * kXcd;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL((k_diff3d_v2<R, true>), dim3(grid), dim3(kBlock), 0,
                          (hipStream_t)stream, (hipEvent_t)start_event,
                          (hipEvent_t)stop_event, 0, yjm1, (double *)nullptr, r->N,
                          c, grid, bpp,
                          make_epi(yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out));
    return (int)hipGetLastError();
}
int esq_rhs_diff3d(void *user, double t, const double *y, double *f, size_t n,
                   void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    if (rhs_variant() != 1 && r->N >= 2) {
        constexpr int R = 8;
        const unsigned NN = (unsigned)r->N * (unsigned)r->N;
        const unsigned bpp = (NN + kBlock - 1) / kBlock;        // blocks per plane
        const unsigned nb = bpp * (unsigned)((r->N + R - 1) / R);
        const unsigned grid = ((nb + kXcd - 1) / kXcd) * kXcd;
        hipLaunchKernelGGL((k_diff3d_v2<R, false>), dim3(grid), dim3(kBlock), 0,
                           (hipStream_t)stream, y, f, r->N, c, grid, bpp, RkcEpi{});
        return (int)hipGetLastError();
    }
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    hipLaunchKernelGGL(k_diff3d, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       y, f, r->N, c, grid, bpr);
    return (int)hipGetLastError();
}

}  // extern "C"
