// esq_stencil3d.hpp -- everything a 3-D seven-point-stencil RHS plugin needs besides its
// POINTWISE FUNCTOR: the sweep with every fused epilogue (esq_rhs_fused_fn), the
// Chebyshev stage (esq_rhs_rkc_fn) and -- for homogeneous Dirichlet problems of one
// field -- the marching chain sweeps of the explicit pairs (esq_chain3d.hpp) and of
// SSV2stab (esq_rkc3d.hpp).  The built-in `Diffusion3D` plugin is its first
// instantiation (esq_rhs_diff3d.hip); a user plugin is another (INTEGRATION.md §4,
// tests/test_gpu_stencil3d.py: the reference's own two demo problems,
// docs/Demo_SSV2stab.ipynb -- a one-field heat equation with a source term and
// time-dependent Dirichlet data, and the two-field combustion problem with mirror
// conditions on three faces).
//
//   struct MyFn {                        // NF fields of N^3 points, state = field 0, 1, ...
//       double p0, p1;                   // parameters (passed by value to the kernels)
//       // homogeneous Dirichlet (every field 0 outside the grid) AND no dependence on
//       // position or time: the 16-byte sweeps and the chain sweeps apply (NF == 1)
//       static constexpr bool kZeroOutside = false, kAutonomous = false;
//       // value of field q at the point OUTSIDE the grid that is the neighbour of grid
//       // point (i, j, l) across `face` (0: i-1, 1: i+1, 2: j-1, 3: j+1, 4: l-1,
//       // 5: l+1); `inside` = the field at (i, j, l) itself (mirror conditions)
//       __device__ double ghost(int q, int face, int i, int j, int l, double inside,
//                               double t) const;
//       // derivatives of all fields at (i, j, l) from their seven-point neighbourhoods
//       __device__ void eval(const esq::Nb3 (&nb)[NF], int i, int j, int l, double t,
//                            double (&f)[NF]) const;
//   };
//   using P = esq::Stencil3D<NF, MyFn>;
//   extern "C" int my_rhs(void* user, double t, const double* y, double* f, size_t n, void* s)
//       { return P::rhs(fn_of(user), N_of(user), t, y, f, s); }
//   extern "C" int my_fused(void* user, double t, const double* y, double* f,
//                           const esq_epilogue* e, size_t n, void* s, void* e0, void* e1)
//       { return P::fused(fn_of(user), N_of(user), t, y, f, e, s, e0, e1); }
//   extern "C" int my_rkc(void* user, double t, const double* yjm1, const double* yjm2,
//                         const double* yn, const double* fn, double mu, double nu, double omn,
//                         double hmus, double ajm1, double* y_out, size_t n, void* s,
//                         void* e0, void* e1)
//       { return P::rkc(fn_of(user), N_of(user), t, yjm1, yjm2, yn, fn, mu, nu, omn, hmus,
//                       ajm1, y_out, s, e0, e1); }
//   (P::chain / P::rkc_chain: functors with kZeroOutside && kAutonomous, NF == 1)
//
// k = (i*N + j)*N + l: l is the fastest index (lane <-> l: coalesced rows).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/extensisq_amd.h"
#include "esq_chain3d.hpp"
#include "esq_epilogue.hpp"
#include "esq_plugin.hpp"
#include "esq_rkc3d.hpp"
#include "esq_stencil2d.hpp"      // band_block, kXcd
#include "esq_terms.hpp"

namespace esq {

// the seven-point neighbourhood of one field at one grid point
struct Nb3 {
    double below, above;      // i - 1, i + 1
    double up, dn;            // j - 1, j + 1
    double lf, rt;            // l - 1, l + 1
    double c;                 // the point itself
};

// a one-field homogeneous-Dirichlet autonomous functor as the fast sweeps see it
template <class Fn>
struct PointSt {
    Fn fn;
    __device__ __forceinline__ double eval(double below, double above, double up, double dn,
                                           double lf, double rt, double centre) const {
        const Nb3 nb[1] = {{below, above, up, dn, lf, rt, centre}};
        double f[1];
        fn.eval(nb, 0, 0, 0, 0.0, f);
        return f[0];
    }
};

// ---------------------------------------------------------------------------
// GENERIC SWEEP: one thread per grid point, all NF fields; neighbours by direct
// loads (L1 / L2 serve the reuse), ghost values from the functor, the epilogue's
// one-double-per-thread twins (esq_epilogue.hpp).  Any N, any boundary condition,
// any number of pointwise-coupled fields.
// ---------------------------------------------------------------------------
template <int NF, class Fn, class Epi>
__global__ __launch_bounds__(kBlock) void k_stencil3d_points(
    const double *__restrict__ u, double *__restrict__ f, Fn fn, int N, double t,
    unsigned nblocks, Epi epi) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const size_t NN = (size_t)N * (size_t)N, npts = NN * (size_t)N;
    const size_t p = (size_t)lb * kBlock + threadIdx.x;
    double local = 0.0;
    if (p < npts) {
        const int i = (int)(p / NN), j = (int)((p / (size_t)N) % (size_t)N),
                  l = (int)(p % (size_t)N);
        Nb3 nb[NF];
        typename Epi::In1 in[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            const double *v = u + (size_t)q * npts + p;
            epi.load1(in[q], (size_t)q * npts + p);
            const double c = v[0];
            nb[q].c = c;
            nb[q].below = i > 0 ? v[-(ptrdiff_t)NN] : fn.ghost(q, 0, i, j, l, c, t);
            nb[q].above = i + 1 < N ? v[NN] : fn.ghost(q, 1, i, j, l, c, t);
            nb[q].up = j > 0 ? v[-N] : fn.ghost(q, 2, i, j, l, c, t);
            nb[q].dn = j + 1 < N ? v[N] : fn.ghost(q, 3, i, j, l, c, t);
            nb[q].lf = l > 0 ? v[-1] : fn.ghost(q, 4, i, j, l, c, t);
            nb[q].rt = l + 1 < N ? v[1] : fn.ghost(q, 5, i, j, l, c, t);
        }
        double out[NF];
        fn.eval(nb, i, j, l, t, out);
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            const size_t k = (size_t)q * npts + p;
            epi.store_f1(f, k, out[q]);
            epi.finish1(in[q], out[q], nb[q].c, k, local);
        }
    }
    if (Epi::kReduce) block_partial(local, epi.red.partials);
}

// ---------------------------------------------------------------------------
// FAST SWEEPS (one field, zero outside, autonomous: St::eval(7 values))
// ---------------------------------------------------------------------------
// scalar reference kernel (ESQ_RHS_VARIANT=1; N = 1)
template <class St>
__global__ __launch_bounds__(kBlock) void k_stencil3d_scalar(const double *__restrict__ u,
                                                             double *__restrict__ f, int N,
                                                             St st, unsigned nblocks,
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
    f[k] = st.eval(a0, a1, b0, b1, c0, c1, uc);
}

// marching version: a thread owns one (j, l) column of the grid (flattened plane
// index p) and walks R planes along i with a rolling (below, centre, above) window;
// the l-neighbours come from adjacent lanes, the j-neighbours are two coalesced
// loads of the centre plane.  3 loads per output instead of 7.  One double per
// thread (N may be odd).
template <int R, class St, class Epi>
__global__ __launch_bounds__(kBlock) void k_stencil3d_march(
    const double *__restrict__ u, double *__restrict__ f, int N, St st, unsigned nblocks,
    unsigned bpp, Epi epi) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const int i0 = (int)(lb / bpp) * R;
    const unsigned p = (lb % bpp) * kBlock + threadIdx.x;     // plane index
    const unsigned NN = (unsigned)N * (unsigned)N;
    double local = 0.0;
    if (i0 < N) {                                             // block-uniform
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
            if (i < N) {
                const double above = at(i + 1);
                double c0 = lane_left(centre), c1 = lane_right(centre);
                if (live) {
                    const size_t k = (size_t)i * NN + p;
                    typename Epi::In1 in;
                    epi.load1(in, k);
                    const double *pl = u + (size_t)i * NN;
                    if (l == 0) c0 = 0.0; else if (lane == 0) c0 = pl[p - 1];
                    if (l + 1 == (unsigned)N) c1 = 0.0;
                    else if (lane == 63 || p + 1 >= NN) c1 = pl[p + 1];
                    const double b0 = j > 0 ? pl[p - N] : 0.0;
                    const double b1 = j + 1 < (unsigned)N ? pl[p + N] : 0.0;
                    const double fy = st.eval(below, above, b0, b1, c0, c1, centre);
                    epi.store_f1(f, k, fy);
                    epi.finish1(in, fy, centre, k, local);
                }
                below = centre;
                centre = above;
            }
        }
    }
    if (Epi::kReduce) block_partial(local, epi.red.partials);
}

// 16-byte version for ANY N (odd edges too): a thread owns one ALIGNED pair
// (e0, e0 + 1) of the flattened state -- so every access of the epilogue (K rows,
// y, outputs: pointwise data) and the sweep's own centre load and store are
// 16-byte accesses, as in the 2-D sweeps.  The pair may straddle a grid row (or a
// plane) when N is odd: each element carries its own (i, j, l) and its own
// boundary tests.  The four neighbour pairs one row / one plane away start at
// e0 -+ N, e0 -+ N^2 -- 8-byte aligned only for odd N: 16-byte loads at 8-byte
// alignment (gfx950 serves them; the compiler emits global_load_dwordx4 for the
// aligned(8) vector type).  No marching: the planes above and below are re-read
// from L2 (a plane is 0.2-1.3 MB; workgroups of one XCD sweep a contiguous range
// of the flattened state).
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));
template <class St, class Epi>
__global__ __launch_bounds__(kBlock) void k_stencil3d_pairs(
    const double *__restrict__ u, double *__restrict__ f, int N, St st, unsigned nblocks,
    size_t n, Epi epi) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const size_t q = (size_t)lb * kBlock + threadIdx.x;          // pair index
    const size_t e0 = 2 * q;
    const unsigned NN = (unsigned)N * (unsigned)N;
    double local = 0.0;
    const bool live0 = e0 < n, live1 = e0 + 1 < n;
    const int lane = threadIdx.x & 63;
    // centre pair first: the lane shifts need it from every lane of the wave
    double2 cc = make_double2(0.0, 0.0);
    if (live0) cc = ld2(u, q);                                     // (padding is zero)
    double lf = lane_left(cc.y), rt = lane_right(cc.x);
    if (live0) {
        typename Epi::In in;
        epi.load(in, q);
        // (i, j, l) of both elements
        const unsigned i0 = (unsigned)(e0 / NN), r0 = (unsigned)(e0 - (size_t)i0 * NN);
        const unsigned j0 = r0 / (unsigned)N, l0 = r0 - j0 * (unsigned)N;
        unsigned i1 = i0, j1 = j0, l1 = l0 + 1;
        if (l1 == (unsigned)N) { l1 = 0; if (++j1 == (unsigned)N) { j1 = 0; ++i1; } }
        const unsigned last = (unsigned)N - 1;
        auto pair_at = [&](size_t e) -> double2 {     // u[e], u[e + 1], e within [0, n - 2]
            const v2d_a8 v = *reinterpret_cast<const v2d_a8 *>(u + e);
            return make_double2(v.x, v.y);
        };
        // neighbour pairs; an address outside the vector is replaced by the centre's
        // (the values are then not used)
        const bool dn_ok = e0 >= NN, up_ok = e0 + NN + 1 < n + (n & 1);
        const bool jm_ok = e0 >= (size_t)N, jp_ok = e0 + N + 1 < n + (n & 1);
        const double2 below = pair_at(dn_ok ? e0 - NN : e0);
        const double2 above = pair_at(up_ok ? e0 + NN : e0);
        const double2 b0 = pair_at(jm_ok ? e0 - N : e0);
        const double2 b1 = pair_at(jp_ok ? e0 + N : e0);
        if (lane == 0 && e0 > 0) lf = u[e0 - 1];
        if (lane == 63 && e0 + 2 < n) rt = u[e0 + 2];
        // element x
        const double xb = (i0 > 0 && dn_ok) ? below.x : 0.0;
        const double xa = (i0 < last && up_ok) ? above.x : 0.0;
        const double x0 = (j0 > 0 && jm_ok) ? b0.x : 0.0;
        const double x1 = (j0 < last && jp_ok) ? b1.x : 0.0;
        const double xl = l0 > 0 ? lf : 0.0;
        const double xr = l0 < last ? cc.y : 0.0;
        double2 fy;
        fy.x = st.eval(xb, xa, x0, x1, xl, xr, cc.x);
        // element y (may be the first of the next row / plane, or padding)
        fy.y = 0.0;
        if (live1) {
            // (the one pair whose second element opens plane 1 / row 1 of plane 0
            // has its lower neighbour at element 0, its first element none)
            const double yb = i1 > 0 ? (dn_ok ? below.y : u[e0 + 1 - NN]) : 0.0;
            const double ya = (i1 < last && up_ok) ? above.y : 0.0;
            const double y0 = j1 > 0 ? (jm_ok ? b0.y : u[e0 + 1 - N]) : 0.0;
            const double y1 = (j1 < last && jp_ok) ? b1.y : 0.0;
            const double yl = l1 > 0 ? cc.x : 0.0;
            const double yr = l1 < last ? rt : 0.0;
            fy.y = st.eval(yb, ya, y0, y1, yl, yr, cc.y);
        }
        epi.store_f(f, q, fy);
        epi.finish(in, fy, cc, q, local);
    }
    if (Epi::kReduce) block_partial(local, epi.red.partials);
}

// tuning / test knobs of a 3-D plugin OBJECT (the built-in plugin: options RKC_FORCE,
// RKC_PLANES, DIFF3D_R of esq_rhs_set_options, defaults from the environment when the
// object is made -- stencil3d_tuning_default; a user plugin fills the struct itself):
// `force` chain sweeps on grids of any size, `planes` per tile of the chain sweeps
// (0: chosen), `march_r` 0 = the 16-byte pair sweep (default), 1 / 2 / 4 / 8 = the
// marching sweep with that many planes per workgroup.  (jt, nw: shape of the Chebyshev
// chain sweeps, 0 = the measured one -- tools/rkc_shape_sweep.sh, round 4)
struct Stencil3dTuning {
    int force = 0, planes = 0, jt = 0, nw = 0, march_r = 0;
};
inline Stencil3dTuning stencil3d_tuning_default() {
    Stencil3dTuning t;
    if (const char *e = env_get("RKC_FORCE")) t.force = atoi(e);
    if (const char *e = env_get("RKC_PLANES")) t.planes = atoi(e);
    if (const char *e = env_get("DIFF3D_R")) t.march_r = atoi(e);
    return t;
}

// ---------------------------------------------------------------------------
// The entry points of a 3-D stencil plugin, by functor.
// ---------------------------------------------------------------------------
template <int NF, class Fn>
struct Stencil3D {
    static constexpr bool kFast = NF == 1 && Fn::kZeroOutside && Fn::kAutonomous;
    using St = PointSt<Fn>;
    static size_t points(int N) { return (size_t)N * (size_t)N * (size_t)N; }

    // ---- launch geometry
    static unsigned grid_points(int N) {
        const size_t nb = (points(N) + kBlock - 1) / kBlock;
        return (unsigned)(((nb + kXcd - 1) / kXcd) * kXcd);
    }
    static unsigned grid_pairs(int N) {
        const size_t pairs = (points(N) + 1) / 2;
        const unsigned nb = (unsigned)((pairs + kBlock - 1) / kBlock);
        return ((nb + kXcd - 1) / kXcd) * kXcd;
    }
    template <int R>
    static unsigned grid_march(int N) {
        const unsigned NN = (unsigned)N * (unsigned)N;
        const unsigned bpp = (NN + kBlock - 1) / kBlock;
        const unsigned nb = bpp * (unsigned)((N + R - 1) / R);
        return ((nb + kXcd - 1) / kXcd) * kXcd;
    }
    // workgroups (= partial sums of a reducing epilogue) of the sweep `fused` runs
    static unsigned grid_fused(int N, const Stencil3dTuning &tune) {
        if constexpr (kFast) return tune.march_r == 0 ? grid_pairs(N) : grid_march<8>(N);
        return grid_points(N);
    }

    template <class Epi>
    static void launch_points(const Fn &fn, int N, double t, const double *y_in, double *f,
                              const Epi &epi, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
        const unsigned grid = grid_points(N);
        hipExtLaunchKernelGGL((k_stencil3d_points<NF, Fn, Epi>), dim3(grid), dim3(kBlock), 0, s,
                              e0, e1, 0, y_in, f, fn, N, t, grid, epi);
    }
    template <class Epi>
    static void launch_pairs(const Fn &fn, int N, const double *y_in, double *f, const Epi &epi,
                             hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
        const unsigned grid = grid_pairs(N);
        hipExtLaunchKernelGGL((k_stencil3d_pairs<St, Epi>), dim3(grid), dim3(kBlock), 0, s, e0,
                              e1, 0, y_in, f, N, St{fn}, grid, points(N), epi);
    }
    template <int R, class Epi>
    static void launch_march(const Fn &fn, int N, const double *y_in, double *f, const Epi &epi,
                             hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
        const unsigned NN = (unsigned)N * (unsigned)N;
        const unsigned bpp = (NN + kBlock - 1) / kBlock;            // blocks per plane
        const unsigned grid = grid_march<R>(N);
        hipExtLaunchKernelGGL((k_stencil3d_march<R, St, Epi>), dim3(grid), dim3(kBlock), 0, s, e0,
                              e1, 0, y_in, f, N, St{fn}, grid, bpp, epi);
    }
    // one sweep with epilogue `epi`, by the functor's class and the tuning (callers have
    // refused N < 2 for the fast class).  SHORT: the Chebyshev stage also runs the
    // marching sweep with 1 / 2 / 4 planes per workgroup (instantiated for its epilogue
    // only: every instantiation of this function costs compile time x ~60 epilogues)
    template <bool SHORT = false, class Epi>
    static void sweep(const Fn &fn, int N, double t, const double *y_in, double *f,
                      const Epi &epi, const Stencil3dTuning &tune, hipStream_t s,
                      hipEvent_t e0, hipEvent_t e1) {
        if constexpr (kFast) {
            if (tune.march_r == 0) { launch_pairs(fn, N, y_in, f, epi, s, e0, e1); return; }
            if constexpr (SHORT) {
                switch (tune.march_r) {
                    case 1: launch_march<1>(fn, N, y_in, f, epi, s, e0, e1); return;
                    case 2: launch_march<2>(fn, N, y_in, f, epi, s, e0, e1); return;
                    case 4: launch_march<4>(fn, N, y_in, f, epi, s, e0, e1); return;
                    default: break;
                }
            }
            launch_march<8>(fn, N, y_in, f, epi, s, e0, e1);
        } else {
            launch_points(fn, N, t, y_in, f, epi, s, e0, e1);
        }
    }

    // f = fun(t, y)                                                 (esq_rhs_fn)
    static int rhs(const Fn &fn, int N, double t, const double *y, double *f, void *stream,
                   const Stencil3dTuning &tune = Stencil3dTuning{}, bool scalar = false) {
        if (N < 1) return ESQ_EINVAL;
        if constexpr (kFast) {
            if (scalar || N < 2) {
                const unsigned bpr = (N + kBlock - 1) / kBlock;
                const unsigned nblocks = bpr * (unsigned)N * (unsigned)N;
                const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
                hipLaunchKernelGGL((k_stencil3d_scalar<St>), dim3(grid), dim3(kBlock), 0,
                                   (hipStream_t)stream, y, f, N, St{fn}, grid, bpr);
                return (int)hipGetLastError();
            }
        }
        EpiNone none{};
        sweep(fn, N, t, y, f, none, tune, (hipStream_t)stream, nullptr, nullptr);
        return (int)hipGetLastError();
    }

    // the sweep + a pointwise epilogue, every kind               (esq_rhs_fused_fn)
    static int fused(const Fn &fn, int N, double t, const double *y_in, double *f,
                     const esq_epilogue *epi, void *stream, void *start_event,
                     void *stop_event, const Stencil3dTuning &tune = Stencil3dTuning{}) {
        if (!epi) return ESQ_EINVAL;
        if (N < 2 || epi->in_row || epi->is_complex) return ESQ_ENOTSUP;
        const unsigned grid = grid_fused(N, tune);
        if (epilogue_reduces(epi)) {
            if ((int)grid > epi->partials_cap) return ESQ_ENOTSUP;
            if (epi->partials_used && !epi->dry_run) *epi->partials_used = (int)grid;
        }
        const int rc = dispatch_epilogue(epi, [&](auto ep) {
            sweep(fn, N, t, y_in, f, ep, tune, (hipStream_t)stream, (hipEvent_t)start_event,
                  (hipEvent_t)stop_event);
        });
        return (rc || epi->dry_run) ? rc : (int)hipGetLastError();
    }

    // sweep + Chebyshev recursion, f not stored                    (esq_rhs_rkc_fn)
    static int rkc(const Fn &fn, int N, double t, const double *yjm1, const double *yjm2,
                   const double *yn, const double *f_n, double mu, double nu, double omn,
                   double hmus, double ajm1, double *y_out, void *stream, void *start_event,
                   void *stop_event, const Stencil3dTuning &tune = Stencil3dTuning{}) {
        if (N < 2) return ESQ_ENOTSUP;
        EpiRkc e{};
        e.yjm2 = yjm2; e.yn = yn; e.fn = f_n; e.out = y_out;
        e.mu = mu; e.nu = nu; e.omn = omn; e.hmus = hmus; e.ajm1 = ajm1;
        // (planes per workgroup of the marching sweep: with the five vectors of a
        // Chebyshev stage resident in the Infinity Cache short marches win)
        sweep</*SHORT=*/true>(fn, N, t, yjm1, nullptr, e, tune, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event);
        return (int)hipGetLastError();
    }

    // `depth` consecutive stages in one marching sweep             (esq_rhs_chain_fn)
    static int chain(const Fn &fn, int N, const double *y_in, const esq_chain *ch, void *stream,
                     void *start_event, void *stop_event,
                     const Stencil3dTuning &tune = Stencil3dTuning{}) {
        if constexpr (kFast) {
            return chain3d(St{fn}, N, y_in, ch, tune.planes, tune.force != 0, stream,
                           start_event, stop_event);
        } else {
            return ESQ_ENOTSUP;
        }
    }

    // `depth` consecutive Chebyshev stages in one marching sweep (esq_rhs_rkc_chain_fn;
    // the forms that open and end a step included).  Grids below 48^3 stay with one
    // launch per stage (a tile's run-in planes and halo points outweigh the saving).
    static int rkc_chain(const Fn &fn, int N, const esq_rkc_chain *ch, void *stream,
                         void *start_event, void *stop_event,
                         const Stencil3dTuning &tune = Stencil3dTuning{}) {
        if (!ch) return ESQ_EINVAL;
        if (!ch->yjm2 || !ch->yn || !ch->fn || !ch->out) return ESQ_EINVAL;
        if (!ch->yjm1 && ch->yjm2 != ch->yn) return ESQ_EINVAL;       // FIRST: y_{j-2} = y_n
        if (ch->fy_out && (ch->out_prev || !ch->partials)) return ESQ_EINVAL;
        if constexpr (!kFast) {
            return ESQ_ENOTSUP;
        } else {
            // (32-bit byte offsets into a vector: esq_rkc3d.hpp)
            if ((unsigned long long)points(N) * 8ull > 0xffffffffull - 16ull) return ESQ_ENOTSUP;
            if (N < 2 || (N < 48 && !tune.force)) return ESQ_ENOTSUP;
            hipStream_t s = (hipStream_t)stream;
            hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
            switch (ch->depth + (ch->fy_out ? 1 : 0)) {           // stage slots of the sweep
                case 2: return rkc_by_depth<2>(fn, N, ch, tune, s, e0, e1);
                case 3: return rkc_by_depth<3>(fn, N, ch, tune, s, e0, e1);
                case 4: return rkc_by_depth<4>(fn, N, ch, tune, s, e0, e1);
                case 5: return rkc_by_depth<5>(fn, N, ch, tune, s, e0, e1);
                case 6: return rkc_by_depth<6>(fn, N, ch, tune, s, e0, e1);
                default: return ESQ_ENOTSUP;
            }
        }
    }

  private:
    // rows per thread and waves per workgroup by depth: the windows, the y_n / f_n
    // delay lines and one plane of operands in flight are (4 D + 7) JT doubles per
    // thread.  D: stage slots of the sweep (LAST: the chain's depth + 1)
    template <int D, int JT, int NW, bool FIRST = false, bool LAST = false>
    static int launch_rkc3d(const Fn &fn, int N, const esq_rkc_chain *ch,
                            const Stencil3dTuning &tune, hipStream_t stream, hipEvent_t e0,
                            hipEvent_t e1) {
        auto kern = k_rkc3d_chain<D, JT, NW, St, FIRST, LAST>;
        // workgroups resident on the chip: per CU from the occupancy of this
        // instantiation (asked once), times the CUs of the device in use
        static std::atomic<int> per_cu_cache{0};
        int per_cu = per_cu_cache.load(std::memory_order_relaxed);
        if (per_cu == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, 0) !=
                    hipSuccess || per_cu < 1)
                per_cu = 1;
            per_cu_cache.store(per_cu, std::memory_order_relaxed);
        }
        const int slots = device_cus() * per_cu;
        if (NW * JT - 2 * D < 1) return ESQ_ENOTSUP;
        const Geo3d g = geo_rkc3d(N, D, JT, NW, slots, tune.planes);
        Rkc3dArgs<D> a;
        a.a = ch->yjm1; a.b = ch->yjm2; a.yn = ch->yn; a.fn = ch->fn;
        a.out = ch->out; a.outp = ch->out_prev;
        a.hmus1 = ch->hmus_first;
        a.h04 = 0.0;
        a.red = RedArgs{};
        for (int k = 0; k < D; ++k) {
            const bool stage = k < ch->depth;                  // (LAST: slot D - 1 is the end)
            a.mu[k] = stage ? ch->mu[k] : 0.0; a.nu[k] = stage ? ch->nu[k] : 0.0;
            a.omn[k] = stage ? ch->omn[k] : 0.0; a.hmus[k] = stage ? ch->hmus[k] : 0.0;
            a.ajm1[k] = stage ? ch->ajm1[k] : 0.0;
        }
        if constexpr (LAST) {
            if ((int)g.grid > ch->partials_cap) return ESQ_ENOTSUP;
            if (ch->partials_used) *ch->partials_used = (int)g.grid;
            a.out = ch->fy_out;                                // the slot's "result" ...
            a.outp = ch->out;                                  // ... and its input: y_{n+1}
            a.h04 = 0.4 * ch->h;
            a.red.atol_vec = ch->atol_vec; a.red.atol_s = ch->atol_s; a.red.rtol = ch->rtol;
            a.red.n_valid = ch->n_valid; a.red.partials = ch->partials;
        }
        // (FIRST: two vectors, both on the first input's wider plane range)
        if (ch->read_amplification) *ch->read_amplification = amp_rkc3d(g, D);
        hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(64 * NW), 0, stream, e0, e1, 0, a,
                              St{fn}, g);
        return (int)hipGetLastError();
    }
    template <int D>
    static int rkc_by_depth(const Fn &fn, int N, const esq_rkc_chain *ch,
                            const Stencil3dTuning &tune, hipStream_t stream, hipEvent_t e0,
                            hipEvent_t e1) {
        // defaults (N = 159 / 400, ms per step, tools/rkc_shape_sweep.sh): depth 4 as
        // sixteen waves of two rows (four waves per SIMD) 1.35 / 20.7, as eight waves of
        // four rows 1.43 / 22.4; depth 3 on 5 x 8 1.56 / 21.4; depth 2 on 4 x 8 1.92
        constexpr int djt = D == 2 ? 4 : D == 3 ? 5 : D == 4 ? 2 : D == 5 ? 4 : 3;
        constexpr int dnw = D == 4 ? 16 : 8;
        const int jt = tune.jt > 0 ? tune.jt : djt, nw = tune.nw > 0 ? tune.nw : dnw;
#define ESQ_RKC_SHAPE(DD, JJ, WW) \
    if (jt == JJ && nw == WW) return launch_rkc3d<DD, JJ, WW>(fn, N, ch, tune, stream, e0, e1);
        // the form that opens a step (ch->yjm1 == NULL) exists on each depth's default shape
        if (!ch->yjm1) {
            if (jt == djt && nw == dnw && !ch->fy_out)
                return launch_rkc3d<D, djt, dnw, true>(fn, N, ch, tune, stream, e0, e1);
            return ESQ_ENOTSUP;
        }
        if (ch->fy_out) {                     /* LAST: D = the chain's depth + 1 */
            if constexpr (D >= 3 && D <= 5) {
                if (jt == djt && nw == dnw)
                    return launch_rkc3d<D, djt, dnw, false, true>(fn, N, ch, tune, stream, e0, e1);
            }
            return ESQ_ENOTSUP;
        }
        if constexpr (D == 2) {
            ESQ_RKC_SHAPE(2, 6, 8) ESQ_RKC_SHAPE(2, 3, 16) ESQ_RKC_SHAPE(2, 4, 8)
        } else if constexpr (D == 3) {
            ESQ_RKC_SHAPE(3, 5, 8) ESQ_RKC_SHAPE(3, 2, 16) ESQ_RKC_SHAPE(3, 4, 8)
        } else if constexpr (D == 4) {
            ESQ_RKC_SHAPE(4, 4, 8) ESQ_RKC_SHAPE(4, 2, 16) ESQ_RKC_SHAPE(4, 3, 8)
        } else if constexpr (D == 5) {
            ESQ_RKC_SHAPE(5, 4, 8) ESQ_RKC_SHAPE(5, 3, 8)
        } else if constexpr (D == 6) {
            ESQ_RKC_SHAPE(6, 3, 8)
        }
#undef ESQ_RKC_SHAPE
        return ESQ_ENOTSUP;
    }
};

}  // namespace esq
