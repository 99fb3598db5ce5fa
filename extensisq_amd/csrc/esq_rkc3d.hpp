// esq_rkc3d.hpp -- D consecutive Runge-Kutta-Chebyshev stages of a 3-D seven-point
// stencil plugin in ONE marching sweep (esq_rhs_rkc_chain_fn,
// include/extensisq_amd.h; reference loop: extensisq/sommeijer.py:291-329).
//
// What it removes.  One launch per stage (esq_rhs_rkc_fn) reads y_{j-1}, y_{j-2},
// y_n, f_n and writes y_j: 5 words per element and stage, ~100 stages per step.
// With  Y_0 = y_{j-1},  Y_{-1} = y_{j-2}  a chain evaluates
//     Y_{k+1} = mu_k Y_k + nu_k Y_{k-1} + omn_k y_n + hmus_k (f(Y_k) - a_k f_n)
// for k = 0 .. D-1 in one kernel: the four inputs are read once, only Y_D and
// Y_{D-1} (the next chain's inputs) are written -- 6 words per D stages, plus
// the halo the tile geometry re-reads.
//
// Geometry.  A workgroup of NW waves owns a patch of the (j, l) plane and marches
// along i (the slowest index, k = (i*N + j)*N + l); stage k runs k planes behind
// stage 0, so Y_k's three-plane window is always complete when stage k needs it:
//   * lane  <->  l  (64 consecutive points of a grid row: coalesced 512-B rows),
//     left/right neighbours by DPP wave shifts (no LDS, no bpermute);
//   * a thread holds JT consecutive rows j of its wave's slice, so the j-neighbours
//     are its own registers -- except the slice's first and last row, which
//     the neighbouring waves hand over through LDS: 2 D doubles per thread written
//     and read per plane, ONE workgroup barrier per plane (all D stages'
//     centre planes were completed an iteration ago, so the exchange is off
//     the stage-to-stage dependency chain), two buffers by iteration parity;
//   * the patch is NW*JT rows x 64 lanes; Y_{k+1} is valid where Y_k was valid one
//     point further out, so D points on every side are halo: stored region
//     (NW*JT - 2D) x (64 - 2D), planes [i_lo, i_hi); D-1 planes of run-in and
//     run-out per tile.
// Dirichlet: every iterate is zero outside the grid -- a stage's result is masked
// with the domain, exactly what the one-stage kernels' zero neighbours mean.
//
// Bit-identical to D launches of the one-stage entry: the same stencil expression
// (St::eval) and the same left-to-right recursion (`one`), each operation rounded.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_terms.hpp"

namespace esq {

template <int D>
struct Rkc3dArgs {
    const double *a, *b, *yn, *fn;          // y_{j-1}, y_{j-2}, y_n, f_n
    double *out, *outp;                     // Y_D, Y_{D-1} (outp may be null)
    double mu[D], nu[D], omn[D], hmus[D], ajm1[D];
    double hmus1;                           // FIRST: a = y_n + hmus1 * f_n, b = y_n
    // LAST: the last of the D stage slots is not a Chebyshev stage but the end of the
    // step: f = St(Y_{D-1}) -> out, Y_{D-1} -> outp, and the partial sums of the error
    // estimate 0.8 (y_n - y) + 0.4 h (f_n + f)  (ESQ_EPI_RKCERR, sommeijer.py:214-220)
    double h04;
    RedArgs red;
};

// tiles: TL x TJ patches per plane, each VL x VJ stored points, R planes deep
struct Geo3d {
    int N, R, VJ, VL;
    unsigned TJ, TL, ntiles, grid;
};

// planes per tile: the launch runs as `rounds` rounds of `slots` resident
// workgroups, each marching R + 2(D-1) planes (+ a fixed start-up cost of about
// two planes); pick the R that minimises rounds x march.  planes > 0 forces R.
inline Geo3d geo_rkc3d(int N, int D, int JT, int NW, int slots, int planes = 0) {
    Geo3d g;
    g.N = N;
    const int maxVJ = NW * JT - 2 * D, maxVL = 64 - 2 * D;
    g.TL = (unsigned)((N + maxVL - 1) / maxVL);
    g.VL = (N + (int)g.TL - 1) / (int)g.TL;
    g.TJ = (unsigned)((N + maxVJ - 1) / maxVJ);
    g.VJ = (N + (int)g.TJ - 1) / (int)g.TJ;
    const size_t inplane = (size_t)g.TJ * g.TL;
    int R = planes;
    if (R <= 0) {
        double best = 0.0;
        for (int r = 1; r <= N; ++r) {
            const size_t tiles = inplane * (size_t)((N + r - 1) / r);
            const size_t rounds = (tiles + (size_t)slots - 1) / (size_t)slots;
            const double cost = (double)rounds * (double)(r + 2 * (D - 1) + 2);
            if (R == 0 || cost < best) { best = cost; R = r; }
        }
    }
    if (R > N) R = N;
    g.R = R;
    g.ntiles = (unsigned)(inplane * (size_t)((N + R - 1) / R));
    g.grid = ((g.ntiles + 7u) / 8u) * 8u;
    return g;
}
// bytes read per byte of the four input vectors, averaged: in the plane every
// vector is loaded on the stored patch + D points per side, along i the first
// input on D planes more per side, the other three on D - 1 (the library books
// the factor in the launch's designed traffic)
inline double amp_rkc3d(const Geo3d &g, int D) {
    auto cover = [&](int v, int halo) {          // points loaded along one axis / N
        double s = 0.0;
        for (int lo = 0; lo < g.N; lo += v) {
            const int hi = lo + v < g.N ? lo + v : g.N;
            const int a = lo - halo > 0 ? lo - halo : 0;
            const int b = hi + halo < g.N ? hi + halo : g.N;
            s += b - a;
        }
        return s / g.N;
    };
    const double plane = cover(g.VL, D) * cover(g.VJ, D);
    return plane * (cover(g.R, D) + 3.0 * cover(g.R, D - 1)) / 4.0;
}

// Buffer addressing (esq_terms.hpp: raw buffer resources over the whole vector,
// masked lanes at offset 0xffffffff): every load and store of the marching loop is
// unconditional; the plane offset travels in the scalar offset.
// St::eval(below, above, up, down, left, right, centre) -> derivative at the point
// (i-1, i+1; j-1, j+1; l-1, l+1); autonomous stencils only (t is not passed).
// Vectors of at most 4 GiB - 16 B (the offsets are 32-bit).
// FIRST: the chain opens a step -- its first input is the first Chebyshev iterate
// y_1 = y_n + hmus1 * f_n (sommeijer.py:289), formed where the window needs it from
// the two vectors the sweep reads anyway (one plane earlier than before), and
// y_{j-2} = y_n: two loads per point and plane instead of four, and no sweep that
// writes y_1.
// LAST: the chain ends a step (Rkc3dArgs::h04): D - 1 Chebyshev stages and the sweep
// that evaluates f(t + h, y_{n+1}) with the error estimate as stage slot D - 1 -- the
// final iterate is still in the window registers, y_n and f_n arrive down the delay
// lines the stages use anyway.
#ifndef ESQ_RKC3D_EARLY
#define ESQ_RKC3D_EARLY -1
#endif
template <int D, int JT, int NW, class St, bool FIRST = false, bool LAST = false>
__global__ __launch_bounds__(64 * NW) void k_rkc3d_chain(Rkc3dArgs<D> ca, St st, Geo3d g) {
    // edge rows of the D centre planes: slot w + 1 belongs to wave w, slots 0 and
    // NW + 1 stay zero (outside the patch), two buffers by iteration parity
    __shared__ double xch[2][D][NW + 2][2][64];
    // XCD x (workgroups with blockIdx % 8 == x) takes a contiguous band of tiles:
    // neighbouring patches and plane ranges meet in that XCD's L2
    const unsigned per = g.grid / 8u;
    const unsigned lb = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (lb >= g.ntiles) {                                        // workgroup-uniform
        if constexpr (LAST) block_partial_w<NW>(0.0, ca.red.partials);
        return;
    }
    double local = 0.0;                                          // LAST: error sum
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int N = g.N;
    const int tl = (int)(lb % g.TL), tj = (int)((lb / g.TL) % g.TJ);
    const int ti = (int)(lb / (g.TL * g.TJ));
    const int l = tl * g.VL - D + lane;
    const int jb = tj * g.VJ - D + w * JT;
    const int i_lo = ti * g.R;
    const int i_hi = i_lo + g.R < N ? i_lo + g.R : N;
    const int Re = i_hi - i_lo;
    const unsigned plane_bytes = (unsigned)N * (unsigned)N * 8u;
    const size_t vec_bytes = (size_t)plane_bytes * (size_t)N;
    const rsrc_t ra = make_rsrc(ca.a, vec_bytes), rb = make_rsrc(ca.b, vec_bytes),
                 ry = make_rsrc(ca.yn, vec_bytes), rf = make_rsrc(ca.fn, vec_bytes),
                 ro = make_rsrc(ca.out, vec_bytes),
                 rp = make_rsrc(ca.outp, ca.outp ? vec_bytes : 0);   // null: stores dropped
    // in[r]: the point is in the grid AND within D points of the stored patch
    // (further out nothing reaches a stored value: treated like the outside, zero)
    const bool l_in = l >= 0 && l < N && lane < g.VL + 2 * D;
    const bool l_own = l_in && lane >= D && lane < D + g.VL;
    bool in[JT];
    unsigned vo[JT], so[JT];          // byte offsets inside a plane: loads / stores
#pragma unroll
    for (int r = 0; r < JT; ++r) {
        const int j = jb + r;
        in[r] = l_in && j >= 0 && j < N && j < (tj + 1) * g.VJ + D;
        const bool own = in[r] && l_own && j >= tj * g.VJ && j < (tj + 1) * g.VJ;
        const unsigned off = ((unsigned)j * (unsigned)N + (unsigned)l) * 8u;
        vo[r] = in[r] ? off : 0xffffffffu;
        so[r] = own ? off : 0xffffffffu;
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    xch[b][k][0][e][lane] = 0.0;
                    xch[b][k][NW + 1][e][lane] = 0.0;
                }
    }
    auto ld = [&](rsrc_t v, int i, int r) -> double {
        const bool ok = i >= 0 && i < N;                         // uniform
        return buf_ld(v, ok ? vo[r] : 0xffffffffu, ok ? (unsigned)i * plane_bytes : 0u);
    };
    auto one = [&](int k, double yjm1, double bb, double c0, double gg, double fy) -> double {
        return __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(ca.mu[k], yjm1), __dmul_rn(ca.nu[k], bb)),
                                   __dmul_rn(ca.omn[k], c0)),
                         __dmul_rn(ca.hmus[k], __dsub_rn(fy, __dmul_rn(ca.ajm1[k], gg))));
    };
#define ESQ_RKC3D_LOAD(IT)                                                  \
    {                                                                       \
        const int i_ = ibase + (IT);                                        \
        _Pragma("unroll") for (int r = 0; r < JT; ++r) {                    \
            if constexpr (FIRST) {                                          \
                py[r] = ld(ry, i_ + 1, r);                                  \
                pf[r] = ld(rf, i_ + 1, r);                                  \
            } else {                                                        \
                pa[r] = ld(ra, i_ + 1, r);                                  \
                pb[r] = ld(rb, i_, r);                                      \
                py[r] = ld(ry, i_, r);                                      \
                pf[r] = ld(rf, i_, r);                                      \
            }                                                               \
        }                                                                   \
    }
    // TWO forms of the marching loop, by tile shape (same arithmetic, same results):
    //  * wide tiles (D JT >= 16: D = 5 on four rows per thread sits at 249 of 256
    //    registers): ONE arriving plane handed from stage to stage, each window moving
    //    on as soon as its stage is done;
    //  * narrow tiles: an arriving plane per stage, all windows moving on at the end of
    //    the iteration -- rkc_chain4 on sixteen waves of two rows 49.9 -> 46.2 us, SSV2stab
    //    at N = 159 1.315 -> 1.235 ms/step (same-box A/B, profiles/r05_experiments.md §2);
    //    the same form on the wide tiles spills (N = 400: 19.0 -> 29.4 ms/step).
    // ESQ_RKC3D_EARLY=0|1 at compile time forces one form (tuning builds).
    constexpr bool kEarly = ESQ_RKC3D_EARLY >= 0 ? ESQ_RKC3D_EARLY != 0 : D * JT >= 16;
    if constexpr (kEarly) {
        // windows: wm[k], wc[k] = Y_k at planes (centre - 1, centre) of stage k;
        // dy[k], df[k] = y_n, f_n at stage k's centre plane (a delay line)
        double wm[D][JT], wc[D][JT], dy[D][JT], df[D][JT];
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int r = 0; r < JT; ++r) wm[k][r] = wc[k][r] = dy[k][r] = df[k][r] = 0.0;
        const int ibase = i_lo - (D - 1);                 // stage 0's first centre plane
        auto first = [&](double y, double f) -> double {          // k_rkc_first's rounding
            return __dadd_rn(y, __dmul_rn(ca.hmus1, f));
        };
        // FIRST: y_n, f_n at stage 0's centre plane (loaded as the plane AFTER the
        // centre one iteration earlier)
        double cy[JT], cf[JT];
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            if constexpr (FIRST) {
                wm[0][r] = first(ld(ry, ibase - 1, r), ld(rf, ibase - 1, r));
                cy[r] = ld(ry, ibase, r);
                cf[r] = ld(rf, ibase, r);
                wc[0][r] = first(cy[r], cf[r]);
            } else {
                wm[0][r] = ld(ra, ibase - 1, r);
                wc[0][r] = ld(ra, ibase, r);
                cy[r] = cf[r] = 0.0;
            }
        }
        // operands of stage 0's plane, requested ONE ITERATION AHEAD (beyond the last
        // iteration: one more plane is requested and never used)
        double pa[JT], pb[JT], py[JT], pf[JT];
        ESQ_RKC3D_LOAD(0)
        const int iters = Re + 2 * (D - 1);
        for (int it = 0; it < iters; ++it) {
            const int i0 = ibase + it;
            double wp[JT], ykm1[JT];
#pragma unroll
            for (int r = 0; r < JT; ++r) {
                if constexpr (FIRST) {
                    wp[r] = first(py[r], pf[r]);
                    ykm1[r] = cy[r];
                    dy[0][r] = cy[r];
                    df[0][r] = cf[r];
                    cy[r] = py[r];
                    cf[r] = pf[r];
                } else {
                    wp[r] = pa[r];
                    ykm1[r] = pb[r];
                    dy[0][r] = py[r];
                    df[0][r] = pf[r];
                }
            }
            ESQ_RKC3D_LOAD(it + 1)
            // the slices' edge rows of all D centre planes change hands
#pragma unroll
            for (int k = 0; k < D; ++k) {
                xch[it & 1][k][w + 1][0][lane] = wc[k][0];
                xch[it & 1][k][w + 1][1][lane] = wc[k][JT - 1];
            }
            __syncthreads();
            double eu[D], ed[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                eu[k] = xch[it & 1][k][w][1][lane];
                ed[k] = xch[it & 1][k][w + 2][0][lane];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double nw[JT];
                if (it >= 2 * k) {                                   // wave-uniform
                    const int ik = i0 - k;
                    const bool pl_ok = ik >= 0 && ik < N;
#pragma unroll
                    for (int r = 0; r < JT; ++r) {
                        const double up = r > 0 ? wc[k][r > 0 ? r - 1 : 0] : eu[k];
                        const double dn = r < JT - 1 ? wc[k][r < JT - 1 ? r + 1 : r] : ed[k];
                        const double lf = lane_left(wc[k][r]);
                        const double rt = lane_right(wc[k][r]);
                        const double fy = st.eval(wm[k][r], wp[r], up, dn, lf, rt, wc[k][r]);
                        if (LAST && k == D - 1) {
                            nw[r] = (pl_ok && in[r]) ? fy : 0.0;
                            if (so[r] != 0xffffffffu) {              // a point this tile stores
                                const double er =
                                    __dadd_rn(__dmul_rn(0.8, __dsub_rn(dy[k][r], wc[k][r])),
                                              __dmul_rn(ca.h04, __dadd_rn(df[k][r], fy)));
                                const size_t e = (size_t)ik * (size_t)N * (size_t)N + so[r] / 8u;
                                local += ratio_sq1(er, wc[k][r], dy[k][r], ca.red.atol_vec,
                                                   ca.red.atol_s, ca.red.rtol, e, ca.red.n_valid);
                            }
                        } else {
                            const double v = one(k, wc[k][r], ykm1[r], dy[k][r], df[k][r], fy);
                            nw[r] = (pl_ok && in[r]) ? v : 0.0;
                        }
                    }
                    if (k == D - 1) {
                        // the last stage is only ever at planes [i_lo, i_hi)
                        const unsigned pl = (unsigned)ik * plane_bytes;
#pragma unroll
                        for (int r = 0; r < JT; ++r) {
                            buf_st(ro, so[r], pl, nw[r]);
                            buf_st(rp, so[r], pl, wc[k][r]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < JT; ++r) nw[r] = 0.0;
                }
                // Y_k's window moves one plane on; its old lower plane is Y_{(k+1)-1}
                // at stage k + 1's centre plane, the fresh values are that stage's
                // upper plane
#pragma unroll
                for (int r = 0; r < JT; ++r) {
                    const double old = wm[k][r];
                    wm[k][r] = wc[k][r];
                    wc[k][r] = wp[r];
                    ykm1[r] = old;
                    wp[r] = nw[r];
                }
            }
#pragma unroll
            for (int k = D - 1; k >= 1; --k)
#pragma unroll
                for (int r = 0; r < JT; ++r) {
                    dy[k][r] = dy[k - 1][r];
                    df[k][r] = df[k - 1][r];
                }
        }
    } else {
        // windows: W[k][0], W[k][1] = Y_k at planes (centre - 1, centre) of stage k,
        // W[k][2] the plane that arrives in this iteration -- one per stage; ALL windows
        // move one plane on at the END of the iteration
        double W[D][3][JT], dy[D][JT], df[D][JT];
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int r = 0; r < JT; ++r) {
                W[k][0][r] = W[k][1][r] = W[k][2][r] = 0.0;
                dy[k][r] = df[k][r] = 0.0;
            }
        const int ibase = i_lo - (D - 1);                 // stage 0's first centre plane
        auto first = [&](double y, double f) -> double {          // k_rkc_first's rounding
            return __dadd_rn(y, __dmul_rn(ca.hmus1, f));
        };
        // FIRST: y_n, f_n at stage 0's centre plane (loaded as the plane AFTER the
        // centre one iteration earlier)
        double cy[JT], cf[JT];
#pragma unroll
        for (int r = 0; r < JT; ++r) {
            if constexpr (FIRST) {
                W[0][0][r] = first(ld(ry, ibase - 1, r), ld(rf, ibase - 1, r));
                cy[r] = ld(ry, ibase, r);
                cf[r] = ld(rf, ibase, r);
                W[0][1][r] = first(cy[r], cf[r]);
            } else {
                W[0][0][r] = ld(ra, ibase - 1, r);
                W[0][1][r] = ld(ra, ibase, r);
                cy[r] = cf[r] = 0.0;
            }
        }
        // operands of stage 0's plane, requested ONE ITERATION AHEAD (beyond the last
        // iteration: one more plane is requested and never used)
        double pa[JT], pb[JT], py[JT], pf[JT];
        ESQ_RKC3D_LOAD(0)
        const int iters = Re + 2 * (D - 1);
        auto body = [&](const int it) {
            constexpr int LO = 0, CE = 1, UP = 2;
            const int i0 = ibase + it;
            double ykm1[JT];
#pragma unroll
            for (int r = 0; r < JT; ++r) {
                if constexpr (FIRST) {
                    W[0][UP][r] = first(py[r], pf[r]);
                    ykm1[r] = cy[r];
                    dy[0][r] = cy[r];
                    df[0][r] = cf[r];
                    cy[r] = py[r];
                    cf[r] = pf[r];
                } else {
                    W[0][UP][r] = pa[r];
                    ykm1[r] = pb[r];
                    dy[0][r] = py[r];
                    df[0][r] = pf[r];
                }
            }
            ESQ_RKC3D_LOAD(it + 1)
            // the slices' edge rows of all D centre planes change hands
#pragma unroll
            for (int k = 0; k < D; ++k) {
                xch[it & 1][k][w + 1][0][lane] = W[k][CE][0];
                xch[it & 1][k][w + 1][1][lane] = W[k][CE][JT - 1];
            }
            __syncthreads();
            double eu[D], ed[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                eu[k] = xch[it & 1][k][w][1][lane];
                ed[k] = xch[it & 1][k][w + 2][0][lane];
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                double nw[JT];
                if (it >= 2 * k) {                                   // wave-uniform
                    const int ik = i0 - k;
                    const bool pl_ok = ik >= 0 && ik < N;
#pragma unroll
                    for (int r = 0; r < JT; ++r) {
                        const double wcr = W[k][CE][r];
                        const double up = r > 0 ? W[k][CE][r > 0 ? r - 1 : 0] : eu[k];
                        const double dn = r < JT - 1 ? W[k][CE][r < JT - 1 ? r + 1 : r] : ed[k];
                        const double lf = lane_left(wcr);
                        const double rt = lane_right(wcr);
                        const double fy = st.eval(W[k][LO][r], W[k][UP][r], up, dn, lf, rt, wcr);
                        const double dyk = dy[k][r], dfk = df[k][r];
                        if (LAST && k == D - 1) {
                            nw[r] = (pl_ok && in[r]) ? fy : 0.0;
                            if (so[r] != 0xffffffffu) {              // a point this tile stores
                                const double er =
                                    __dadd_rn(__dmul_rn(0.8, __dsub_rn(dyk, wcr)),
                                              __dmul_rn(ca.h04, __dadd_rn(dfk, fy)));
                                const size_t e = (size_t)ik * (size_t)N * (size_t)N + so[r] / 8u;
                                local += ratio_sq1(er, wcr, dyk, ca.red.atol_vec,
                                                   ca.red.atol_s, ca.red.rtol, e, ca.red.n_valid);
                            }
                        } else {
                            const double v = one(k, wcr, ykm1[r], dyk, dfk, fy);
                            nw[r] = (pl_ok && in[r]) ? v : 0.0;
                        }
                    }
                    if (k == D - 1) {
                        // the last stage is only ever at planes [i_lo, i_hi)
                        const unsigned pl = (unsigned)ik * plane_bytes;
#pragma unroll
                        for (int r = 0; r < JT; ++r) {
                            buf_st(ro, so[r], pl, nw[r]);
                            buf_st(rp, so[r], pl, W[k][CE][r]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < JT; ++r) nw[r] = 0.0;
                }
                // Y_k's old lower plane is Y_{(k+1)-1} at stage k + 1's centre plane, the
                // fresh values are that stage's arriving plane
#pragma unroll
                for (int r = 0; r < JT; ++r) {
                    ykm1[r] = W[k][LO][r];
                    if (k + 1 < D) W[k + 1 < D ? k + 1 : k][UP][r] = nw[r];
                }
            }
            // every window moves one plane on, every delay line one stage
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int r = 0; r < JT; ++r) {
                    W[k][0][r] = W[k][1][r];
                    W[k][1][r] = W[k][2][r];
                }
#pragma unroll
            for (int k = D - 1; k >= 1; --k)
#pragma unroll
                for (int r = 0; r < JT; ++r) {
                    dy[k][r] = dy[k - 1][r];
                    df[k][r] = df[k - 1][r];
                }
        };
        for (int it = 0; it < iters; ++it) body(it);
    }
#undef ESQ_RKC3D_LOAD
    if constexpr (LAST) block_partial_w<NW>(local, ca.red.partials);
}

}  // namespace esq
