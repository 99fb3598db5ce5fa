// esq_chain.hpp -- D consecutive Runge-Kutta stages in ONE marching sweep of a
// 2-D five-point stencil plugin (esq_rhs_chain_fn, include/extensisq_amd.h).
//
// What it removes.  The one-stage sweeps (esq_epilogue.hpp) write the argument
// of the next stage,  ys = y + h*(init + sum_j a_j K_j),  to memory and the next
// sweep reads it back; consecutive sweeps read y and mostly the same K rows.
// Measured (profiles/r02_experiments.md): every sweep runs at the memory side's
// request rate, where a written byte costs two read bytes -- so the only lever
// left is to move fewer bytes.  Here stage k of the chain runs k grid rows
// behind stage 0:
//
//   iteration it, stage k at row  rho_k = rho_0 - k :
//       K_k[rho_k]     = f(T_k[rho_k - 1 .. rho_k + 1])           (T_0 = the input)
//       T_e[rho_k]    += c_{e,k} K_k[rho_k]     for every later target e > k
//       T_{k+1}[rho_k] is complete now: it enters stage k+1's window (registers)
//                      or, for the last stage, is stored (next argument / y_new)
//
// Target e's sum for a row starts when stage 0 visits the row (leading partial
// sum `init`, the K rows read from memory -- ONCE for all D targets) and takes
// the chain's own derivatives as they appear, one per iteration, in ascending
// column order: the same FMA chain as every other kernel here, so K rows and
// states are bit-identical to D one-stage sweeps.  The intermediate arguments
// never touch memory; y is read once per chain.
//
// Geometry.  One wave owns a tile of R rows x (64 - 2(D-1)) column pairs; lane l
// holds pair  W*ct - (D-1) + l.  Stage k is valid on lanes [k, 63-k] and on
// D-1-k halo rows above and below the tile, so every left/right neighbour comes
// from a wave shuffle and NO single-lane fix-up loads exist.  Overhead: 2(D-1)/R
// rows and 2(D-1)/64 lanes of loads and stage evaluations.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_terms.hpp"

namespace esq {

// device-side description of a chain (built from `esq_chain` by make_chain_args)
template <int D, int NU>
struct ChainArgs {
    static constexpr int kD = D, kNU = NU;
    static constexpr int NUa = NU > 0 ? NU : 1;
    const double *rows[NUa];         // K rows read from memory (union over targets)
    double cu[D][NUa];               // cu[e][u]: weight of rows[u] in target e + 1
    double eu[NUa];                  // SOLERR: error weights of the last target
    unsigned umask[D];               // rows that take part in target e + 1's chain
    double ck[D][D];                 // ck[e][k]: weight of K_k in target e + 1 (k <= e)
    double ek[D];                    // SOLERR: error weight of K_k
    unsigned kmask[D];               // stages whose K takes part in target e + 1
    const double *init[D];           // leading partial sum of target e + 1 or nullptr
    const double *y;                 // base state; nullptr: the chain's own input
    double h;
    double *fk[D];                   // where K_k goes (nullptr: not stored)
    double *out;                     // last target
    int f_nt;
    RedArgs red;
};

// NF fields of N x N (state = field 0, field 1, ... one after the other);
// Fn::eval(centres, laplacians) -> derivatives, all per column pair.
// KINDLAST: ESQ_EPI_STAGE (the last target is a stage argument / y_new of an
// FSAL pair) or ESQ_EPI_SOLERR (y_new and the error partial sums).
template <int NF, bool PERIODIC, int D, int NU, int KINDLAST, class Fn>
__global__ __launch_bounds__(kBlock) void k_chain2d(
    const double *__restrict__ ys, ChainArgs<D, NU> ca, Fn fn, int N, int R,
    unsigned tpr, unsigned ntiles, unsigned nblocks, unsigned xcd) {
    constexpr int H = D - 1;                       // halo rows / lanes per side
    constexpr int W = 64 - 2 * H;                  // last-stage pairs per tile
    constexpr bool SOLERR = KINDLAST == ESQ_EPI_SOLERR;
    // XCD band remap as in the one-stage sweeps: XCD x takes a contiguous band
    const unsigned per = (nblocks + xcd - 1) / xcd;
    const unsigned lb = (blockIdx.x % xcd) * per + blockIdx.x / xcd;
    const unsigned tile = lb * (kBlock / 64) + (threadIdx.x >> 6);
    double local = 0.0;
    if (lb < nblocks && tile < ntiles) {                        // wave-uniform
        const int npairs = N / 2;
        const int lane = threadIdx.x & 63;
        const int pc = W * (int)(tile % tpr) - H + lane;
        const bool indom = pc >= 0 && pc < npairs;
        const bool live = PERIODIC ? (pc >= -H && pc < npairs + H) : indom;
        const int pw = PERIODIC ? (pc < 0 ? pc + npairs : (pc >= npairs ? pc - npairs : pc))
                                : pc;
        const bool store_ok = indom && lane >= H && lane < 64 - H;
        const int r0 = (int)(tile / tpr) * R;
        const int Re = (N - r0) < R ? (N - r0) : R;
        const size_t fstride = (size_t)N * (size_t)npairs;     // pairs per field
        auto row_ok = [&](int r) { return PERIODIC || (r >= 0 && r < N); };
        auto wrap = [&](int r) {
            return PERIODIC ? (r < 0 ? r + N : (r >= N ? r - N : r)) : r;
        };
        auto ld_ys = [&](int r, int f) -> double2 {
            if (!live || !row_ok(r)) return make_double2(0.0, 0.0);
            return ld2(ys, (size_t)f * fstride + (size_t)wrap(r) * npairs + pw);
        };
        const double2 zero = make_double2(0.0, 0.0);
        // windows of the D stages: rows rho_k - 1, rho_k, rho_k + 1 of T_k
        double2 wm[D][NF], wc[D][NF], wp[D][NF];
        // acc[e][k]: target e + 1's sum for the row stage k is at (k <= e)
        double2 acc[D][D][NF], acce[D][NF], yf[D][NF];
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                wm[k][f] = wc[k][f] = wp[k][f] = zero;
                yf[k][f] = acce[k][f] = zero;
#pragma unroll
                for (int e = 0; e < D; ++e) acc[e][k][f] = zero;
            }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            wm[0][f] = ld_ys(r0 - H - 1, f);
            wc[0][f] = ld_ys(r0 - H, f);
        }
        const int iters = Re + 2 * H;
        for (int it = 0; it < iters; ++it) {
            const int rho0 = r0 - H + it;                  // stage 0's row
            const bool act0 = live && row_ok(rho0);
            const size_t base0 = (size_t)wrap(rho0) * npairs + pw;
            // ---- every load of the iteration before the first use
            double2 u[ChainArgs<D, NU>::NUa][NF], in[D][NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                wp[0][f] = ld_ys(rho0 + 1, f);
                const size_t k2 = (size_t)f * fstride + base0;
#pragma unroll
                for (int j = 0; j < NU; ++j) u[j][f] = act0 ? ld2_nt(ca.rows[j], k2) : zero;
                yf[0][f] = ca.y ? (act0 ? ld2(ca.y, k2) : zero) : wc[0][f];
#pragma unroll
                for (int e = 0; e < D; ++e)
                    in[e][f] = (act0 && ca.init[e]) ? ld2_nt(ca.init[e], k2) : zero;
            }
            // ---- the D targets' sums for row rho0: leading partial + memory rows
#pragma unroll
            for (int e = 0; e < D; ++e)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    double2 s = in[e][f], se = zero;
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        if ((ca.umask[e] >> j) & 1u) {             // uniform
                            s.x = fma(ca.cu[e][j], u[j][f].x, s.x);
                            s.y = fma(ca.cu[e][j], u[j][f].y, s.y);
                            if (SOLERR && e == D - 1) {
                                se.x = fma(ca.eu[j], u[j][f].x, se.x);
                                se.y = fma(ca.eu[j], u[j][f].y, se.y);
                            }
                        }
                    }
                    acc[e][0][f] = s;
                    if (SOLERR && e == D - 1) acce[0][f] = se;
                }
            // ---- the stages, each one row behind its predecessor
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (it >= 2 * k) {                                 // wave-uniform
                    const int rho = rho0 - k;
                    const bool actk = live && row_ok(rho);
                    const bool own = rho >= r0 && rho < r0 + Re;
                    const size_t basek = (size_t)wrap(rho) * npairs + pw;
                    double2 cc[NF], lap[NF], fK[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const double lf = __shfl_up(wc[k][f].y, 1, 64);
                        const double rt = __shfl_down(wc[k][f].x, 1, 64);
                        cc[f] = wc[k][f];
                        lap[f].x = ((wm[k][f].x + wp[k][f].x) + (lf + wc[k][f].y)) -
                                   4.0 * wc[k][f].x;
                        lap[f].y = ((wm[k][f].y + wp[k][f].y) + (wc[k][f].x + rt)) -
                                   4.0 * wc[k][f].y;
                    }
                    fn.eval(cc, lap, fK);
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const size_t k2 = (size_t)f * fstride + basek;
                        if (ca.fk[k] && own && store_ok) {
                            if (ca.f_nt) st2_nt(ca.fk[k], k2, fK[f]);
                            else st2(ca.fk[k], k2, fK[f]);
                        }
                        // K_k enters the sums of the later targets
#pragma unroll
                        for (int e = k; e < D; ++e) {
                            if ((ca.kmask[e] >> k) & 1u) {         // uniform
                                acc[e][k][f].x = fma(ca.ck[e][k], fK[f].x, acc[e][k][f].x);
                                acc[e][k][f].y = fma(ca.ck[e][k], fK[f].y, acc[e][k][f].y);
                                if (SOLERR && e == D - 1) {
                                    acce[k][f].x = fma(ca.ek[k], fK[f].x, acce[k][f].x);
                                    acce[k][f].y = fma(ca.ek[k], fK[f].y, acce[k][f].y);
                                }
                            }
                        }
                        // target k + 1 is complete for this row
                        const double2 s = acc[k][k][f];
                        const double2 t =
                            make_double2(__dadd_rn(yf[k][f].x, __dmul_rn(ca.h, s.x)),
                                         __dadd_rn(yf[k][f].y, __dmul_rn(ca.h, s.y)));
                        if (k + 1 < D) {
                            wp[k + 1 < D ? k + 1 : k][f] = actk ? t : zero;
                        } else if (own && store_ok) {
                            st2(ca.out, k2, t);
                            if (SOLERR) {
                                const double2 er = make_double2(__dmul_rn(ca.h, acce[k][f].x),
                                                                __dmul_rn(ca.h, acce[k][f].y));
                                local += ratio_sq<false>(er, yf[k][f], t, ca.red.atol_vec,
                                                         ca.red.atol_s, ca.red.rtol, k2,
                                                         ca.red.n_valid);
                            }
                        }
                    }
                }
            }
            // ---- every row moves one stage on
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int k = D - 1; k >= 1; --k) {
                    yf[k][f] = yf[k - 1][f];
                    acce[k][f] = acce[k - 1][f];
#pragma unroll
                    for (int e = k; e < D; ++e) acc[e][k][f] = acc[e][k - 1][f];
                }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    wm[k][f] = wc[k][f];
                    wc[k][f] = wp[k][f];
                }
            }
        }
    }
    if (SOLERR) block_partial(local, ca.red.partials);
}

}  // namespace esq
