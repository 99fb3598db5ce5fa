// esq_pair.hpp -- TWO consecutive Runge-Kutta stages in ONE marching sweep of a
// 2-D five-point stencil plugin (esq_rhs_pair_fn, include/extensisq_amd.h).
//
// What it removes.  The one-stage sweeps (esq_epilogue.hpp) write the argument
// of stage i+1,  ys_{i+1} = y + h*(init + sum_j a_{i+1,j} K_j),  to memory and
// the next sweep reads it back; both sweeps read y and mostly the same K rows.
// Measured (profiles/r02_experiments.md): every sweep runs at the memory side's
// request rate, where a written byte costs two read bytes -- so the only lever
// left is to move fewer bytes.  Here stage i is evaluated at grid row r and
// stage i+1 ONE ROW BEHIND it:
//
//     row r   :  K_i[r]     = f(ys_i[r-1..r+1])              (stage A)
//                ys_{i+1}[r] = y[r] + h*(init_a + sum ca_u U_u[r] + ca_self K_i[r])
//                                                           -> registers only
//     row r-1 :  K_{i+1}[r-1] = f(ys_{i+1}[r-2..r])          (stage B)
//                out[r-1]     = y[r-1] + h*(init_b + sum cb_u U_u[r-1]
//                                        + cb_prev K_i[r-1] + cb_self K_{i+1}[r-1])
//
// ys_{i+1} never touches memory, y and the K rows U both sums share are read
// once.  B's FMA chain runs over ascending column index like every other kernel
// here; everything but its last term (the fresh K_{i+1}) is known when row r-1
// is stage A's row, so the chain's prefix is carried one row in registers
// instead of the operands: results are bit-identical to two one-stage sweeps.
//
// Geometry.  One wave owns a tile of R rows x 62 column pairs (+ one halo pair
// each side: the lanes 0 and 63 evaluate stage A only, so that stage B's left
// and right neighbours come from wave shuffles and NO single-lane fix-up loads
// exist) and marches down it; stage A also runs on one halo row above and one
// below the tile.  Overhead: 2/R rows and 2/64 lanes of stage A.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_terms.hpp"

namespace esq {

constexpr int kPairCols = 62;          // stage-B pairs per wave tile

// device-side description of a pair (built from `esq_pair` by make_pair_args)
template <int NU>
struct PairArgs {
    static constexpr int kNU = NU;
    const double *rows[NU > 0 ? NU : 1];   // K rows read from memory (both sums)
    double ca[NU > 0 ? NU : 1];            // weight in stage A's sum (0: not in it)
    double cb[NU > 0 ? NU : 1];            // weight in stage B's sum / y_new
    double eb[NU > 0 ? NU : 1];            // SOLERR: error weight
    unsigned b_mask;                       // rows that take part in B's chain(s)
    int prev_in_b;                         // K_i takes part in B's chain(s)
    const double *init_a, *init_b, *y;
    double ca_self, cb_prev, cb_self, eb_prev, eb_self, h;
    double *out;                           // B's output vector
    int store_fa, f_nt;
    RedArgs red;
};

// Dirichlet / periodic index helpers of one wave tile
template <bool PERIODIC>
struct PairGeo {
    int N, npairs, lane, pc, pw;
    bool indom, live, store_ok;
    __device__ __forceinline__ void init(int N_, unsigned ct) {
        N = N_;
        npairs = N_ / 2;
        lane = threadIdx.x & 63;
        pc = kPairCols * (int)ct - 1 + lane;
        indom = pc >= 0 && pc < npairs;
        live = PERIODIC ? (pc >= -1 && pc <= npairs) : indom;
        pw = PERIODIC ? (pc < 0 ? pc + npairs : (pc >= npairs ? pc - npairs : pc)) : pc;
        store_ok = indom && lane >= 1 && lane <= kPairCols;
    }
    __device__ __forceinline__ bool row_ok(int r) const {
        return PERIODIC || (r >= 0 && r < N);
    }
    __device__ __forceinline__ int wrap(int r) const {
        return PERIODIC ? (r < 0 ? r + N : (r >= N ? r - N : r)) : r;
    }
};

// NF fields of N x N (state = field 0, field 1, ... one after the other);
// Fn::eval(centres, laplacians) -> derivatives, all per column pair.
// KINDB: ESQ_EPI_STAGE (B forms a stage argument / y_new of an FSAL pair) or
// ESQ_EPI_SOLERR (B forms y_new and the error partial sums).
template <int NF, bool PERIODIC, int NU, int KINDB, class Fn>
__global__ __launch_bounds__(kBlock) void k_pair2d(
    const double *__restrict__ ys, double *__restrict__ fa, double *__restrict__ fb,
    PairArgs<NU> pa, Fn fn, int N, int R, unsigned tpr, unsigned ntiles,
    unsigned nblocks, unsigned xcd) {
    // XCD band remap as in the one-stage sweeps: XCD x takes a contiguous band
    const unsigned per = (nblocks + xcd - 1) / xcd;
    const unsigned lb = (blockIdx.x % xcd) * per + blockIdx.x / xcd;
    const unsigned tile = lb * (kBlock / 64) + (threadIdx.x >> 6);
    double local = 0.0;
    if (lb < nblocks && tile < ntiles) {                        // wave-uniform
        PairGeo<PERIODIC> g;
        g.init(N, tile % tpr);
        const int r0 = (int)(tile / tpr) * R;
        const int Re = (N - r0) < R ? (N - r0) : R;
        const size_t fstride = (size_t)N * (size_t)g.npairs;   // pairs per field
        double2 a_m[NF], a_c[NF], a_p[NF], b_m[NF], b_c[NF], b_p[NF];
        double2 pb_prev[NF], pe_prev[NF], y_prev[NF];
        auto ld_ys = [&](int r, int f) -> double2 {
            if (!g.live || !g.row_ok(r)) return make_double2(0.0, 0.0);
            return ld2(ys, (size_t)f * fstride + (size_t)g.wrap(r) * g.npairs + g.pw);
        };
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            a_m[f] = ld_ys(r0 - 2, f);
            a_c[f] = ld_ys(r0 - 1, f);
            b_m[f] = b_c[f] = make_double2(0.0, 0.0);
            pb_prev[f] = pe_prev[f] = y_prev[f] = make_double2(0.0, 0.0);
        }
        for (int it = 0; it < Re + 2; ++it) {
            const int r = r0 - 1 + it;                     // stage A's row
            const bool act = g.live && g.row_ok(r);        // A computes here
            const size_t base = (size_t)g.wrap(r) * g.npairs + g.pw;
            // ---- every load of the iteration before the first use
            double2 u[NU > 0 ? NU : 1][NF], yc[NF], ia[NF], ib[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                a_p[f] = ld_ys(r + 1, f);
                const size_t k2 = (size_t)f * fstride + base;
                const double2 z = make_double2(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < NU; ++j) u[j][f] = act ? ld2_nt(pa.rows[j], k2) : z;
                yc[f] = act ? ld2(pa.y, k2) : z;
                ia[f] = (act && pa.init_a) ? ld2_nt(pa.init_a, k2) : z;
                ib[f] = (act && pa.init_b) ? ld2_nt(pa.init_b, k2) : z;
            }
            // ---- stage A at row r
            double2 cA[NF], lapA[NF], fA[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const double lf = __shfl_up(a_c[f].y, 1, 64);
                const double rt = __shfl_down(a_c[f].x, 1, 64);
                cA[f] = a_c[f];
                lapA[f].x = ((a_m[f].x + a_p[f].x) + (lf + a_c[f].y)) - 4.0 * a_c[f].x;
                lapA[f].y = ((a_m[f].y + a_p[f].y) + (a_c[f].x + rt)) - 4.0 * a_c[f].y;
            }
            fn.eval(cA, lapA, fA);
            const bool own_row = it >= 1 && it <= Re;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const size_t k2 = (size_t)f * fstride + base;
                if (pa.store_fa && own_row && g.store_ok) {
                    if (pa.f_nt) st2_nt(fa, k2, fA[f]); else st2(fa, k2, fA[f]);
                }
                // argument of stage B at row r (registers only)
                double2 acc = ia[f];
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    if (pa.ca[j] != 0.0) {                     // uniform
                        acc.x = fma(pa.ca[j], u[j][f].x, acc.x);
                        acc.y = fma(pa.ca[j], u[j][f].y, acc.y);
                    }
                }
                if (pa.ca_self != 0.0) {
                    acc.x = fma(pa.ca_self, fA[f].x, acc.x);
                    acc.y = fma(pa.ca_self, fA[f].y, acc.y);
                }
                const double2 ysn = make_double2(__dadd_rn(yc[f].x, __dmul_rn(pa.h, acc.x)),
                                                 __dadd_rn(yc[f].y, __dmul_rn(pa.h, acc.y)));
                b_p[f] = act ? ysn : make_double2(0.0, 0.0);
            }
            // ---- stage B at row r - 1 (its window is complete now)
            if (it >= 2) {                                     // wave-uniform
                double2 cB[NF], lapB[NF], fB[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const double lf = __shfl_up(b_c[f].y, 1, 64);
                    const double rt = __shfl_down(b_c[f].x, 1, 64);
                    cB[f] = b_c[f];
                    lapB[f].x = ((b_m[f].x + b_p[f].x) + (lf + b_c[f].y)) - 4.0 * b_c[f].x;
                    lapB[f].y = ((b_m[f].y + b_p[f].y) + (b_c[f].x + rt)) - 4.0 * b_c[f].y;
                }
                fn.eval(cB, lapB, fB);
                if (g.store_ok) {
                    const size_t bbase = (size_t)g.wrap(r - 1) * g.npairs + g.pw;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const size_t k2 = (size_t)f * fstride + bbase;
                        if (pa.f_nt) st2_nt(fb, k2, fB[f]); else st2(fb, k2, fB[f]);
                        double2 sb = pb_prev[f];
                        if (KINDB == ESQ_EPI_SOLERR) {
                            // the fresh row is part of the union of the supports: a
                            // zero weight contributes fma(0, v, s), as in EpiSolErr
                            double2 se = pe_prev[f];
                            sb.x = fma(pa.cb_self, fB[f].x, sb.x);
                            sb.y = fma(pa.cb_self, fB[f].y, sb.y);
                            se.x = fma(pa.eb_self, fB[f].x, se.x);
                            se.y = fma(pa.eb_self, fB[f].y, se.y);
                            double2 yn, er;
                            yn.x = __dadd_rn(y_prev[f].x, __dmul_rn(pa.h, sb.x));
                            yn.y = __dadd_rn(y_prev[f].y, __dmul_rn(pa.h, sb.y));
                            er.x = __dmul_rn(pa.h, se.x);
                            er.y = __dmul_rn(pa.h, se.y);
                            st2(pa.out, k2, yn);
                            local += ratio_sq<false>(er, y_prev[f], yn, pa.red.atol_vec,
                                                     pa.red.atol_s, pa.red.rtol, k2,
                                                     pa.red.n_valid);
                        } else {
                            if (pa.cb_self != 0.0) {
                                sb.x = fma(pa.cb_self, fB[f].x, sb.x);
                                sb.y = fma(pa.cb_self, fB[f].y, sb.y);
                            }
                            st2(pa.out, k2,
                                make_double2(__dadd_rn(y_prev[f].x, __dmul_rn(pa.h, sb.x)),
                                             __dadd_rn(y_prev[f].y, __dmul_rn(pa.h, sb.y))));
                        }
                    }
                }
            }
            // ---- prefix of B's chain(s) for row r (all terms but the fresh K_{i+1})
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                double2 sb = ib[f], se = make_double2(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    if ((pa.b_mask >> j) & 1u) {               // uniform
                        sb.x = fma(pa.cb[j], u[j][f].x, sb.x);
                        sb.y = fma(pa.cb[j], u[j][f].y, sb.y);
                        if (KINDB == ESQ_EPI_SOLERR) {
                            se.x = fma(pa.eb[j], u[j][f].x, se.x);
                            se.y = fma(pa.eb[j], u[j][f].y, se.y);
                        }
                    }
                }
                if (pa.prev_in_b) {
                    sb.x = fma(pa.cb_prev, fA[f].x, sb.x);
                    sb.y = fma(pa.cb_prev, fA[f].y, sb.y);
                    if (KINDB == ESQ_EPI_SOLERR) {
                        se.x = fma(pa.eb_prev, fA[f].x, se.x);
                        se.y = fma(pa.eb_prev, fA[f].y, se.y);
                    }
                }
                pb_prev[f] = sb;
                pe_prev[f] = se;
                y_prev[f] = yc[f];
                a_m[f] = a_c[f];
                a_c[f] = a_p[f];
                b_m[f] = b_c[f];
                b_c[f] = b_p[f];
            }
        }
    }
    if (KINDB == ESQ_EPI_SOLERR) block_partial(local, pa.red.partials);
}

}  // namespace esq
