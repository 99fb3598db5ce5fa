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
// A row (or a chain member) that takes no part in a target enters it with weight
// +0.0 (round 6; the 3-D sweeps since round 5): fma(0, v, s) == s for every finite v,
// so the values are those of the skipping sums bit for bit, and a non-finite v
// poisons the target exactly as NumPy's K[:i].T @ A[i, :i] over ALL rows does
// (common.py:355).  Until round 5 every FMA pair sat behind a participation test
// (one s_bitcmp + a real branch, i.e. its own basic block): without them the
// Brusselator's whole-step Ts5 chain runs in 156 instead of 263 us, Pr8's heat chains
// 11 % faster, the metric's 1-3 % (profiles/r06_experiments.md, section 6).
//
// Target e's sum for a row starts when stage 0 visits the row (the K rows read
// from memory -- ONCE for all D targets; a leading partial sum of the blocked
// accumulation is simply the first such row, with weight 1) and takes
// the chain's own derivatives as they appear, one per iteration, in ascending
// column order: the same FMA chain as every other kernel here, so K rows and
// states are bit-identical to D one-stage sweeps.  The intermediate arguments
// never touch memory; y is read once per chain.
//
// Geometry.  One wave owns a tile of R rows x (64 - 2(D-1)) column pairs; lane l
// holds pair  W*ct - (D-1) + l.  Stage k is valid on lanes [k, 63-k], so every
// left/right neighbour comes from a wave shuffle and NO single-lane fix-up loads
// exist.  Vertically, tiles come in DIVERGING PAIRS (round 6): two tiles of one
// workgroup start at the row boundary they share -- the upper one marches up, the
// lower one down -- and hand each other the ONE value per stage the neighbour's
// stencil reaches across that boundary (T_k of their first row, through LDS, in
// the first D iterations), so a tile has no run-in rows at all; at its far end
// stage k walks D-1-k halo rows of the next pair (which arrives there at the same
// moment from the other side: the shared rows meet in L2).  Overhead: (D-1)/R rows
// -- half of the independent tiles' 2(D-1)/R, and D-1 marching iterations per tile
// less, which is what counts on the short tiles of latency-bound grids (Ts5 at
// N = 1000: 10 instead of 15 iterations) -- and 2(D-1)/64 lanes of loads and stage
// evaluations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_options.hpp"
#include "esq_terms.hpp"

namespace esq {

// Register budget of the two-field (Brusselator) instantiations, from
// `hipcc -Rpass-analysis=kernel-resource-usage`: the largest number of memory
// rows per (depth, last kind) that still leaves two waves per SIMD.  Wider
// chains drop to one wave per SIMD and lose more than they save (measured:
// chain3+solerr<6> at one wave per SIMD took 590 us where a two-wave chain2 +
// a single sweep take 360).  The block planner (esq_step.hip) uses the same
// table.
struct ChainCaps {
    int stage[7], solerr[7];
};
// Tile height of a plugin object's chain sweeps: forced (tests: small tiles put the halo
// rows everywhere and lift the small-grid rule) or 0 = the library's rules
// (esq_stencil2d.hpp, geo_chain).  Part of the plugin OBJECT (option CHAIN_ROWS of
// esq_rhs_set_options; a user plugin passes its own to Stencil2D::chain): until round 5
// a process-wide table that one solver's options leaked through to the next.
struct ChainTuning {
    int rows = 0;
    bool rows_set = false;
};
// the process default (ESQ_CHAIN_ROWS), for plugin objects that were given no option
inline ChainTuning chain_tuning_default() {
    ChainTuning t;
    if (const char *env = env_get("CHAIN_ROWS")) { t.rows_set = true; t.rows = atoi(env); }
    return t;
}
// compute units of the device the calling thread has selected (the tile-height rules
// fill "one round of resident waves"); 256 where there is no device (detached
// contexts: esq_plan_describe)
inline int device_cus() {
    static std::atomic<int> cache[64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return 256;
    }
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            v < 1) {
            (void)hipGetLastError();
            v = 256;
        }
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
// split = one field per wave (k_chain2d<..., SPLIT = true>): the budget of a
// one-field kernel.  Depth 5 and 6 are instantiated with up to 6 memory rows.
inline ChainCaps chain_caps(bool split = true) {
    return split ? ChainCaps{{0, 0, 9, 9, 9, 6, 6}, {0, 0, 9, 9, 9, 6, 6}}
                 : ChainCaps{{0, 0, 8, 5, 1, -1, -1}, {0, 0, 7, 2, -1, -1, -1}};
}
inline bool chain_within_caps(int depth, bool solerr, int nu, bool split = true) {
    if (depth < 2 || depth > 6) return false;
    const ChainCaps c = chain_caps(split);
    return nu <= (solerr ? c.solerr[depth] : c.stage[depth]);
}

// device-side description of a chain (built from `esq_chain` by make_chain_args)
template <int D, int NU>
struct ChainArgs {
    static constexpr int kD = D, kNU = NU;
    static constexpr int NUa = NU > 0 ? NU : 1;
    const double *rows[NUa];         // K rows read from memory (union over targets)
    double cu[D][NUa];               // cu[e][u]: weight of rows[u] in target e + 1
    double eu[NUa];                  // SOLERR: error weights of the last target
    double ck[D][D];                 // ck[e][k]: weight of K_k in target e + 1 (k <= e)
    double ek[D];                    // SOLERR: error weight of K_k
    double c0[NUa];                  // FROMROWS: weights of rows[] in the chain's own
                                     // input T_0 = y + h * sum_u c0[u] rows[u]
    const double *y;                 // base state; nullptr: the chain's own input
    double h;
    double *fk[D];                   // where K_k goes (nullptr: not stored)
    double *out;                     // last target
    int f_nt;
    unsigned ld_nt;                  // non-temporal loads: bit 0 the input, bit 1 y,
                                     // bit 8 + u: rows[u] (read for the last time)
    RedArgs red;
};

// NF fields of N x N (state = field 0, field 1, ... one after the other);
// Fn::eval(centres, laplacians) -> derivatives, all per column pair.
// KINDLAST: ESQ_EPI_STAGE (the last target is a stage argument / y_new of an
// FSAL pair), ESQ_EPI_SOLERR (y_new and the error partial sums; ca.out == nullptr:
// the target is only the partner of y in the scale -- the early error estimate of
// BS5, whose y_pre nobody reads, bogacki.py:343-346) or ESQ_EPI_ERRNORM (FSAL pairs:
// the chain's LAST STAGE is the end-point evaluation K_s = f(t + h, y_new) -- its
// argument, target D - 1, is y_new and is stored to ca.out as well as handed on in
// registers -- and "target D" is no vector but the error sum
//     err = h * (sum_u cu[D-1][u] rows[u] + sum_{k < D} ck[D-1][k] K_k)
// with the error weights E where a stage target has its row of A; the partial sums
// of |err / (atol + rtol max(|y|, |y_new|))|^2 go to ca.red.partials.  common.py:
// 341-351 in one sweep with the stages before it).
#ifndef ESQ_CHAIN_PREFETCH
#define ESQ_CHAIN_PREFETCH true
#endif
// SPLIT (fields coupled only pointwise, e.g. the Brusselator's u and v): a tile is
// worked by NFT waves, ONE FIELD EACH (workgroup = 64 * NFT threads = one tile);
// the centre values a stage's pointwise function needs from the other fields go
// through LDS (2 x 1 KiB per field and stage, one workgroup barrier per grid
// row).  Per-wave register use is that of a one-field kernel, so depth-4
// chains with 8-9 memory rows keep two waves per SIMD -- the whole Pr8 step
// becomes E + three chains.  Same arithmetic, bit-identical.
// FROMROWS: the chain's input T_0 (the argument of its first stage) is not read from
// memory but formed from y and the memory rows the chain loads anyway
// (ca.c0) -- the previous sweep then need not write it.  The row sets
// are requested one grid row further ahead (T_0's row must exist before stage 0's
// window takes it) and the targets' sums over a row set wait one iteration.
template <int NFT, bool PERIODIC, int D, int NU, int KINDLAST, class Fn,
          bool SPLIT = false, bool FROMROWS = false, bool PREFETCH = ESQ_CHAIN_PREFETCH>
__global__ __launch_bounds__(kBlock) void k_chain2d(
    const double *__restrict__ ys, ChainArgs<D, NU> ca, Fn fn, int N, int npr,
    unsigned tpr, unsigned ntiles, unsigned nblocks, unsigned xcd) {
    constexpr int NF = SPLIT ? 1 : NFT;            // fields per wave
    constexpr int H = D - 1;                       // halo rows / lanes per side
    constexpr int W = 64 - 2 * H;                  // last-stage pairs per tile
    constexpr bool SOLERR = KINDLAST == ESQ_EPI_SOLERR;
    constexpr bool ERRN = KINDLAST == ESQ_EPI_ERRNORM;
    static_assert(!ERRN || D >= 2, "the end-point stage follows at least one stage");
    // a workgroup works on whole PAIRS of vertically adjacent tiles: one pair (each
    // tile NFT waves, one per field) or kBlock / 128 pairs (each tile one wave)
    constexpr int TPB = SPLIT ? 2 : kBlock / 64;        // tiles per workgroup
    constexpr int WAVES = SPLIT ? 2 * NFT : kBlock / 64;    // waves per workgroup
    static_assert(WAVES * 64 <= kBlock && TPB % 2 == 0, "a workgroup holds whole pairs");
    // SPLIT: the centre rows of all D stages, double-buffered by iteration parity
    __shared__ double2 xch[SPLIT ? 2 : 1][SPLIT ? D : 1][SPLIT ? 2 * NFT : 1][SPLIT ? 64 : 1];
    // T_k (k = 1 .. D - 1) of each tile's FIRST row: what the pair's other tile needs
    // from across the boundary they both start at
    __shared__ double2 pairx[TPB][D > 1 ? D - 1 : 1][NFT][64];
    // XCD band remap as in the one-stage sweeps: XCD x takes a contiguous band
    const unsigned per = (nblocks + xcd - 1) / xcd;
    const unsigned lb = (blockIdx.x % xcd) * per + blockIdx.x / xcd;
    // wave-uniform by construction; said so, the tile, its row range and every
    // branch on them are scalar for the compiler too
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned tib = SPLIT ? wave / NFT : wave;          // tile within the workgroup
    const unsigned tile = lb * TPB + tib;
    const int fbase = SPLIT ? (int)(wave % NFT) : 0;   // first field of this wave
    double local = 0.0;
    const bool active = lb < nblocks && tile < ntiles;          // wave-uniform
    if (active) {
        const int npairs = N / 2;
        const int lane = threadIdx.x & 63;
        // tile -> (pair, upper | lower half); pairs in row-major order of the grid
        const unsigned pq = tile >> 1, half = tile & 1u;
        const int pc = W * (int)(pq % tpr) - H + lane;
        const bool indom = pc >= 0 && pc < npairs;
        const bool live = PERIODIC ? (pc >= -H && pc < npairs + H) : indom;
        const int pw = PERIODIC ? (pc < 0 ? pc + npairs : (pc >= npairs ? pc - npairs : pc))
                                : pc;
        const bool store_ok = indom && lane >= H && lane < 64 - H;
        // the pair's rows [p0, p1), split in the middle: the upper tile marches UP
        // from the split, the lower one DOWN (the arithmetic does not see the
        // direction: the Laplacian adds the rows above and below in one commutative
        // add).  The pair below arrives at p1 from the other side at the same moment:
        // the run-out rows the two share meet in the XCD's L2.
        const int prow = (int)(pq / tpr);
        const int p0 = (int)(((long long)N * prow) / npr);
        const int p1 = (int)(((long long)N * (prow + 1)) / npr);
        const int mid = p0 + (p1 - p0) / 2;
        const int r0 = half ? mid : p0;
        const int Re = half ? p1 - mid : mid - p0;
        const int Re_max = (p1 - mid) > (mid - p0) ? (p1 - mid) : (mid - p0);
        const int dirn = half ? 1 : -1;
        const int rbase = dirn > 0 ? r0 : r0 + Re - 1;          // stage 0's first row
        const size_t fstride = (size_t)N * (size_t)npairs;     // pairs per field
        auto row_ok = [&](int r) { return PERIODIC || (r >= 0 && r < N); };
        auto wrap = [&](int r) {
            return PERIODIC ? (r < 0 ? r + N : (r >= N ? r - N : r)) : r;
        };
        // periodic grids: every row exists, and a lane beyond the halo columns
        // (its results reach no stored lane) simply reads column pair 0 -- no load
        // sits under a lane mask.  Dirichlet grids: outside is zero.
        const int pwl = PERIODIC ? (live ? pw : 0) : pw;
        auto ld_ys = [&](int r, int f) -> double2 {
            if (!PERIODIC && (!live || !row_ok(r))) return make_double2(0.0, 0.0);
            const size_t k_ = (size_t)(fbase + f) * fstride + (size_t)wrap(r) * npairs + pwl;
            return (ca.ld_nt & 1u) ? ld2_nt(ys, k_) : ld2(ys, k_);
        };
        const double2 zero = make_double2(0.0, 0.0);
        // the base of the targets' sums: y, or -- a chain from the state, ca.y == NULL --
        // the chain's own input.  One pointer, always loaded through (the row is in
        // cache: it was the window's next row an iteration ago): `ca.y ? yrow : wc[0]`
        // on two double2 lvalues makes the compiler select an ADDRESS and send the
        // windows through scratch
        const double *__restrict__ ybase = ca.y ? ca.y : ys;
        // every weight as a scalar of its own: taken straight from the argument
        // struct, the compiler keeps whole 16-register load tuples alive and, out
        // of scalar registers, re-reads a TUPLE for every use (706 v_readlane per
        // row in chain4+solerr<8>, half of its vector instructions)
        double w_cu[D][ChainArgs<D, NU>::NUa], w_eu[ChainArgs<D, NU>::NUa], w_ck[D][D],
            w_ek[D];
#pragma unroll
        for (int e = 0; e < D; ++e) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                asm("s_mov_b64 %0, %1" : "=s"(w_cu[e][j]) : "s"(ca.cu[e][j]));
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                asm("s_mov_b64 %0, %1" : "=s"(w_ck[e][k]) : "s"(ca.ck[e][k]));
            }
            asm("s_mov_b64 %0, %1" : "=s"(w_ek[e]) : "s"(ca.ek[e]));
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            asm("s_mov_b64 %0, %1" : "=s"(w_eu[j]) : "s"(ca.eu[j]));
        }
        double w_c0[ChainArgs<D, NU>::NUa];
        if constexpr (FROMROWS) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                asm("s_mov_b64 %0, %1" : "=s"(w_c0[j]) : "s"(ca.c0[j]));
            }
        }
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
        if constexpr (!FROMROWS) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                wm[0][f] = ld_ys(rbase - dirn, f);
                wc[0][f] = ld_ys(rbase, f);
            }
        }
        // no run-in rows (the partner hands the boundary values over); SPLIT: the
        // workgroup's barrier per row needs the same count in both tiles of the pair
        const int iters = (SPLIT ? Re_max : Re) + H;
        // operands of stage 0's row: loaded ONE ITERATION AHEAD, so that a wave
        // always has a whole row set in flight behind the row it computes on
        // (the sweeps are latency-bound: few waves per SIMD at this register use).
        // FROMROWS: the set of iteration IT is that of the row AFTER stage 0's.
        double2 u[ChainArgs<D, NU>::NUa][NF], yrow[NF], ysn[NF];
#define ESQ_CHAIN_LOAD_ROW(IT)                                                     \
    {                                                                              \
        const int rho_ = rbase + dirn * ((IT) + (FROMROWS ? 1 : 0));               \
        const bool act_ = PERIODIC || (live && row_ok(rho_));                      \
        const size_t base_ = (size_t)wrap(rho_) * npairs + pwl;                    \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) {                           \
            if (!FROMROWS) ysn[f] = ld_ys(rho_ + dirn, f);                         \
            const size_t k2_ = (size_t)(fbase + f) * fstride + base_;              \
            _Pragma("unroll") for (int j = 0; j < NU; ++j)                         \
                u[j][f] = !act_ ? zero : ((ca.ld_nt >> (8 + j)) & 1u)                 \
                              ? ld2_nt(ca.rows[j], k2_) : ld2(ca.rows[j], k2_);       \
            yrow[f] = !act_ ? zero : (ca.ld_nt & 2u) ? ld2_nt(ybase, k2_)           \
                                                     : ld2(ybase, k2_);        \
        }                                                                          \
    }
        // FROMROWS: T_0 of the row a set belongs to (same ascending FMA chain, the
        // same two roundings as the sweep that would have written it)
#define ESQ_CHAIN_FORM_T0(IT, U_, Y_, DST)                                                 \
    {                                                                              \
        const int rho_ = rbase + dirn * ((IT) + 1);                                \
        const bool act_ = live && row_ok(rho_);                                    \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) {                           \
            double2 s0_ = zero;                                                    \
            _Pragma("unroll") for (int j = 0; j < NU; ++j) {                       \
                s0_.x = fma(w_c0[j], U_[j][f].x, s0_.x);                           \
                s0_.y = fma(w_c0[j], U_[j][f].y, s0_.y);                           \
            }                                                                      \
            const double2 t0_ = make_double2(__dadd_rn(Y_[f].x, __dmul_rn(ca.h, s0_.x)), \
                                             __dadd_rn(Y_[f].y, __dmul_rn(ca.h, s0_.y))); \
            /* componentwise: `c ? a : b` on two double2 lvalues selects an        \
               ADDRESS and loads through it -- a round trip through scratch */     \
            DST[f] = make_double2(act_ ? t0_.x : 0.0, act_ ? t0_.y : 0.0);         \
        }                                                                          \
    }
        // the D targets' sums over a row set (leading partial + memory rows)
#define ESQ_CHAIN_ROW_SUMS(S, SE)                                                  \
    _Pragma("unroll") for (int e = 0; e < D; ++e)                                  \
    _Pragma("unroll") for (int f = 0; f < NF; ++f) {                               \
        double2 s_ = zero, se_ = zero;                                             \
        _Pragma("unroll") for (int j = 0; j < NU; ++j) {                           \
            s_.x = fma(w_cu[e][j], uc[j][f].x, s_.x);                              \
            s_.y = fma(w_cu[e][j], uc[j][f].y, s_.y);                              \
            if (SOLERR && e == D - 1) {                                            \
                se_.x = fma(w_eu[j], uc[j][f].x, se_.x);                           \
                se_.y = fma(w_eu[j], uc[j][f].y, se_.y);                           \
            }                                                                      \
        }                                                                          \
        S[e][f] = s_;                                                              \
        if (SOLERR && e == D - 1) SE[f] = se_;                                     \
    }
        // FROMROWS: sums and base row of the set that arrived one iteration ago
        double2 sh[D][NF], seh[NF], yh[NF];
        if constexpr (FROMROWS) {
            ESQ_CHAIN_LOAD_ROW(-2)                       // row rbase - dirn
            ESQ_CHAIN_FORM_T0(-2, u, yrow, wm[0])
            ESQ_CHAIN_LOAD_ROW(-1)                       // row rbase
            ESQ_CHAIN_FORM_T0(-1, u, yrow, wc[0])
            {
                double2 uc[ChainArgs<D, NU>::NUa][NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    yh[f] = yrow[f];
#pragma unroll
                    for (int j = 0; j < NU; ++j) uc[j][f] = u[j][f];
                }
                ESQ_CHAIN_ROW_SUMS(sh, seh)
            }
        }
        ESQ_CHAIN_LOAD_ROW(0)
        for (int it = 0; it < iters; ++it) {
            const int rho0 = rbase + dirn * it;            // stage 0's row
            // ---- take over the row loaded one iteration ago ...
            double2 uc[ChainArgs<D, NU>::NUa][NF], ycur[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if constexpr (!FROMROWS) {
                    wp[0][f] = ysn[f];
                    yf[0][f] = yrow[f];
                } else {
                    ycur[f] = yrow[f];
                }
#pragma unroll
                for (int j = 0; j < NU; ++j) uc[j][f] = u[j][f];
            }
            // ---- ... and request the next one before any arithmetic
            if (PREFETCH) {
                if (it + 1 < iters) ESQ_CHAIN_LOAD_ROW(it + 1)
            }
            if constexpr (FROMROWS) {
                ESQ_CHAIN_FORM_T0(it, uc, ycur, wp[0])
            }
            if constexpr (SPLIT) {
                // the other fields' centre values, for ALL stages at once: a stage's
                // centre row was completed an iteration ago, so the exchange (LDS
                // write, ONE workgroup barrier, LDS read) is off the stage-to-stage
                // dependency chain.  Two buffers: a wave that runs ahead writes
                // iteration it + 1's values while its partner still reads it's.
#pragma unroll
                for (int k = 0; k < D; ++k) xch[it & 1][k][wave][lane] = wc[k][0];
                __syncthreads();
            } else {
                // the pair's boundary values were written in the iteration before.  An
                // LDS-only barrier: __syncthreads() would also drain vmcnt -- the row
                // set requested one iteration ahead
                if (it >= 1 && it < D) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // stage `it` starts in this iteration, at the tile's first row: the row
            // before it is the partner's first row
            // ---- the D targets' sums: of row rho0 (this set), or -- FROMROWS -- of
            // the row after it, held for one iteration
            if constexpr (!FROMROWS) {
                double2 s0[D][NF], se0[NF];
                ESQ_CHAIN_ROW_SUMS(s0, se0)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int e = 0; e < D; ++e) acc[e][0][f] = s0[e][f];
                    if (SOLERR) acce[0][f] = se0[f];
                }
            } else {
                double2 s0[D][NF], se0[NF];
                ESQ_CHAIN_ROW_SUMS(s0, se0)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int e = 0; e < D; ++e) {
                        acc[e][0][f] = sh[e][f];
                        sh[e][f] = s0[e][f];
                    }
                    if (SOLERR) {
                        acce[0][f] = seh[f];
                        seh[f] = se0[f];
                    }
                    yf[0][f] = yh[f];
                    yh[f] = ycur[f];
                }
            }
            // ---- the stages, each one row behind its predecessor
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (it >= k) {                                     // wave-uniform
                    const int rho = rho0 - dirn * k;
                    const bool actk = live && row_ok(rho);
                    const bool own = rho >= r0 && rho < r0 + Re;
                    const size_t basek = (size_t)wrap(rho) * npairs + pw;
                    double2 cc[NF], lap[NF], fK[NF];
                    // the window's row before this one -- as SCALARS (a conditional
                    // overwrite of a double2 makes the compiler select an ADDRESS and
                    // send the windows through scratch)
                    double upx[NF], upy[NF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) { upx[f] = wm[k][f].x; upy[f] = wm[k][f].y; }
                    if (k >= 1 && it == k) {                       // wave-uniform
                        // the stage's first row: the row before it is the partner's
                        // first row (written an iteration ago, behind a barrier)
                        asm volatile("");
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const double *pv = reinterpret_cast<const double *>(
                                &pairx[tib ^ 1u][k >= 1 ? k - 1 : 0][fbase + f][lane]);
                            upx[f] = pv[0];
                            upy[f] = pv[1];
                        }
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const double lf = lane_left(wc[k][f].y);
                        const double rt = lane_right(wc[k][f].x);
                        cc[f] = wc[k][f];
                        lap[f].x = ((upx[f] + wp[k][f].x) + (lf + wc[k][f].y)) -
                                   4.0 * wc[k][f].x;
                        lap[f].y = ((upy[f] + wp[k][f].y) + (wc[k][f].x + rt)) -
                                   4.0 * wc[k][f].y;
                    }
                    if constexpr (SPLIT) {
                        double2 call[NFT];
#pragma unroll
                        for (int g = 0; g < NFT; ++g)
                            call[g] = xch[it & 1][k][tib * NFT + g][lane];
                        fK[0] = fn.eval_one(fbase, call, lap[0]);
                    } else {
                        fn.eval(cc, lap, fK);
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const size_t k2 = (size_t)(fbase + f) * fstride + basek;
                        if (ca.fk[k] && own && store_ok) {
                            if (ca.f_nt) st2_nt(ca.fk[k], k2, fK[f]);
                            else st2(ca.fk[k], k2, fK[f]);
                        }
                        // K_k enters the sums of the later targets
#pragma unroll
                        for (int e = k; e < D; ++e) {
                            acc[e][k][f].x = fma(w_ck[e][k], fK[f].x, acc[e][k][f].x);
                            acc[e][k][f].y = fma(w_ck[e][k], fK[f].y, acc[e][k][f].y);
                            if (SOLERR && e == D - 1) {
                                acce[k][f].x = fma(w_ek[k], fK[f].x, acce[k][f].x);
                                acce[k][f].y = fma(w_ek[k], fK[f].y, acce[k][f].y);
                            }
                        }
                        // target k + 1 is complete for this row
                        const double2 s = acc[k][k][f];
                        const double2 t =
                            make_double2(__dadd_rn(yf[k][f].x, __dmul_rn(ca.h, s.x)),
                                         __dadd_rn(yf[k][f].y, __dmul_rn(ca.h, s.y)));
                        if (k + 1 < D) {
                            wp[k + 1 < D ? k + 1 : k][f] =
                                make_double2(actk ? t.x : 0.0, actk ? t.y : 0.0);
                            // the tile's first row: the partner's stage k + 1 needs it
                            if (it == k)
                                pairx[tib][k][fbase + f][lane] =
                                    make_double2(actk ? t.x : 0.0, actk ? t.y : 0.0);
                            // ERRNORM: the end-point stage's argument is y_new
                            if (ERRN && k == D - 2 && own && store_ok) st2(ca.out, k2, t);
                        } else if (ERRN) {
                            // the error sum of the row: this stage's input (centre) is
                            // y_new, yf the state it started from
                            if (own && store_ok) {
                                const double2 er = make_double2(__dmul_rn(ca.h, s.x),
                                                                __dmul_rn(ca.h, s.y));
                                if (ca.red.atol_vec)
                                    local += ratio_sq<false>(er, yf[k][f], cc[f], ca.red.atol_vec,
                                                             ca.red.atol_s, ca.red.rtol, k2,
                                                             ca.red.n_valid);
                                else
                                    local += ratio_sq<false>(er, yf[k][f], cc[f], nullptr,
                                                             ca.red.atol_s, ca.red.rtol, k2,
                                                             ca.red.n_valid);
                            }
                        } else if (own && store_ok) {
                            if (ca.out) st2(ca.out, k2, t);
                            if (SOLERR) {
                                const double2 er = make_double2(__dmul_rn(ca.h, acce[k][f].x),
                                                                __dmul_rn(ca.h, acce[k][f].y));
                                // two copies: with a scalar atol no load (and no wait
                                // for the prefetched row behind it) is here
                                if (ca.red.atol_vec)
                                    local += ratio_sq<false>(er, yf[k][f], t, ca.red.atol_vec,
                                                             ca.red.atol_s, ca.red.rtol, k2,
                                                             ca.red.n_valid);
                                else
                                    local += ratio_sq<false>(er, yf[k][f], t, nullptr,
                                                             ca.red.atol_s, ca.red.rtol, k2,
                                                             ca.red.n_valid);
                            }
                        }
                    }
                }
            }
            if (!PREFETCH) {
                if (it + 1 < iters) ESQ_CHAIN_LOAD_ROW(it + 1)
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
    if (!SPLIT && !active) {
        // (a wave without a tile keeps the workgroup's barrier count)
        for (int it = 1; it < D; ++it) asm volatile("s_barrier" ::: "memory");
    }
    if (SOLERR || ERRN) block_partial_w<WAVES>(local, ca.red.partials);
}
#undef ESQ_CHAIN_LOAD_ROW
#undef ESQ_CHAIN_FORM_T0
#undef ESQ_CHAIN_ROW_SUMS

}  // namespace esq
