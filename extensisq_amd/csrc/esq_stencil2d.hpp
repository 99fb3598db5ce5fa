// esq_stencil2d.hpp -- everything a 2-D five-point-stencil RHS plugin needs besides
// its POINTWISE FUNCTOR: the one-stage sweep (all fused epilogues, the on-the-fly
// first-stage input, the Chebyshev recursion), the marching chain sweeps
// (esq_chain.hpp) and their launch geometry.  The built-in Brusselator and heat
// plugins are two instantiations of this header; a user plugin is a third
// (tests/test_gpu_parity.py::test_user_compiled_chain_plugin).
//
//   struct MyFn {                       // NF fields of N x N, state = field 0, 1, ...
//       double p0, p1;                  // parameters (passed by value to the kernels)
//       // centres c[f] and five-point Laplacians lap[f] = ((up + down) + (left +
//       // right)) - 4 c  of one column PAIR (.x, .y)  ->  derivatives f[f]
//       __device__ void eval(const double2 (&c)[NF], const double2 (&lap)[NF],
//                            double2 (&f)[NF]) const;
//       // the same for ONE field (chain sweeps that give each field a wave of its
//       // own: fields coupled only pointwise); c holds all NF centres
//       __device__ double2 eval_one(int field, const double2 (&c)[NF], double2 lap) const;
//   };
//   using P = esq::Stencil2D<NF, /*PERIODIC=*/true, MyFn>;
//   extern "C" int my_rhs(void* user, double t, const double* y, double* f, size_t n, void* s)
//       { return P::rhs(fn_of(user), N_of(user), y, f, s); }
//   extern "C" int my_fused(void* user, double t, const double* y, double* f,
//                           const esq_epilogue* e, size_t n, void* s, void* e0, void* e1)
//       { return P::fused(fn_of(user), N_of(user), y, f, e, s, e0, e1); }
//   extern "C" int my_chain(void* user, const double* y, const esq_chain* c, size_t n,
//                           void* s, void* e0, void* e1)
//       { return P::chain(fn_of(user), N_of(user), y, c, s, e0, e1); }
//
// N must be even (16-byte column pairs); odd grids: return ESQ_ENOTSUP from the
// fused / chain entries and keep a scalar kernel behind esq_rhs_fn.
// Workgroups are dealt round-robin over the 8 XCDs, so block b is remapped to a
// contiguous band of rows per XCD (blockIdx % 8 = XCD label): the up/down
// neighbour rows then hit the SAME XCD's L2 instead of being fetched twice.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_plugin.hpp"
#include "esq_rkc2d.hpp"
#include "esq_terms.hpp"

namespace esq {

constexpr int kXcd = 8;

// band remap: logical block id such that XCD x (label blockIdx%8) sweeps the
// contiguous range [x*per, (x+1)*per)
__device__ __forceinline__ unsigned band_block(unsigned b, unsigned nblocks) {
    const unsigned per = (nblocks + kXcd - 1) / kXcd;
    return (b % kXcd) * per + b / kXcd;
}

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

// ---------------------------------------------------------------------------
// Vectorised 5-point sweeps (even N): one thread owns a column PAIR (16-byte
// accesses); the left/right neighbours come from the adjacent lanes (DPP wave
// shifts), only the lanes at a wave or row edge touch memory for them.
// ---------------------------------------------------------------------------
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
        lf = lane_left(c.y);
        rt = lane_right(c.x);
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

// the on-the-fly first-stage input is instantiated for the epilogues a first
// stage can have: the second stage's argument with at most one row from memory
template <class E> inline constexpr bool kFirstStage = false;
template <> inline constexpr bool kFirstStage<EpiStage<0>> = true;
template <> inline constexpr bool kFirstStage<EpiStage<1>> = true;
inline bool first_stage_ok(const esq_epilogue *e) {
    return e->kind == ESQ_EPI_STAGE && e->nt <= 1 && e->in_base;
}
inline SrcAxpy axpy_of(const esq_epilogue *e) {
    return SrcAxpy{e->in_base, e->in_row, e->in_c, e->in_h};
}

// ONE-STAGE SWEEP.  One wave tile = 64 column pairs of ONE grid row; all three
// window rows of every field are requested up front, together with the
// epilogue's operands, so every load of the thread is in flight before the first
// use.  `Epi` (esq_epilogue.hpp) says what happens to the fresh derivative.  The
// epilogues are pointwise: nothing is recomputed on halos.
template <int NF, bool PERIODIC, class Fn, class Epi, class Src>
__global__ __launch_bounds__(kBlock) void k_stencil2d_sweep(
    Src ys, double *__restrict__ f, Epi epi, Fn fn, int N, unsigned nblocks, unsigned wpr) {
    const unsigned tile = band_block(blockIdx.x, nblocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int i = (int)(tile / wpr);
    double local = 0.0;
    if (i < N) {                                           // wave-uniform
        const size_t NN = (size_t)N * N;
        RowWin<PERIODIC, Src> W[NF];
        size_t k2[NF];
        typename Epi::In in[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            W[q].src = ys;
            W[q].base = (size_t)q * NN;
            W[q].N = N;
            W[q].npairs = (unsigned)N / 2;
            W[q].pair = (tile % wpr) * 64 + (threadIdx.x & 63);
            W[q].live = W[q].pair < W[q].npairs;
            k2[q] = ((size_t)q * NN + (size_t)i * N) / 2 + (W[q].live ? W[q].pair : 0);
            epi.load(in[q], k2[q]);
        }
        double2 up[NF], cc[NF], dn[NF], lap[NF], out[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            up[q] = W[q].row(i - 1);
            cc[q] = W[q].row(i);
            dn[q] = W[q].row(i + 1);
        }
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            double lf, rt;
            W[q].sides(i, cc[q], lf, rt);
            lap[q].x = ((up[q].x + dn[q].x) + (lf + cc[q].y)) - 4.0 * cc[q].x;
            lap[q].y = ((up[q].y + dn[q].y) + (cc[q].x + rt)) - 4.0 * cc[q].y;
        }
        fn.eval(cc, lap, out);
        if (W[0].live) {
#pragma unroll
            for (int q = 0; q < NF; ++q) epi.store_f(f, k2[q], out[q]);
#pragma unroll
            for (int q = 0; q < NF; ++q) epi.finish(in[q], out[q], cc[q], k2[q], local);
        }
    }
    if (Epi::kReduce) block_partial(local, epi.red.partials);
}

// ---- launch geometry of the 2-D sweeps: one wave tile per 64 column pairs
struct Geo2d {
    unsigned wpr, grid;
};
inline Geo2d geo2d(int N) {
    Geo2d g;
    g.wpr = (N / 2 + 63) / 64;                                  // wave tiles per row
    const unsigned tiles = g.wpr * (unsigned)N;
    const unsigned nblocks = (tiles + kBlock / 64 - 1) / (kBlock / 64);
    g.grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    return g;
}

// ---- launch geometry of the marching chain sweeps (esq_chain.hpp): one wave per
// tile of R rows x (64 - 2(D-1)) column pairs.  R balances the halo rows
// (2(D-1) per tile) against keeping every wave slot of the chip busy: the tile
// count is made a multiple of the resident waves (256 CUs x waves per CU from
// the occupancy query), so that the launch runs as whole rounds.
struct GeoChain {
    int R;
    unsigned tpr, ntiles, nblocks, grid;
    bool clamped = false;     // R is the minimum height: the grid leaves wave slots free
};
template <class Kernel>
inline int chain_waves_per_cu(Kernel kern, unsigned block) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, (int)block, 0) != hipSuccess ||
        blocks < 1)
        blocks = 2;
    int waves = blocks * (int)(block / 64);
    if (waves > 32) waves = 32;
    return waves;
}
// Until round 5 a chain tile carried 2(D-1) halo rows and the plugins refused chains on
// grids below ~450 x 450, where D single sweeps were as fast (Pr8 at N = 316: 78 us
// unchained, 87 chained).  With diverging pairs (no run-in rows), no test per FMA pair and
// four-row tiles the chains win down to N = 96 -- Pr8 on the Brusselator 0.066 -> 0.050
// ms/step there, 0.095 -> 0.056 at N = 448; Ts5 on the heat plugin 0.033 -> 0.023 at
// N = 128 (tools/r06_small_grids.sh): round 2's "65 us launch floor" of small device-RHS
// states was a floor of thirteen launches.  With two-row tiles (where the grid leaves wave
// slots free) the chains win from the smallest grid the kernels take, N = 16: Pr8 0.069 ->
// 0.040 ms/step, Ts5 0.035 -> 0.020.
inline bool chain_fits_grid(int N, int depth, const ChainTuning &tune) {
    (void)depth; (void)tune;
    return N >= 16;
}
// tiles_per_block: wave tiles a workgroup works on; waves_per_tile: waves that
// share one tile (the split sweeps: one per field)
inline GeoChain geo_chain(int N, int depth, int waves_per_cu, int tiles_per_block,
                          int waves_per_tile, const ChainTuning &tune,
                          bool tall_if_one_round = false, int min_rows = 0) {
    GeoChain g;
    const int W = 64 - 2 * (depth - 1);
    g.tpr = ((unsigned)N / 2 + W - 1) / W;
    const size_t cus = (size_t)device_cus();
    int R = tune.rows_set ? tune.rows : 0;                          // tuning / tests
    if (R <= 0) {
        // ONE round of resident waves at the kernel's own occupancy: the sweeps
        // are latency-bound per wave (a row's loads are one iteration ahead, no
        // more), so every wave slot should hold a tile, and a second, partial
        // round costs a whole one.  Tiles of at most 48 rows (beyond, two rounds
        // of shorter tiles), at least depth + 2 (the halo rows are recomputed).
        // Measured (profiles/r03_experiments.md, tools/small_chain_sweep.py): Pr8
        // at N = 1000: R = 9..12 160 us/step, R = 18 at one wave per SIMD 300;
        // at N = 2236: R = 44 (two waves per SIMD) / 30 (three).
        auto rows_for = [&](size_t slots) -> int {
            const size_t max_row_tiles = slots / (size_t)waves_per_tile / g.tpr;
            if (max_row_tiles == 0) return N + 1;
            return (int)(((size_t)N + max_row_tiles - 1) / max_row_tiles);
        };
        // tall_if_one_round (one-field sweeps, light rows -- the heat plugin): where
        // ONE wave per SIMD already gives tiles of 24..48 rows the launch is
        // bandwidth-bound and the fewer halo rows win (heat Pr9 at N = 2236:
        // R = 44 0.587 ms/step, R = 15 at three waves per SIMD 0.62)
        // ... and where TWO waves per SIMD give tiles of 20+ rows, those: half as many
        // marching steps per wave and a second wave to fill the first one's issue gaps
        // (Pr9 at N = 2236, tools/chain_rows_sweep.sh: R = 22 0.5235 ms/step with
        // chain5<0> 50 us, R = 44 0.540 with 65 us; R = 15, 18, 26 .. 36 -- three waves,
        // or a partial second round -- 0.549 .. 0.600)
        if (tall_if_one_round) {
            const int tall_waves = 2;     // (tools/chain_rows_sweep.sh, round 4)
            for (int w = tall_waves < waves_per_cu / 4 ? tall_waves : waves_per_cu / 4;
                 w >= 1 && R <= 0; --w) {
                const int cand = rows_for(cus * 4 * (size_t)w);
                if (cand >= (w == 1 ? 24 : 20) && cand <= 48) R = cand;
            }
        }
        size_t rounds = 1;
        if (R <= 0) R = rows_for(cus * (size_t)waves_per_cu);
        while (R > 48) {
            ++rounds;
            R = rows_for(cus * (size_t)waves_per_cu * rounds);
        }
        // (the halo rows are recomputed: the Brusselator's heavier rows want depth + 2,
        // the heat sweeps fill the wave slots down to `depth` rows -- Ts5 at N = 1000:
        // chain5<1> 31 us on 7-row tiles, 27 on 5-row tiles, 36 on 4-row tiles)
        // (independent tiles with run-in rows -- the Chebyshev chains: depth + 2 unless
        // the caller says otherwise; the diverging pairs of the explicit chains: down to
        // two-row tiles where the grid does not fill the wave slots -- Pr8 at N = 128,
        // tools/r06_small_grids.sh: R = 2 0.041 ms/step, 3 0.047, 4 0.052, 6 0.064)
        if (min_rows <= 0) min_rows = depth + 2;
        if (R < min_rows) { R = min_rows; g.clamped = true; }
    }
    if (R > N) R = N;
    g.R = R;
    g.ntiles = g.tpr * (unsigned)((N + R - 1) / R);
    g.nblocks = (g.ntiles + tiles_per_block - 1) / tiles_per_block;
    g.grid = ((g.nblocks + kXcd - 1) / kXcd) * kXcd;
    return g;
}

// ... for the chain sweeps of the explicit pairs (k_chain2d): the tiles come in
// DIVERGING PAIRS that split the rows [N prow / npr, N (prow + 1) / npr) between them
// (esq_chain.hpp); R from the rules above, npr pair rows, both tiles of every pair
// non-empty
struct GeoPairs {
    int R, npr;
    unsigned tpr, ntiles, nblocks, grid;
};
inline GeoPairs geo_chain_pairs(int N, int depth, int waves_per_cu, int tiles_per_block,
                                int waves_per_tile, const ChainTuning &tune,
                                bool tall_if_one_round = false, int min_rows = 0) {
    const GeoChain g = geo_chain(N, depth, waves_per_cu, tiles_per_block, waves_per_tile, tune,
                                 tall_if_one_round, min_rows);
    GeoPairs p;
    p.R = g.R;
    p.tpr = g.tpr;
    // never MORE tiles than the independent tiles of that height would be: the rules
    // above fill exactly one round of wave slots, and a tile beyond them costs a whole
    // second round (first version: 52 instead of 51 tile rows at N = 2236 put 2080
    // waves on 2048 slots -- chain5<0> 102 -> 162 us)
    p.npr = ((N + g.R - 1) / g.R) / 2;
    {
        // ... unless there is room: then pairs of at most 2R rows, so that no tile is
        // taller than R (N = 500, R = 4: 63 pairs of 7.9 rows instead of 62 of 8.1, whose
        // five-row tiles set the pace of every workgroup)
        // (only where R is the minimum height: a height chosen to fill exactly one round
        // -- Pr9 on the heat plugin at N = 2236: 107 tile rows of 21 on 2048 slots -- must
        // not get a 108th)
        const int up = (N + 2 * g.R - 1) / (2 * g.R);
        const size_t slots = (size_t)device_cus() * (size_t)waves_per_cu;
        if (g.clamped && (size_t)2 * up * g.tpr * (size_t)waves_per_tile <= slots) p.npr = up;
    }
    if (p.npr > N / 2) p.npr = N / 2;
    if (p.npr < 1) p.npr = 1;
    p.ntiles = 2u * p.tpr * (unsigned)p.npr;
    p.nblocks = (p.ntiles + tiles_per_block - 1) / tiles_per_block;
    p.grid = ((p.nblocks + kXcd - 1) / kXcd) * kXcd;
    return p;
}

// ---------------------------------------------------------------------------
// The three (four) entry points of a 2-D stencil plugin, by functor.
//   SPLIT_CHAINS: the chain sweeps give every field a wave of its own (fields
//   coupled only pointwise, Fn::eval_one): the register budget of a one-field
//   kernel, so depth-4 chains with 8-9 memory rows keep two waves per SIMD.
// ---------------------------------------------------------------------------
template <int NF, bool PERIODIC, class Fn, bool SPLIT_CHAINS = (NF > 1)>
struct Stencil2D {
    static bool grid_ok(int N) { return N % 2 == 0 && N >= 4; }

    // f = fun(t, y)                                             (esq_rhs_fn)
    static int rhs(const Fn &fn, int N, const double *y, double *f, void *stream) {
        if (!grid_ok(N)) return ESQ_ENOTSUP;
        const Geo2d g = geo2d(N);
        EpiNone ep{};
        hipLaunchKernelGGL((k_stencil2d_sweep<NF, PERIODIC, Fn, EpiNone, SrcPlain>),
                           dim3(g.grid), dim3(kBlock), 0, (hipStream_t)stream, SrcPlain{y}, f,
                           ep, fn, N, g.grid, g.wpr);
        return (int)hipGetLastError();
    }

    // the sweep + a pointwise epilogue                          (esq_rhs_fused_fn)
    static int fused(const Fn &fn, int N, const double *y_in, double *f,
                     const esq_epilogue *epi, void *stream, void *start_event,
                     void *stop_event) {
        if (!epi) return ESQ_EINVAL;
        if (!grid_ok(N)) return ESQ_ENOTSUP;
        const Geo2d g = geo2d(N);
        if (epilogue_reduces(epi)) {
            if ((int)g.grid > epi->partials_cap) return ESQ_ENOTSUP;
            if (epi->partials_used && !epi->dry_run) *epi->partials_used = (int)g.grid;
        }
        if (epi->in_row && !first_stage_ok(epi)) return ESQ_ENOTSUP;
        const int rc = dispatch_epilogue(epi, [&](auto ep) {
            using E = decltype(ep);
            if constexpr (kFirstStage<E>) {
                if (epi->in_row) {
                    hipExtLaunchKernelGGL((k_stencil2d_sweep<NF, PERIODIC, Fn, E, SrcAxpy>),
                                          dim3(g.grid), dim3(kBlock), 0, (hipStream_t)stream,
                                          (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                                          axpy_of(epi), f, ep, fn, N, g.grid, g.wpr);
                    return;
                }
            }
            hipExtLaunchKernelGGL((k_stencil2d_sweep<NF, PERIODIC, Fn, E, SrcPlain>),
                                  dim3(g.grid), dim3(kBlock), 0, (hipStream_t)stream,
                                  (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                                  SrcPlain{y_in}, f, ep, fn, N, g.grid, g.wpr);
        });
        return (rc || epi->dry_run) ? rc : (int)hipGetLastError();
    }

    // sweep + Chebyshev recursion, f not stored                 (esq_rhs_rkc_fn)
    static int rkc(const Fn &fn, int N, const double *yjm1, const EpiRkc &epi, void *stream,
                   void *start_event, void *stop_event) {
        if (!grid_ok(N)) return ESQ_ENOTSUP;
        const Geo2d g = geo2d(N);
        hipExtLaunchKernelGGL((k_stencil2d_sweep<NF, PERIODIC, Fn, EpiRkc, SrcPlain>),
                              dim3(g.grid), dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                              SrcPlain{yjm1}, (double *)nullptr, epi, fn, N, g.grid, g.wpr);
        return (int)hipGetLastError();
    }

    // `depth` consecutive Chebyshev stages in one marching sweep (esq_rhs_rkc_chain_fn;
    // one-field plugins; the FIRST form -- the chain opens a step -- and the LAST
    // form -- it ends one: f(t_end, y) and the error estimate as one more stage slot --
    // included).  Depth 2 .. 6 (LAST: 2 .. 5).
    static int rkc_chain(const Fn &fn, int N, const esq_rkc_chain *ch, void *stream,
                         void *start_event, void *stop_event,
                         const ChainTuning *tuning = nullptr) {
        // (a plugin object's own tile height, else the process default)
        const ChainTuning tune = tuning ? *tuning : chain_tuning_default();
        if (!ch || !ch->yjm2 || !ch->yn || !ch->fn || !ch->out) return ESQ_EINVAL;
        if (!ch->yjm1 && ch->yjm2 != ch->yn) return ESQ_EINVAL;     // FIRST: y_{j-2} = y_n
        if constexpr (NF != 1) {
            return ESQ_ENOTSUP;
        } else {
            if (N % 2 != 0 || N < 16) return ESQ_ENOTSUP;
            if (ch->fy_out && (ch->out_prev || !ch->partials || !ch->yjm1)) return ESQ_EINVAL;
            const int slots = ch->depth + (ch->fy_out ? 1 : 0);
            if (ch->depth < 2 || slots > 6) return ESQ_ENOTSUP;
            if (!chain_fits_grid(N, ch->depth + 1, tune)) return ESQ_ENOTSUP;
            if ((unsigned long long)N * N * 8ull > 0xffffffffull - 16ull) return ESQ_ENOTSUP;
            int rc = ESQ_ENOTSUP;
            auto launch = [&](auto depth_c, auto first_c, auto last_c) {
                constexpr int DD = decltype(depth_c)::value;           // stage slots
                constexpr bool kFirst = decltype(first_c)::value;
                constexpr bool kLast = decltype(last_c)::value;
                auto kern = k_rkc2d_chain<PERIODIC, DD, Fn, kFirst, kLast>;
                static const int wpc = chain_waves_per_cu(kern, (unsigned)kBlock);
                // (tile width: 64 - 2*ceil(D/2) pairs = geo_chain's rule for depth
                // ceil(D/2) + 1; D - 1 run-in rows per side)
                const GeoChain g = geo_chain(N, (DD + 1) / 2 + 1, wpc, kBlock / 64, 1, tune,
                                             /*tall_if_one_round=*/true, /*min_rows=*/DD);
                Rkc2dArgs<DD> a;
                a.a = ch->yjm1; a.b = ch->yjm2; a.yn = ch->yn; a.fn = ch->fn;
                a.out = ch->out; a.outp = ch->out_prev;
                a.hmus1 = ch->hmus_first;
                a.h04 = 0.0;
                a.red = RedArgs{};
                for (int k = 0; k < DD; ++k) {
                    const bool stage = k < ch->depth;      // (LAST: slot DD - 1 is the end)
                    a.mu[k] = stage ? ch->mu[k] : 0.0; a.nu[k] = stage ? ch->nu[k] : 0.0;
                    a.omn[k] = stage ? ch->omn[k] : 0.0; a.hmus[k] = stage ? ch->hmus[k] : 0.0;
                    a.ajm1[k] = stage ? ch->ajm1[k] : 0.0;
                }
                if constexpr (kLast) {
                    if ((int)g.grid > ch->partials_cap) { rc = ESQ_ENOTSUP; return; }
                    if (ch->partials_used) *ch->partials_used = (int)g.grid;
                    a.out = ch->fy_out;                    // the slot's "result" ...
                    a.outp = ch->out;                      // ... and its input: y_{n+1}
                    a.h04 = 0.4 * ch->h;
                    a.red.atol_vec = ch->atol_vec; a.red.atol_s = ch->atol_s;
                    a.red.rtol = ch->rtol; a.red.n_valid = ch->n_valid;
                    a.red.partials = ch->partials;
                }
                if (ch->read_amplification)
                    *ch->read_amplification = (double)(g.R + 2 * (DD - 1)) / g.R * 64.0 /
                                              (64 - 2 * ((DD + 1) / 2));
                hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(kBlock), 0, (hipStream_t)stream,
                                      (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, a, fn,
                                      N, g.R, g.tpr, g.ntiles, g.nblocks, (unsigned)kXcd);
                rc = (int)hipGetLastError();
            };
            auto by_form = [&](auto depth_c) {
                if (ch->fy_out) launch(depth_c, std::false_type{}, std::true_type{});
                else if (ch->yjm1) launch(depth_c, std::false_type{}, std::false_type{});
                else launch(depth_c, std::true_type{}, std::false_type{});
            };
            switch (slots) {
                case 2: by_form(std::integral_constant<int, 2>{}); break;
                case 3: by_form(std::integral_constant<int, 3>{}); break;
                case 4: by_form(std::integral_constant<int, 4>{}); break;
                case 5: by_form(std::integral_constant<int, 5>{}); break;
                case 6: by_form(std::integral_constant<int, 6>{}); break;
                default: break;
            }
            return rc;
        }
    }

    // `depth` consecutive stages in one marching sweep          (esq_rhs_chain_fn)
    // tall_tiles / min_rows: tile-height rules of geo_chain (light one-field rows)
    // LO..HI: the depths this call is instantiated for (a plugin may give each range
    // a translation unit of its own; a depth outside it: ESQ_ENOTSUP)
    template <int LO = 2, int HI = 6>
    static int chain(const Fn &fn, int N, const double *y_in, const esq_chain *chain,
                     void *stream, void *start_event, void *stop_event,
                     bool tall_tiles = false, int min_rows = 0,
                     const ChainTuning *tuning = nullptr) {
        constexpr bool split = SPLIT_CHAINS && NF > 1;
        if (!chain) return ESQ_EINVAL;
        if (N % 2 != 0 || N < 16) return ESQ_ENOTSUP;
        const ChainTuning tune = tuning ? *tuning : chain_tuning_default();
        if (!chain_fits_grid(N, chain->depth, tune)) return ESQ_ENOTSUP;
        // register budget (esq_chain.hpp, ChainCaps)
        if (NF > 1 && !chain_within_caps(chain->depth, chain->kind_last == ESQ_EPI_SOLERR,
                                         chain->nu, split))
            return ESQ_ENOTSUP;
        // (several fields in one wave: no kernel forms its own input)
        if (chain->from_rows && !(split || NF == 1)) return ESQ_ENOTSUP;
        // a chain in the MIDDLE of a step that forms its own input (stage kind, 4+
        // memory rows) pays only where a wave carries one field of several: the
        // one-field heat sweep of that form moves its words at 3.5 TB/s against
        // 5.4 for the plain chain (Pr9, N = 2236: 0.546 vs 0.519 ms/step), the
        // Brusselator's gains (Pr9 1.074 vs 1.158, Ts5 0.302 vs 0.317)
        if (chain->from_rows && chain->kind_last == ESQ_EPI_STAGE && chain->nu >= 4 && !split)
            return ESQ_ENOTSUP;
        int rc_launch = 0;
        auto body = [&](auto ca, auto kind, auto split_c, auto from_c) {
            using CA = decltype(ca);
            constexpr bool kSplit = decltype(split_c)::value;
            constexpr bool kFrom = decltype(from_c)::value && (kSplit || NF == 1);
            if (decltype(from_c)::value && !kFrom) { rc_launch = ESQ_ENOTSUP; return; }
            auto kern = k_chain2d<NF, PERIODIC, CA::kD, CA::kNU, decltype(kind)::value, Fn,
                                  kSplit, kFrom>;
            // (a workgroup holds whole pairs of tiles: one pair of NF-wave tiles, or
            // kBlock / 128 pairs of one-wave tiles)
            const unsigned block = kSplit ? 128u * NF : (unsigned)kBlock;
            static const int wpc = chain_waves_per_cu(kern, block);   // per instantiation
            const GeoPairs g = geo_chain_pairs(N, CA::kD, wpc, kSplit ? 2 : kBlock / 64,
                                               kSplit ? NF : 1, tune, tall_tiles,
                                               /*min_rows=*/2);
            (void)min_rows;
            if (decltype(kind)::value == ESQ_EPI_SOLERR ||
                decltype(kind)::value == ESQ_EPI_ERRNORM) {
                if ((int)g.grid > chain->partials_cap) { rc_launch = ESQ_ENOTSUP; return; }
                if (chain->partials_used) *chain->partials_used = (int)g.grid;
            }
            if (chain->read_amplification)
                *chain->read_amplification =
                    (double)(g.R + (CA::kD - 1) + (kFrom ? 2 : 0)) / g.R * 64.0 /
                    (64 - 2 * (CA::kD - 1));
            hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(block), 0, (hipStream_t)stream,
                                  (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, y_in, ca,
                                  fn, N, g.npr, g.tpr, g.ntiles, g.nblocks, (unsigned)kXcd);
        };
        // (several fields in one wave: the register budget ends at depth 4)
        constexpr int kDeepest = (split || NF == 1) ? 6 : 4;
        const int rc = dispatch_chain<(HI < kDeepest ? HI : kDeepest), LO>(
            chain, [&](auto ca, auto kind, auto from_c) {
                body(ca, kind, std::integral_constant<bool, split>{}, from_c); });
        if (rc || chain->dry_run) return rc;
        return rc_launch ? rc_launch : (int)hipGetLastError();
    }
};

}  // namespace esq
