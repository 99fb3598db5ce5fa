// esq_rhs_common.hpp -- shared by the built-in device RHS plugins (the synthetic
// workloads named in BASELINE.json `configs`).  They stand where
// the user's Python callable `fun(t, y)` stands in the reference
// (common.py:356); their NumPy twins, used by the tests, are in
// oracle/problems.py and use the same operation order (the library is built
// with -ffp-contract=off), so the two agree bit for bit.
//
// All kernels are stencil sweeps: one HBM read + one HBM write per element is
// the floor ("RHS-min" in BASELINE.md); neighbour reuse is served by L1/L2.
// Workgroups are dealt round-robin over the 8 XCDs, so block b is remapped to
// a contiguous band of rows per XCD (blockIdx % 8 = XCD label): the up/down
// neighbour rows then hit the SAME XCD's L2 instead of being fetched twice.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/extensisq_amd.h"
#include "esq_epilogue.hpp"
#include "esq_plugin.hpp"
#include "esq_terms.hpp"

namespace esq_rhs {

constexpr int kBlock = 256;
constexpr int kXcd = 8;

enum Kind { DIAG = 1, HEAT2D = 2, BRUSS2D = 3, DIFF3D = 4, CDIAG = 5 };

struct Rhs {
    int kind;
    int N;
    int device;
    double alpha, a, b;
    double amp;
    double *lam_dev;
    size_t n;
    // tuning / test knobs, read from the environment ONCE when the plugin object is
    // made (ESQ_RKC_FORCE: chain sweeps on grids of any size; ESQ_RKC_PLANES:
    // planes per tile of the 3-D chain sweeps, 0 = chosen by geo_rkc3d)
    int rkc_force, rkc_planes, rkc_jt, rkc_nw, diff3d_r;
};

// band remap: logical block id such that XCD x (label blockIdx%8) sweeps the
// contiguous range [x*per, (x+1)*per)
__device__ __forceinline__ unsigned band_block(unsigned b, unsigned nblocks) {
    const unsigned per = (nblocks + kXcd - 1) / kXcd;
    return (b % kXcd) * per + b / kXcd;
}

// ESQ_RHS_VARIANT=1: scalar kernels instead of the vectorised sweeps (tests)
inline int rhs_variant() {
    static const int v = getenv("ESQ_RHS_VARIANT") ? atoi(getenv("ESQ_RHS_VARIANT")) : 0;
    return v;
}

inline int make(void **out, Rhs proto) {
    if (!out) return ESQ_EINVAL;
    Rhs *r = (Rhs *)malloc(sizeof(Rhs));
    if (!r) return ESQ_ENOMEM;
    *r = proto;
    *out = r;
    return 0;
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
        lf = esq::lane_left(c.y);
        rt = esq::lane_right(c.x);
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
template <> inline constexpr bool kFirstStage<esq::EpiStage<0>> = true;
template <> inline constexpr bool kFirstStage<esq::EpiStage<1>> = true;
inline bool first_stage_ok(const esq_epilogue *e) {
    return e->kind == ESQ_EPI_STAGE && e->nt <= 1 && e->in_base;
}
inline SrcAxpy axpy_of(const esq_epilogue *e) {
    return SrcAxpy{e->in_base, e->in_row, e->in_c, e->in_h};
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
// a chain of depth D carries 2(D-1) halo rows per tile and marches row by row: on
// grids below ~450 x 450 a step is bound by its launches and D single sweeps are
// as fast or faster (tools/small_chain_sweep.py: Brusselator Pr8 at N = 316 78 us
// unchained, 87 with depth-2 chains; at N = 500 100 against 87..98 chained; heat
// Pr9 at N = 448 90 against 98); ESQ_CHAIN_ROWS (tests) lifts the rule
inline bool chain_fits_grid(int N, int depth) {
    if (getenv("ESQ_CHAIN_ROWS")) return true;
    const int W = 64 - 2 * (depth - 1);
    const size_t tpr = ((size_t)N / 2 + W - 1) / W;
    return (size_t)N * tpr >= 2048;
}
// alternate tile rows march in opposite directions (esq_chain.hpp);
// ESQ_CHAIN_SERPENTINE=0: all downwards
inline unsigned chain_serpentine() {
    const char *e = getenv("ESQ_CHAIN_SERPENTINE");
    return (e && atoi(e) == 0) ? 0u : 1u;
}
// tiles_per_block: wave tiles a workgroup works on; waves_per_tile: waves that
// share one tile (the split sweeps: one per field)
inline GeoChain geo_chain(int N, int depth, int waves_per_cu, int tiles_per_block,
                          int waves_per_tile, bool tall_if_one_round = false,
                          int min_rows = 0) {
    GeoChain g;
    const int W = 64 - 2 * (depth - 1);
    g.tpr = ((unsigned)N / 2 + W - 1) / W;
    const char *env = getenv("ESQ_CHAIN_ROWS");           // tuning / tests
    int R = env ? atoi(env) : 0;
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
        if (tall_if_one_round) {
            const int cand = rows_for((size_t)256 * 4);
            if (cand >= 24 && cand <= 48) R = cand;
        }
        size_t rounds = 1;
        if (R <= 0) R = rows_for((size_t)256 * (size_t)waves_per_cu);
        while (R > 48) {
            ++rounds;
            R = rows_for((size_t)256 * (size_t)waves_per_cu * rounds);
        }
        // (the halo rows are recomputed: the Brusselator's heavier rows want depth + 2,
        // the heat sweeps fill the wave slots down to `depth` rows -- Ts5 at N = 1000:
        // chain5<1> 31 us on 7-row tiles, 27 on 5-row tiles, 36 on 4-row tiles)
        if (min_rows <= 0) min_rows = depth + 2;
        if (R < min_rows) R = min_rows;
    }
    if (R > N) R = N;
    g.R = R;
    g.ntiles = g.tpr * (unsigned)((N + R - 1) / R);
    g.nblocks = (g.ntiles + tiles_per_block - 1) / tiles_per_block;
    g.grid = ((g.nblocks + kXcd - 1) / kXcd) * kXcd;
    return g;
}

inline RkcEpi make_epi(const double *yjm2, const double *yn, const double *fn,
                       double mu, double nu, double omn, double hmus, double ajm1,
                       double *out) {
    RkcEpi e{};
    e.yjm2 = yjm2; e.yn = yn; e.fn = fn; e.out = out;
    e.mu = mu; e.nu = nu; e.omn = omn; e.hmus = hmus; e.ajm1 = ajm1;
    return e;
}

}  // namespace esq_rhs
