// esq_rhs_diff3d.hip -- 3-D diffusion, 7-point Laplacian, Dirichlet 0
// (BASELINE.json configs[3], the SSV2stab workload).
#include "esq_rhs_common.hpp"
#include "esq_rkc3d.hpp"
#include "esq_chain3d.hpp"

using namespace esq_rhs;

namespace {

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

// 3-D diffusion, marching version: a thread owns one (j, l) column of the grid
// (flattened plane index p) and walks R planes along i with a rolling
// (below, centre, above) window; the l-neighbours come from adjacent lanes,
// the j-neighbours are two coalesced loads of the centre plane.  3 loads per
// output instead of 7; arithmetic order identical to k_diff3d.
// Epi: what happens to the fresh derivative while it is in a register
// (esq_epilogue.hpp, the one-double-per-thread twins -- N may be odd, so no
// 16-byte accesses): EpiNone plain RHS, EpiStage / EpiBlock / EpiSolErr /
// EpiErrNorm the Runge-Kutta arithmetic that follows a stage evaluation,
// EpiRkc the Chebyshev recursion (f not stored), EpiRkcErr the end of a
// Chebyshev step.
template <int R, class Epi>
__global__ __launch_bounds__(kBlock) void k_diff3d_sweep(
    const double *__restrict__ u, double *__restrict__ f, int N, double c,
    unsigned nblocks, unsigned bpp, Epi epi) {
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
                double c0 = esq::lane_left(centre), c1 = esq::lane_right(centre);
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
                    const double fy =
                        c * ((((below + above) + (b0 + b1)) + (c0 + c1)) - 6.0 * centre);
                    epi.store_f1(f, k, fy);
                    epi.finish1(in, fy, centre, k, local);
                }
                below = centre;
                centre = above;
            }
        }
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}

// ---------------------------------------------------------------------------
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
// of the flattened state).  Same expression, same order as k_diff3d.
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));
template <class Epi>
__global__ __launch_bounds__(kBlock) void k_diff3d_pairs(
    const double *__restrict__ u, double *__restrict__ f, int N, double c,
    unsigned nblocks, size_t n, Epi epi) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const size_t q = (size_t)lb * kBlock + threadIdx.x;          // pair index
    const size_t e0 = 2 * q;
    const unsigned NN = (unsigned)N * (unsigned)N;
    double local = 0.0;
    const bool live0 = e0 < n, live1 = e0 + 1 < n;
    const int lane = threadIdx.x & 63;
    // centre pair first: the lane shifts need it from every lane of the wave
    double2 cc = make_double2(0.0, 0.0);
    if (live0) cc = esq::ld2(u, q);                                // (padding is zero)
    double lf = esq::lane_left(cc.y), rt = esq::lane_right(cc.x);
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
        fy.x = c * ((((xb + xa) + (x0 + x1)) + (xl + xr)) - 6.0 * cc.x);
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
            fy.y = c * ((((yb + ya) + (y0 + y1)) + (yl + yr)) - 6.0 * cc.y);
        }
        epi.store_f(f, q, fy);
        epi.finish(in, fy, cc, q, local);
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}
inline unsigned grid_diff3d_pairs(const Rhs *r) {
    const size_t pairs = (r->n + 1) / 2;
    const unsigned nb = (unsigned)((pairs + kBlock - 1) / kBlock);
    return ((nb + kXcd - 1) / kXcd) * kXcd;
}
template <class Epi>
void launch_diff3d_pairs(const Rhs *r, const double *y_in, double *f, const Epi &epi,
                         hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    const unsigned grid = grid_diff3d_pairs(r);
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL((k_diff3d_pairs<Epi>), dim3(grid), dim3(kBlock), 0, stream, e0, e1,
                          0, y_in, f, r->N, c, grid, r->n, epi);
}

template <int R, class Epi>
void launch_diff3d(const Rhs *r, const double *y_in, double *f, const Epi &epi,
                   hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    const unsigned NN = (unsigned)r->N * (unsigned)r->N;
    const unsigned bpp = (NN + kBlock - 1) / kBlock;            // blocks per plane
    const unsigned nb = bpp * (unsigned)((r->N + R - 1) / R);
    const unsigned grid = ((nb + kXcd - 1) / kXcd) * kXcd;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL((k_diff3d_sweep<R, Epi>), dim3(grid), dim3(kBlock), 0, stream, e0,
                          e1, 0, y_in, f, r->N, c, grid, bpp, epi);
}
template <int R>
unsigned grid_diff3d(const Rhs *r) {
    const unsigned NN = (unsigned)r->N * (unsigned)r->N;
    const unsigned bpp = (NN + kBlock - 1) / kBlock;
    const unsigned nb = bpp * (unsigned)((r->N + R - 1) / R);
    return ((nb + kXcd - 1) / kXcd) * kXcd;
}

// the seven-point Laplacian as the marching chain sweep sees it (esq_rkc3d.hpp):
// the expression of k_diff3d / k_diff3d_v2, operation for operation
struct Diff3dSt {
    double c;
    __device__ __forceinline__ double eval(double below, double above, double up,
                                           double dn, double lf, double rt,
                                           double centre) const {
        return c * ((((below + above) + (up + dn)) + (lf + rt)) - 6.0 * centre);
    }
};

// rows per thread and waves per workgroup by depth: the windows, the y_n / f_n
// delay lines and one plane of operands in flight are (4 D + 7) JT doubles per
// thread.  ESQ_RKC_CFG="JT,NW" (read when the plugin object is made) picks another
// instantiated shape (tuning).
// D: stage slots of the sweep (LAST: the chain's depth + 1)
template <int D, int JT, int NW, bool FIRST = false, bool LAST = false>
int launch_rkc3d(const Rhs *r, const esq_rkc_chain *ch, hipStream_t stream,
                 hipEvent_t e0, hipEvent_t e1) {
    auto kern = esq::k_rkc3d_chain<D, JT, NW, Diff3dSt, FIRST, LAST>;
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
    const int slots = esq::device_cus() * per_cu;
    if (NW * JT - 2 * D < 1) return ESQ_ENOTSUP;
    const esq::Geo3d g = esq::geo_rkc3d(r->N, D, JT, NW, slots, r->rkc_planes);
    esq::Rkc3dArgs<D> a;
    a.a = ch->yjm1; a.b = ch->yjm2; a.yn = ch->yn; a.fn = ch->fn;
    a.out = ch->out; a.outp = ch->out_prev;
    a.hmus1 = ch->hmus_first;
    a.h04 = 0.0;
    a.red = esq::RedArgs{};
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
    if (ch->read_amplification) *ch->read_amplification = esq::amp_rkc3d(g, D);
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(64 * NW), 0, stream, e0, e1, 0, a,
                          Diff3dSt{c}, g);
    return (int)hipGetLastError();
}
#define ESQ_RKC_SHAPE(DD, JJ, WW) \
    if (jt == JJ && nw == WW) return launch_rkc3d<DD, JJ, WW>(r, ch, stream, e0, e1);
// the form that opens a step (ch->yjm1 == NULL) exists on each depth's default shape
#define ESQ_RKC_SHAPE_FIRST(DD, JJ, WW)                                               \
    if (!ch->yjm1) {                                                                  \
        if (jt == JJ && nw == WW && !ch->fy_out)                                      \
            return launch_rkc3d<DD, JJ, WW, true>(r, ch, stream, e0, e1);             \
        return ESQ_ENOTSUP;                                                           \
    }                                                                                 \
    if (ch->fy_out) {                     /* LAST: DD = the chain's depth + 1 */      \
        if constexpr (DD >= 3 && DD <= 5) {                                           \
            if (jt == JJ && nw == WW)                                                 \
                return launch_rkc3d<DD, JJ, WW, false, true>(r, ch, stream, e0, e1);  \
        }                                                                             \
        return ESQ_ENOTSUP;                                                           \
    }
template <int D>
int launch_rkc3d_d(const Rhs *r, const esq_rkc_chain *ch, hipStream_t stream,
                   hipEvent_t e0, hipEvent_t e1) {
    // defaults (N = 159 / 400, ms per step, tools/rkc_shape_sweep.sh): depth 4 as
    // sixteen waves of two rows (four waves per SIMD) 1.35 / 20.7, as eight waves of
    // four rows 1.43 / 22.4; depth 3 on 5 x 8 1.56 / 21.4; depth 2 on 4 x 8 1.92
    constexpr int djt = D == 2 ? 4 : D == 3 ? 5 : D == 4 ? 2 : D == 5 ? 4 : 3;
    constexpr int dnw = D == 4 ? 16 : 8;
    const int jt = r->rkc_jt > 0 ? r->rkc_jt : djt, nw = r->rkc_nw > 0 ? r->rkc_nw : dnw;
    ESQ_RKC_SHAPE_FIRST(D, djt, dnw)
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
    return ESQ_ENOTSUP;
}
#undef ESQ_RKC_SHAPE
#undef ESQ_RKC_SHAPE_FIRST

}  // namespace

extern "C" {

// D Chebyshev stages per launch (esq_rhs_rkc_chain_fn).  Grids below 48^3 stay
// with one launch per stage (a tile's run-in planes and halo points outweigh the
// saving); ESQ_RKC_FORCE=1 when the plugin object is made lifts the rule (tests).
int esq_rhs_diff3d_rkc_chain(void *user, const esq_rkc_chain *ch, size_t n, void *stream,
                             void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !ch) return ESQ_EINVAL;
    if (!ch->yjm2 || !ch->yn || !ch->fn || !ch->out) return ESQ_EINVAL;
    if (!ch->yjm1 && ch->yjm2 != ch->yn) return ESQ_EINVAL;       // FIRST: y_{j-2} = y_n
    if (ch->fy_out && (ch->out_prev || !ch->partials)) return ESQ_EINVAL;
    // (32-bit byte offsets into a vector: esq_rkc3d.hpp)
    if ((unsigned long long)n * 8ull > 0xffffffffull - 16ull) return ESQ_ENOTSUP;
    if (r->N < 2 || (r->N < 48 && !r->rkc_force)) return ESQ_ENOTSUP;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    switch (ch->depth + (ch->fy_out ? 1 : 0)) {           // stage slots of the sweep
        case 2: return launch_rkc3d_d<2>(r, ch, s, e0, e1);
        case 3: return launch_rkc3d_d<3>(r, ch, s, e0, e1);
        case 4: return launch_rkc3d_d<4>(r, ch, s, e0, e1);
        case 5: return launch_rkc3d_d<5>(r, ch, s, e0, e1);
        case 6: return launch_rkc3d_d<6>(r, ch, s, e0, e1);
        default: return ESQ_ENOTSUP;
    }
}

// D consecutive Runge-Kutta stages per launch (esq_rhs_chain_fn, esq_chain3d.hpp)
int esq_rhs_diff3d_chain(void *user, const double *y_in, const esq_chain *chain, size_t n,
                         void *stream, void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !chain) return ESQ_EINVAL;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    return esq::chain3d(Diff3dSt{c}, r->N, y_in, chain, r->rkc_planes, r->rkc_force != 0,
                        stream, start_event, stop_event);
}

int esq_rhs_diff3d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = DIFF3D; r.N = N; r.n = (size_t)N * N * N;
    r.rkc_force = getenv("ESQ_RKC_FORCE") ? atoi(getenv("ESQ_RKC_FORCE")) : 0;
    r.rkc_planes = getenv("ESQ_RKC_PLANES") ? atoi(getenv("ESQ_RKC_PLANES")) : 0;
    r.diff3d_r = getenv("ESQ_DIFF3D_R") ? atoi(getenv("ESQ_DIFF3D_R")) : 0;
    r.rkc_jt = r.rkc_nw = 0;
    if (const char *e = getenv("ESQ_RKC_CFG")) sscanf(e, "%d,%d", &r.rkc_jt, &r.rkc_nw);
    return make(user_out, r);
}

int esq_rhs_diff3d_rkc(void *user, double t, const double *yjm1, const double *yjm2,
                       const double *yn, const double *fn, double mu, double nu,
                       double omn, double hmus, double ajm1, double *y_out,
                       size_t n, void *stream, void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    if (r->N < 2) return ESQ_ENOTSUP;
    // planes per workgroup: with the five vectors of a Chebyshev stage resident in
    // the Infinity Cache (n = 4e6: 160 MB) short marches win -- more workgroups in
    // flight, the re-read planes are cache hits (N = 159, us per stage: R = 1 23.2,
    // 2 22.5-23.4, 3 23.0-23.6, 4 23.4-24.7, 8 24.6-24.9, 16 23.3, 32 29.9; the step
    // 2.20 ms at R = 2 against 2.47-2.60 at R = 8).  ESQ_DIFF3D_R (read when the
    // plugin object is made): 0 = the 16-byte pair sweep (default), 1 / 2 / 4 / 8 =
    // the marching sweep with that many planes per workgroup.
    const RkcEpi epi = make_epi(yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    switch (r->diff3d_r) {
        case 0: launch_diff3d_pairs(r, yjm1, nullptr, epi, s, e0, e1); break;
        case 1: launch_diff3d<1>(r, yjm1, nullptr, epi, s, e0, e1); break;
        case 2: launch_diff3d<2>(r, yjm1, nullptr, epi, s, e0, e1); break;
        case 4: launch_diff3d<4>(r, yjm1, nullptr, epi, s, e0, e1); break;
        default: launch_diff3d<8>(r, yjm1, nullptr, epi, s, e0, e1); break;
    }
    return (int)hipGetLastError();
}
// fused entry: the Runge-Kutta arithmetic that follows a stage evaluation (and
// the end of a Chebyshev step) inside the sweep, every epilogue kind; the
// on-the-fly first-stage input (ESQ_FUSE_SRC) is not offered
int esq_rhs_diff3d_fused(void *user, double t, const double *y_in, double *f,
                         const esq_epilogue *epi, size_t n, void *stream,
                         void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !epi) return ESQ_EINVAL;
    if (r->N < 2 || epi->in_row || epi->is_complex) return ESQ_ENOTSUP;
    constexpr int R = 8;
    const bool pairs = r->diff3d_r == 0;
    const unsigned grid = pairs ? grid_diff3d_pairs(r) : grid_diff3d<R>(r);
    if (esq::epilogue_reduces(epi)) {
        if ((int)grid > epi->partials_cap) return ESQ_ENOTSUP;
        if (epi->partials_used && !epi->dry_run) *epi->partials_used = (int)grid;
    }
    const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
        if (pairs)
            launch_diff3d_pairs(r, y_in, f, ep, (hipStream_t)stream,
                                (hipEvent_t)start_event, (hipEvent_t)stop_event);
        else
            launch_diff3d<R>(r, y_in, f, ep, (hipStream_t)stream, (hipEvent_t)start_event,
                             (hipEvent_t)stop_event);
    });
    return (rc || epi->dry_run) ? rc : (int)hipGetLastError();
}
int esq_rhs_diff3d(void *user, double t, const double *y, double *f, size_t n,
                   void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n) return ESQ_EINVAL;
    if (rhs_variant() != 1 && r->N >= 2) {
        esq::EpiNone none{};
        if (r->diff3d_r == 0)
            launch_diff3d_pairs(r, y, f, none, (hipStream_t)stream, nullptr, nullptr);
        else
            launch_diff3d<8>(r, y, f, none, (hipStream_t)stream, nullptr, nullptr);
        return (int)hipGetLastError();
    }
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    hipLaunchKernelGGL(k_diff3d, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       y, f, r->N, c, grid, bpr);
    return (int)hipGetLastError();
}

}  // extern "C"
