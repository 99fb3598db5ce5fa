// esq_rhs_diff3d.hip -- 3-D diffusion, 7-point Laplacian, Dirichlet 0
// (BASELINE.json configs[3], the SSV2stab workload).
#include "esq_rhs_common.hpp"
#include "esq_rkc3d.hpp"

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
// MODE 0: f = rhs(u);  1: Chebyshev recursion (EpiRkc, f not stored);
// 2: end of a Chebyshev step (EpiRkcErr): f stored + error partial sums
constexpr int kPlain = 0, kRkc = 1, kRkcErr = 2;
template <int R, int MODE>
__global__ __launch_bounds__(kBlock) void k_diff3d_v2(
    const double *__restrict__ u, double *__restrict__ f, int N, double c,
    unsigned nblocks, unsigned bpp, RkcEpi epi, esq::EpiRkcErr err) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const int i0 = (int)(lb / bpp) * R;
    const unsigned p = (lb % bpp) * kBlock + threadIdx.x;     // plane index
    const unsigned NN = (unsigned)N * (unsigned)N;
    double local = 0.0;
    if (i0 >= N) {                                            // block-uniform
        if (MODE == kRkcErr) esq::block_partial(local, err.red.partials);
        return;
    }
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
        double c0 = esq::lane_left(centre), c1 = esq::lane_right(centre);
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
            if (MODE == kRkc) {
                epi.out[k] = epi.one(centre, epi.yjm2[k], epi.yn[k], epi.fn[k], fy);
            } else {
                f[k] = fy;
                if (MODE == kRkcErr) {
                    const double b = err.yn[k];
                    const double er = err.one(centre, b, err.fn[k], fy);
                    const double at = err.red.atol_vec ? err.red.atol_vec[k]
                                                       : err.red.atol_s;
                    const double sc =
                        at + err.red.rtol * esq::pmax(fabs(centre), fabs(b));
                    const double q = er / sc;
                    local += q * q;
                }
            }
        }
        below = centre;
        centre = above;
    }
    if (MODE == kRkcErr) esq::block_partial(local, err.red.partials);
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
template <int D, int JT, int NW>
int launch_rkc3d(const Rhs *r, const esq_rkc_chain *ch, hipStream_t stream,
                 hipEvent_t e0, hipEvent_t e1) {
    auto kern = esq::k_rkc3d_chain<D, JT, NW, Diff3dSt>;
    static int slots = 0;                    // workgroups resident on the chip
    if (slots == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, 0) !=
                hipSuccess || per_cu < 1)
            per_cu = 1;
        slots = 256 * per_cu;
    }
    if (NW * JT - 2 * D < 1) return ESQ_ENOTSUP;
    const esq::Geo3d g = esq::geo_rkc3d(r->N, D, JT, NW, slots, r->rkc_planes);
    esq::Rkc3dArgs<D> a;
    a.a = ch->yjm1; a.b = ch->yjm2; a.yn = ch->yn; a.fn = ch->fn;
    a.out = ch->out; a.outp = ch->out_prev;
    for (int k = 0; k < D; ++k) {
        a.mu[k] = ch->mu[k]; a.nu[k] = ch->nu[k]; a.omn[k] = ch->omn[k];
        a.hmus[k] = ch->hmus[k]; a.ajm1[k] = ch->ajm1[k];
    }
    if (ch->read_amplification) *ch->read_amplification = esq::amp_rkc3d(g, D);
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(64 * NW), 0, stream, e0, e1, 0, a,
                          Diff3dSt{c}, g);
    return (int)hipGetLastError();
}
#define ESQ_RKC_SHAPE(DD, JJ, WW) \
    if (jt == JJ && nw == WW) return launch_rkc3d<DD, JJ, WW>(r, ch, stream, e0, e1);
template <int D>
int launch_rkc3d_d(const Rhs *r, const esq_rkc_chain *ch, hipStream_t stream,
                   hipEvent_t e0, hipEvent_t e1) {
    // defaults (N = 159 / 400, ms per step, tools/rkc_shape_sweep.sh): depth 4 as
    // sixteen waves of two rows (four waves per SIMD) 1.35 / 20.7, as eight waves of
    // four rows 1.43 / 22.4; depth 3 on 5 x 8 1.56 / 21.4; depth 2 on 4 x 8 1.92
    constexpr int djt = D == 2 ? 4 : D == 3 ? 5 : D == 4 ? 2 : D == 5 ? 4 : 3;
    constexpr int dnw = D == 4 ? 16 : 8;
    const int jt = r->rkc_jt > 0 ? r->rkc_jt : djt, nw = r->rkc_nw > 0 ? r->rkc_nw : dnw;
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

}  // namespace

extern "C" {

// D Chebyshev stages per launch (esq_rhs_rkc_chain_fn).  Grids below 48^3 stay
// with one launch per stage (a tile's run-in planes and halo points outweigh the
// saving); ESQ_RKC_FORCE=1 when the plugin object is made lifts the rule (tests).
int esq_rhs_diff3d_rkc_chain(void *user, const esq_rkc_chain *ch, size_t n, void *stream,
                             void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !ch) return ESQ_EINVAL;
    if (!ch->yjm1 || !ch->yjm2 || !ch->yn || !ch->fn || !ch->out) return ESQ_EINVAL;
    if (r->N < 2 || (r->N < 48 && !r->rkc_force)) return ESQ_ENOTSUP;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    switch (ch->depth) {
        case 2: return launch_rkc3d_d<2>(r, ch, s, e0, e1);
        case 3: return launch_rkc3d_d<3>(r, ch, s, e0, e1);
        case 4: return launch_rkc3d_d<4>(r, ch, s, e0, e1);
        case 5: return launch_rkc3d_d<5>(r, ch, s, e0, e1);
        case 6: return launch_rkc3d_d<6>(r, ch, s, e0, e1);
        default: return ESQ_ENOTSUP;
    }
}

int esq_rhs_diff3d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = DIFF3D; r.N = N; r.n = (size_t)N * N * N;
    r.rkc_force = getenv("ESQ_RKC_FORCE") ? atoi(getenv("ESQ_RKC_FORCE")) : 0;
    r.rkc_planes = getenv("ESQ_RKC_PLANES") ? atoi(getenv("ESQ_RKC_PLANES")) : 0;
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
    // 2.20 ms at R = 2 against 2.47-2.60 at R = 8).  ESQ_DIFF3D_R overrides.
    static const int Rsel = getenv("ESQ_DIFF3D_R") ? atoi(getenv("ESQ_DIFF3D_R")) : 2;
    const unsigned NN = (unsigned)r->N * (unsigned)r->N;
    const unsigned bpp = (NN + kBlock - 1) / kBlock;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    auto go = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        const unsigned nb = bpp * (unsigned)((r->N + R - 1) / R);
        const unsigned grid = ((nb + kXcd - 1) / kXcd) * kXcd;
        hipExtLaunchKernelGGL((k_diff3d_v2<R, kRkc>), dim3(grid), dim3(kBlock), 0,
                              (hipStream_t)stream, (hipEvent_t)start_event,
                              (hipEvent_t)stop_event, 0, yjm1, (double *)nullptr, r->N,
                              c, grid, bpp,
                              make_epi(yjm2, yn, fn, mu, nu, omn, hmus, ajm1, y_out),
                              esq::EpiRkcErr{});
    };
    switch (Rsel) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 16: go(std::integral_constant<int, 16>{}); break;
        case 32: go(std::integral_constant<int, 32>{}); break;
        default: go(std::integral_constant<int, 8>{}); break;
    }
    return (int)hipGetLastError();
}
// fused entry: only the end of a Chebyshev step (ESQ_EPI_RKCERR) is fused here
int esq_rhs_diff3d_fused(void *user, double t, const double *y_in, double *f,
                         const esq_epilogue *epi, size_t n, void *stream,
                         void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIFF3D || n != r->n || !epi) return ESQ_EINVAL;
    if (epi->kind != ESQ_EPI_RKCERR || r->N < 2 || epi->in_row) return ESQ_ENOTSUP;
    if (epi->is_complex || !epi->rows[0] || !epi->rows[1] || !epi->partials)
        return ESQ_EINVAL;
    constexpr int R = 8;
    const unsigned NN = (unsigned)r->N * (unsigned)r->N;
    const unsigned bpp = (NN + kBlock - 1) / kBlock;
    const unsigned nb = bpp * (unsigned)((r->N + R - 1) / R);
    const unsigned grid = ((nb + kXcd - 1) / kXcd) * kXcd;
    if ((int)grid > epi->partials_cap) return ESQ_ENOTSUP;
    if (epi->partials_used) *epi->partials_used = (int)grid;
    const double c = (double)(r->N + 1) * (double)(r->N + 1);
    hipExtLaunchKernelGGL((k_diff3d_v2<R, kRkcErr>), dim3(grid), dim3(kBlock), 0,
                          (hipStream_t)stream, (hipEvent_t)start_event,
                          (hipEvent_t)stop_event, 0, y_in, f, r->N, c, grid, bpp,
                          RkcEpi{}, esq::make_rkcerr(epi));
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
        hipLaunchKernelGGL((k_diff3d_v2<R, kPlain>), dim3(grid), dim3(kBlock), 0,
                           (hipStream_t)stream, y, f, r->N, c, grid, bpp, RkcEpi{},
                           esq::EpiRkcErr{});
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
