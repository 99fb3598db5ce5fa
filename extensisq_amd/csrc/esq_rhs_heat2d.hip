// esq_rhs_heat2d.hip -- 2-D heat equation, 5-point Laplacian, Dirichlet 0
// (BASELINE.json configs[1], configs[4]).
#include "esq_rhs_common.hpp"

using namespace esq_rhs;

namespace {

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

// pointwise part for the two-stage marching sweep: f = c * laplacian
struct HeatFn {
    double c;
    __device__ __forceinline__ void eval(const double2 (&)[1], const double2 (&lap)[1],
                                         double2 (&f)[1]) const {
        f[0].x = c * lap[0].x;
        f[0].y = c * lap[0].y;
    }
};

}  // namespace

extern "C" {

int esq_rhs_heat2d_create(void **user_out, int N) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = HEAT2D; r.N = N; r.n = (size_t)N * N;
    return make(user_out, r);
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

int esq_rhs_heat2d_chain(void *user, const double *y_in, const esq_chain *chain,
                          size_t n, void *stream, void *start_event,
                          void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != HEAT2D || n != r->n || !chain) return ESQ_EINVAL;
    if (r->N % 2 != 0 || r->N < 16) return ESQ_ENOTSUP;
    if (!chain_fits_grid(r->N, chain->depth)) return ESQ_ENOTSUP;
    const HeatFn fn{(double)(r->N + 1) * (double)(r->N + 1)};
    int rc_launch = 0;
    const int rc = esq::dispatch_chain<6>(chain, [&](auto ca, auto kind, auto from_c) {
        using CA = decltype(ca);
        constexpr bool kFrom = decltype(from_c)::value;
        auto kern = esq::k_chain2d<1, false, CA::kD, CA::kNU, decltype(kind)::value, HeatFn, false,
                                   kFrom>;
        static const int wpc = chain_waves_per_cu(kern, (unsigned)kBlock);   // per instantiation
        const GeoChain g = geo_chain(r->N, CA::kD, wpc, kBlock / 64, 1, /*tall_if_one_round=*/true,
                                     /*min_rows=*/CA::kD);
        if (chain->read_amplification)
            *chain->read_amplification = (double)(g.R + 2 * (CA::kD - 1) + (kFrom ? 2 : 0)) /
                                         g.R * 64.0 / (64 - 2 * (CA::kD - 1));
        if (decltype(kind)::value == ESQ_EPI_SOLERR) {
            if ((int)g.grid > chain->partials_cap) { rc_launch = ESQ_ENOTSUP; return; }
            if (chain->partials_used) *chain->partials_used = (int)g.grid;
        }
        hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, y_in,
                              ca, fn, r->N, g.R, g.tpr, g.ntiles, g.nblocks,
                              (unsigned)kXcd, chain_serpentine());
    });
    if (rc) return rc;
    return rc_launch ? rc_launch : (int)hipGetLastError();
}

}  // extern "C"
