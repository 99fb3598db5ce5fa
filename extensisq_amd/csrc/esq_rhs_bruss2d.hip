// esq_rhs_bruss2d.hip -- 2-D Brusselator reaction-diffusion, periodic
// (BASELINE.json configs[2], the north-star workload).
#include "esq_rhs_common.hpp"

using namespace esq_rhs;

namespace {

// 2-D Brusselator, periodic.  y = [u.ravel(), v.ravel()]
//   du = (A + u*u*v - (B+1)*u) + d*lap(u);  dv = (B*u - u*u*v) + d*lap(v)
template <bool NTS>
__global__ __launch_bounds__(kBlock) void k_bruss2d(
    const double *__restrict__ y, double *__restrict__ f, int N, double d,
    double A, double B, unsigned nblocks, unsigned bpr) {
    const unsigned lb = band_block(blockIdx.x, nblocks);
    const unsigned i = lb / bpr;
    const unsigned j = (lb % bpr) * kBlock + threadIdx.x;
    if (i >= (unsigned)N || j >= (unsigned)N) return;
    const size_t NN = (size_t)N * N;
    const double *__restrict__ u = y;
    const double *__restrict__ v = y + NN;
    const unsigned im = i == 0 ? N - 1 : i - 1, ip = i + 1 == (unsigned)N ? 0 : i + 1;
    const unsigned jm = j == 0 ? N - 1 : j - 1, jp = j + 1 == (unsigned)N ? 0 : j + 1;
    const size_t k = (size_t)i * N + j;
    const size_t kup = (size_t)im * N + j, kdn = (size_t)ip * N + j;
    const size_t klf = (size_t)i * N + jm, krt = (size_t)i * N + jp;
    const double uc = u[k], vc = v[k];
    const double lapu = ((u[kup] + u[kdn]) + (u[klf] + u[krt])) - 4.0 * uc;
    const double lapv = ((v[kup] + v[kdn]) + (v[klf] + v[krt])) - 4.0 * vc;
    const double uuv = uc * uc * vc;
    const double fu = ((A + uuv) - (B + 1.0) * uc) + d * lapu;
    const double fv = (B * uc - uuv) + d * lapv;
    if (NTS) {
        __builtin_nontemporal_store(fu, f + k);
        __builtin_nontemporal_store(fv, f + NN + k);
    } else {
        f[k] = fu;
        f[NN + k] = fv;
    }
}

// ---------------------------------------------------------------------------
// SWEEPS.  One wave tile = 64 column pairs of ONE grid row; all three window
// rows are requested up front, together with the epilogue's operands, so every
// load of the thread is in flight before the first use.  `Epi` (esq_epilogue.hpp)
// says what happens to the fresh derivative: store only (EpiNone), next stage
// argument (EpiStage), blocked accumulation (EpiBlock), solution + error norm
// (EpiSolErr), FSAL error norm (EpiErrNorm), Chebyshev recursion (EpiRkc).
// The epilogues are pointwise: nothing is recomputed on halos.
// ---------------------------------------------------------------------------
template <class Epi, class Src>
__global__ __launch_bounds__(kBlock) void k_bruss2d_sweep(
    Src ys, double *__restrict__ f, Epi epi, int N,
    double d, double A, double B, unsigned nblocks, unsigned wpr) {
    const unsigned tile = band_block(blockIdx.x, nblocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int i = (int)(tile / wpr);
    double local = 0.0;
    if (i < N) {                                           // wave-uniform
        const size_t NN = (size_t)N * N;
        RowWin<true, Src> U, V;
        U.src = V.src = ys;
        U.base = 0; V.base = NN;
        U.N = V.N = N;
        U.npairs = V.npairs = (unsigned)N / 2;
        U.pair = V.pair = (tile % wpr) * 64 + (threadIdx.x & 63);
        U.live = V.live = U.pair < U.npairs;
        const size_t k2 = ((size_t)i * N) / 2 + (U.live ? U.pair : 0);   // N even
        const size_t v2 = NN / 2 + k2;
        typename Epi::In cu, cv;
        epi.load(cu, k2);
        epi.load(cv, v2);
        const double2 uu = U.row(i - 1), uc = U.row(i), ud = U.row(i + 1);
        const double2 vu = V.row(i - 1), vc = V.row(i), vd = V.row(i + 1);
        double ul, urt, vl, vrt;
        U.sides(i, uc, ul, urt);
        V.sides(i, vc, vl, vrt);
        double2 fu, fv;
        {
            const double lapx = ((uu.x + ud.x) + (ul + uc.y)) - 4.0 * uc.x;
            const double lapy = ((uu.y + ud.y) + (uc.x + urt)) - 4.0 * uc.y;
            const double lvx = ((vu.x + vd.x) + (vl + vc.y)) - 4.0 * vc.x;
            const double lvy = ((vu.y + vd.y) + (vc.x + vrt)) - 4.0 * vc.y;
            const double uuvx = uc.x * uc.x * vc.x, uuvy = uc.y * uc.y * vc.y;
            fu.x = ((A + uuvx) - (B + 1.0) * uc.x) + d * lapx;
            fu.y = ((A + uuvy) - (B + 1.0) * uc.y) + d * lapy;
            fv.x = (B * uc.x - uuvx) + d * lvx;
            fv.y = (B * uc.y - uuvy) + d * lvy;
        }
        if (U.live) {
            epi.store_f(f, k2, fu);
            epi.store_f(f, v2, fv);
            epi.finish(cu, fu, uc, k2, local);
            epi.finish(cv, fv, vc, v2, local);
        }
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}

// pointwise part of the Brusselator for the two-stage marching sweep: centres
// and five-point Laplacians of (u, v) -> (du, dv), same operation order as above
struct BrussFn {
    double d, A, B;
    __device__ __forceinline__ void eval(const double2 (&c)[2], const double2 (&lap)[2],
                                         double2 (&f)[2]) const {
        const double uuvx = c[0].x * c[0].x * c[1].x, uuvy = c[0].y * c[0].y * c[1].y;
        f[0].x = ((A + uuvx) - (B + 1.0) * c[0].x) + d * lap[0].x;
        f[0].y = ((A + uuvy) - (B + 1.0) * c[0].y) + d * lap[0].y;
        f[1].x = (B * c[0].x - uuvx) + d * lap[1].x;
        f[1].y = (B * c[0].y - uuvy) + d * lap[1].y;
    }
    // one field only (split chain sweeps: a wave per field), same operations
    __device__ __forceinline__ double2 eval_one(int field, const double2 (&c)[2],
                                                double2 lap) const {
        const double uuvx = c[0].x * c[0].x * c[1].x, uuvy = c[0].y * c[0].y * c[1].y;
        if (field == 0)
            return make_double2(((A + uuvx) - (B + 1.0) * c[0].x) + d * lap.x,
                                ((A + uuvy) - (B + 1.0) * c[0].y) + d * lap.y);
        return make_double2((B * c[0].x - uuvx) + d * lap.x, (B * c[0].y - uuvy) + d * lap.y);
    }
};

}  // namespace

extern "C" {

int esq_rhs_bruss2d_create(void **user_out, int N, double alpha, double a,
                           double b) {
    if (N < 1) return ESQ_EINVAL;
    Rhs r{};
    r.kind = BRUSS2D; r.N = N; r.n = 2 * (size_t)N * N;
    r.alpha = alpha; r.a = a; r.b = b;
    return make(user_out, r);
}

int esq_rhs_bruss2d(void *user, double t, const double *y, double *f, size_t n,
                    void *stream) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n) return ESQ_EINVAL;
    const double d = r->alpha * ((double)r->N * (double)r->N);
    if (r->N % 2 == 0 && r->N >= 4 && rhs_variant() != 1) {
        const Geo2d g = geo2d(r->N);
        esq::EpiNone ep{};
        hipLaunchKernelGGL((k_bruss2d_sweep<esq::EpiNone, SrcPlain>), dim3(g.grid),
                           dim3(kBlock), 0, (hipStream_t)stream, SrcPlain{y}, f, ep, r->N, d, r->a, r->b,
                           g.grid, g.wpr);
        return (int)hipGetLastError();
    }
    const unsigned bpr = (r->N + kBlock - 1) / kBlock;
    unsigned nblocks = bpr * (unsigned)r->N;
    const unsigned grid = ((nblocks + kXcd - 1) / kXcd) * kXcd;
    static const bool nts = getenv("ESQ_RHS_STORE_NT") && atoi(getenv("ESQ_RHS_STORE_NT"));
    if (nts)
        hipLaunchKernelGGL(k_bruss2d<true>, dim3(grid), dim3(kBlock), 0,
                           (hipStream_t)stream, y, f, r->N, d, r->a, r->b, grid, bpr);
    else
        hipLaunchKernelGGL(k_bruss2d<false>, dim3(grid), dim3(kBlock), 0,
                           (hipStream_t)stream, y, f, r->N, d, r->a, r->b, grid, bpr);
    return (int)hipGetLastError();
}

int esq_rhs_bruss2d_fused(void *user, double t, const double *y_in, double *f,
                          const esq_epilogue *epi, size_t n, void *stream,
                          void *start_event, void *stop_event) {
    (void)t;
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n || !epi) return ESQ_EINVAL;
    if (r->N % 2 != 0 || r->N < 4) return ESQ_ENOTSUP;
    const Geo2d g = geo2d(r->N);
    if (esq::epilogue_reduces(epi)) {
        if ((int)g.grid > epi->partials_cap) return ESQ_ENOTSUP;
        if (epi->partials_used) *epi->partials_used = (int)g.grid;
    }
    const double d = r->alpha * ((double)r->N * (double)r->N);
    if (epi->in_row && !first_stage_ok(epi)) return ESQ_ENOTSUP;
    const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
        using E = decltype(ep);
        if constexpr (kFirstStage<E>) {
            if (epi->in_row) {
                hipExtLaunchKernelGGL((k_bruss2d_sweep<E, SrcAxpy>), dim3(g.grid),
                                      dim3(kBlock), 0, (hipStream_t)stream,
                                      (hipEvent_t)start_event, (hipEvent_t)stop_event,
                                      0, axpy_of(epi), f, ep, r->N, d, r->a, r->b,
                                      g.grid, g.wpr);
                return;
            }
        }
        hipExtLaunchKernelGGL((k_bruss2d_sweep<E, SrcPlain>), dim3(g.grid),
                              dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                              SrcPlain{y_in}, f, ep, r->N, d, r->a, r->b, g.grid,
                              g.wpr);
    });
    return rc ? rc : (int)hipGetLastError();
}

int esq_rhs_bruss2d_chain(void *user, const double *y_in, const esq_chain *chain,
                          size_t n, void *stream, void *start_event,
                          void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != BRUSS2D || n != r->n || !chain) return ESQ_EINVAL;
    if (r->N % 2 != 0 || r->N < 16) return ESQ_ENOTSUP;
    if (!chain_fits_grid(r->N, chain->depth)) return ESQ_ENOTSUP;
    // register budget (esq_chain.hpp, ChainCaps): one field per wave keeps two
    // waves per SIMD up to depth 4 with 9 memory rows; ESQ_CHAIN_SPLIT=0 runs
    // both fields in one wave (the first version; narrower caps)
    static const bool split = !getenv("ESQ_CHAIN_SPLIT") || atoi(getenv("ESQ_CHAIN_SPLIT")) != 0;
    if (!esq::chain_within_caps(chain->depth, chain->kind_last == ESQ_EPI_SOLERR,
                                chain->nu, split))
        return ESQ_ENOTSUP;
    const BrussFn fn{r->alpha * ((double)r->N * (double)r->N), r->a, r->b};
    int rc_launch = 0;
    auto body = [&](auto ca, auto kind, auto split_c, auto from_c) {
        using CA = decltype(ca);
        constexpr bool kSplit = decltype(split_c)::value;
        constexpr bool kFrom = decltype(from_c)::value && kSplit;
        if (decltype(from_c)::value && !kSplit) { rc_launch = ESQ_ENOTSUP; return; }
        auto kern = esq::k_chain2d<2, true, CA::kD, CA::kNU, decltype(kind)::value, BrussFn, kSplit,
                                   kFrom>;
        const unsigned block = kSplit ? 128u : (unsigned)kBlock;
        static const int wpc = chain_waves_per_cu(kern, block);       // per instantiation
        const GeoChain g = geo_chain(r->N, CA::kD, wpc, kSplit ? 1 : kBlock / 64, kSplit ? 2 : 1);
        if (decltype(kind)::value == ESQ_EPI_SOLERR) {
            if ((int)g.grid > chain->partials_cap) { rc_launch = ESQ_ENOTSUP; return; }
            if (chain->partials_used) *chain->partials_used = (int)g.grid;
        }
        if (chain->read_amplification)
            *chain->read_amplification = (double)(g.R + 2 * (CA::kD - 1) + (kFrom ? 2 : 0)) /
                                         g.R * 64.0 / (64 - 2 * (CA::kD - 1));
        hipExtLaunchKernelGGL(kern, dim3(g.grid), dim3(block), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, y_in,
                              ca, fn, r->N, g.R, g.tpr, g.ntiles, g.nblocks,
                              (unsigned)kXcd, chain_serpentine());
    };
    const int rc = split
        ? esq::dispatch_chain<6>(chain, [&](auto ca, auto kind, auto from_c) {
              body(ca, kind, std::true_type{}, from_c); })
        : esq::dispatch_chain<4>(chain, [&](auto ca, auto kind, auto from_c) {
              body(ca, kind, std::false_type{}, from_c); });
    if (rc) return rc;
    return rc_launch ? rc_launch : (int)hipGetLastError();
}

}  // extern "C"
