// esq_rhs_diag.hip -- pointwise plugins  f = lam*y + amp*sin(t)  for real and
// complex states (the complex one carries the reference's complex-state tests,
// tests/test_ivp.py:216-259, tests/test_rk.py:92-98, onto the device path).
#include "esq_rhs_common.hpp"

using namespace esq_rhs;

namespace {

// f = lam*y + amp*sin(t)
__global__ __launch_bounds__(kBlock) void k_diag(const double *__restrict__ y,
                                                 double *__restrict__ f,
                                                 const double *__restrict__ lam,
                                                 double forcing, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        f[i] = lam[i] * y[i] + forcing;
}

// complex state: element k is the pair (re, im) at doubles 2k, 2k+1;
// f = lam*y + F with the four real products and two sums of a complex multiply,
// each rounded (NumPy's scalar loop)
__device__ __forceinline__ double2 cmul_add(double2 l, double2 y, double2 F) {
    return make_double2((l.x * y.x - l.y * y.y) + F.x, (l.x * y.y + l.y * y.x) + F.y);
}
__global__ __launch_bounds__(kBlock) void k_cdiag(const double *__restrict__ y,
                                                  double *__restrict__ f,
                                                  const double *__restrict__ lam,
                                                  double2 forcing, size_t n_cplx) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_cplx; i += stride)
        esq::st2(f, i, cmul_add(esq::ld2(lam, i), esq::ld2(y, i), forcing));
}

// f = lam*y + forcing, pointwise: the same epilogues on a grid-stride loop
// (lam holds exactly n doubles; the state vectors are zero-padded to a multiple
// of 512, and the padding must stay zero).  CPLX: n counts doubles (2 per
// element), forcing is complex.
template <bool CPLX, class Epi, class Src>
__global__ __launch_bounds__(kBlock) void k_diag_sweep(
    Src y, double *__restrict__ f, Epi epi,
    const double *__restrict__ lam, double2 forcing, size_t n, size_t n2) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    double local = 0.0;
    for (size_t i2 = (size_t)blockIdx.x * kBlock + threadIdx.x; i2 < n2; i2 += stride) {
        typename Epi::In in;
        epi.load(in, i2);
        const double2 yc = y.ld2(i2);
        double2 fy = make_double2(0.0, 0.0);
        if (CPLX) {
            if (2 * i2 < n) fy = cmul_add(esq::ld2(lam, i2), yc, forcing);
        } else {
            if (2 * i2 < n) fy.x = lam[2 * i2] * yc.x + forcing.x;
            if (2 * i2 + 1 < n) fy.y = lam[2 * i2 + 1] * yc.y + forcing.x;
        }
        epi.store_f(f, i2, fy);
        epi.finish(in, fy, yc, i2, local);
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}

template <bool CPLX>
int diag_fused(Rhs *r, double2 forcing, const double *y_in, double *f,
               const esq_epilogue *epi, size_t n, void *stream, void *start_event,
               void *stop_event) {
    // the state vectors are padded to a multiple of 512 doubles
    const size_t n_pad = ((n + 511) / 512) * 512, n2 = n_pad / 2;
    size_t blocks = (n2 + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    if (esq::epilogue_reduces(epi)) {
        if ((int)blocks > epi->partials_cap) return ESQ_ENOTSUP;
        if (epi->partials_used && !epi->dry_run) *epi->partials_used = (int)blocks;
    }
    if (epi->in_row && !first_stage_ok(epi)) return ESQ_ENOTSUP;
    if ((epi->is_complex != 0) != CPLX) return ESQ_EINVAL;
    const int rc = esq::dispatch_epilogue<CPLX>(epi, [&](auto ep) {
        using E = decltype(ep);
        if constexpr (kFirstStage<E>) {
            if (epi->in_row) {
                hipExtLaunchKernelGGL((k_diag_sweep<CPLX, E, SrcAxpy>),
                                      dim3((unsigned)blocks), dim3(kBlock), 0,
                                      (hipStream_t)stream, (hipEvent_t)start_event,
                                      (hipEvent_t)stop_event, 0, axpy_of(epi), f, ep,
                                      r->lam_dev, forcing, n, n2);
                return;
            }
        }
        hipExtLaunchKernelGGL((k_diag_sweep<CPLX, E, SrcPlain>), dim3((unsigned)blocks),
                              dim3(kBlock), 0, (hipStream_t)stream,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0,
                              SrcPlain{y_in}, f, ep, r->lam_dev, forcing, n, n2);
    });
    return (rc || epi->dry_run) ? rc : (int)hipGetLastError();
}

int diag_create(void **user_out, int kind, int device, const double *lam_host,
                size_t n_doubles, double amp_re, double amp_im) {
    if (!lam_host || n_doubles == 0) return ESQ_EINVAL;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    double *d = nullptr;
    e = hipMalloc(&d, n_doubles * sizeof(double));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(d, lam_host, n_doubles * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return (int)e; }
    Rhs r{};
    r.kind = kind; r.n = n_doubles; r.lam_dev = d; r.amp = amp_re; r.a = amp_im;
    r.device = device;
    return make(user_out, r);
}
double2 forcing_of(const Rhs *r, double t) {
    if (r->amp == 0.0 && r->a == 0.0) return make_double2(0.0, 0.0);
    const double s = sin(t);
    return make_double2(r->amp * s, r->a * s);
}

}  // namespace

extern "C" {

int esq_rhs_diag_create(void **user_out, int device, const double *lam_host,
                        size_t n, double forcing_amp) {
    return diag_create(user_out, DIAG, device, lam_host, n, forcing_amp, 0.0);
}
int esq_rhs_cdiag_create(void **user_out, int device, const double *lam_host,
                         size_t n_complex, double amp_re, double amp_im) {
    return diag_create(user_out, CDIAG, device, lam_host, 2 * n_complex, amp_re,
                       amp_im);
}
int esq_rhs_set_options(void *user, const char *options) {
    if (!user) return ESQ_EINVAL;
    esq::Options o;
    std::string bad;
    if (o.parse(options, esq::kOptPlugin, &bad) != 0) return ESQ_EINVAL;
    apply_options((Rhs *)user, o);
    return 0;
}
int esq_rhs_free(void *user) {
    if (!user) return 0;
    Rhs *r = (Rhs *)user;
    if ((r->kind == DIAG || r->kind == CDIAG) && r->lam_dev) {
        (void)hipSetDevice(r->device);
        (void)hipFree(r->lam_dev);
    }
    free(r);
    return 0;
}

int esq_rhs_diag(void *user, double t, const double *y, double *f, size_t n,
                 void *stream) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIAG || n != r->n) return ESQ_EINVAL;
    const double forcing = r->amp != 0.0 ? r->amp * sin(t) : 0.0;
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_diag, dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, y, f, r->lam_dev, forcing, n);
    return (int)hipGetLastError();
}
int esq_rhs_diag_fused(void *user, double t, const double *y_in, double *f,
                       const esq_epilogue *epi, size_t n, void *stream,
                       void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != DIAG || n != r->n || !epi) return ESQ_EINVAL;
    return diag_fused<false>(r, forcing_of(r, t), y_in, f, epi, n, stream,
                             start_event, stop_event);
}
/* complex state: n counts doubles (two per element) */
int esq_rhs_cdiag(void *user, double t, const double *y, double *f, size_t n,
                  void *stream) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != CDIAG || n != r->n) return ESQ_EINVAL;
    size_t blocks = (n / 2 + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_cdiag, dim3((unsigned)blocks), dim3(kBlock), 0,
                       (hipStream_t)stream, y, f, r->lam_dev, forcing_of(r, t), n / 2);
    return (int)hipGetLastError();
}
int esq_rhs_cdiag_fused(void *user, double t, const double *y_in, double *f,
                        const esq_epilogue *epi, size_t n, void *stream,
                        void *start_event, void *stop_event) {
    Rhs *r = (Rhs *)user;
    if (!r || r->kind != CDIAG || n != r->n || !epi) return ESQ_EINVAL;
    return diag_fused<true>(r, forcing_of(r, t), y_in, f, epi, n, stream,
                            start_event, stop_event);
}

}  // extern "C"
