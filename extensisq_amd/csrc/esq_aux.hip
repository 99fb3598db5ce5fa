// esq_aux.hip -- off the per-step hot loop: device-resident dense output
// (common.py:358-368, 766-790), Runge-Kutta-Chebyshev launches
// (sommeijer.py:273-329), vector plumbing of the power iteration and of the
// starting-step estimate (common.py:519-763).
#include "esq_internal.hpp"
#include "esq_aux_kernels.hpp"

using namespace esqi;

struct esq_dense {
    int device = 0;
    size_t len = 0, len_pad = 0;
    int np = 0;
    unsigned grid = 0;
    double *mem = nullptr;      // (np + 2) vectors: Qh columns, base, scratch
    size_t mem_bytes = 0;
    hipStream_t stream = nullptr;
};
template <int NT>
void launch_dense_n(esq_ctx *c, const DenseArgs &a, int np, double scale) {
    hipLaunchKernelGGL(k_dense_q<NT>, dim3(c->grid_stream), dim3(kBlock), 0,
                       c->stream, a, np, scale, c->len_pad / 2);
}
extern "C" {
// Q_k = scale * sum_j W[j][k] * src[j]  (k < p, ascending j: one fused pass over the
// sources for all columns), base = a copy of `base`: the interpolant owns its memory
static int dense_build(esq_ctx *c, const double *const *src, int nsrc, const double *W, int p,
                       double scale, const double *base, esq_dense **out) {
    if (p < 1 || p > kMaxCols) return fail(c, ESQ_EINVAL, "bad interpolant width %d", p);
    esq_dense *d = new (std::nothrow) esq_dense();
    if (!d) return ESQ_ENOMEM;
    d->device = c->device;
    d->len = c->len;
    d->len_pad = c->len_pad;
    d->np = p;
    d->grid = c->grid_stream;
    d->mem_bytes = (size_t)(p + 2) * d->len_pad * sizeof(double);
    hipError_t e = dev_acquire(c->device, (void **)&d->mem, d->mem_bytes);
    if (e != hipSuccess) {
        delete d;
        return fail(c, (int)e, "hipMalloc for the interpolant failed: %s",
                    hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { dev_release(d->device, d->mem, d->mem_bytes); delete d; return fail(c, (int)e, "stream"); }
    DenseArgs a;
    int nt = 0;
    for (int j = 0; j < nsrc; ++j) {
        bool any = false;
        for (int k = 0; k < p; ++k) any = any || W[(size_t)j * p + k] != 0.0;
        if (!any) continue;
        if (nt >= kMaxTerms) { esq_dense_destroy(d); return fail(c, ESQ_EINVAL, "too many rows"); }
        a.p[nt] = src[j];
        for (int k = 0; k < kMaxCols; ++k) a.w[nt][k] = k < p ? W[(size_t)j * p + k] : 0.0;
        ++nt;
    }
    for (int j = nt; j < kMaxTerms; ++j) {
        a.p[j] = nullptr;
        for (int k = 0; k < kMaxCols; ++k) a.w[j][k] = 0.0;
    }
    for (int k = 0; k < kMaxCols; ++k) a.q[k] = k < p ? d->mem + (size_t)k * d->len_pad : nullptr;
    if (nt < 1) { esq_dense_destroy(d); return fail(c, ESQ_EINVAL, "the weights are all zero"); }
    switch (nt) {
#define CASE(N) case N: launch_dense_n<N>(c, a, p, scale); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17)
        CASE(18) CASE(19) CASE(20)
#undef CASE
    }
    e = hipMemcpyAsync(d->mem + (size_t)p * d->len_pad, base,
                       d->len_pad * sizeof(double), hipMemcpyDefault, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        esq_dense_destroy(d);
        return fail(c, (int)e, "interpolant build failed: %s", hipGetErrorString(e));
    }
    *out = d;
    return 0;
}
static double *dense_vec(esq_ctx *c, int r);      // (vec_ptr, below)

int esq_dense_create(esq_ctx *c, const double *P, int rows, int p, double h,
                     int from_end, esq_dense **out) {
    if (!c || !P || !out) return ESQ_EINVAL;
    ENTER_KEEP(c);
    ENSURE_ROWS(c);
    if (rows < 1 || rows > c->n_rows || p < 1 || p > kMaxCols)
        return fail(c, ESQ_EINVAL, "bad interpolant shape (%d, %d)", rows, p);
    std::vector<const double *> src(rows);
    for (int j = 0; j < rows; ++j) src[j] = c->krow[c->kmap_last[j]];
    // base state: after esq_rk_accept, Y is the new state, YNEW the pre-step one
    return dense_build(c, src.data(), rows, P, p, h, from_end ? c->y : c->ynew, out);
}
// The same interpolant object from ANY vectors of the context (ids as in esq_vec_*: a
// physical K row >= 0, ESQ_VEC_Y ...):  Q_k = sum_j W[j][k] * vec_j,  base = vec(base_vec).
// What the C1 cubic Hermite interpolant of the reference (common.py:793-821;
// SSV2stab: sommeijer.py:400-406) becomes in Horner form -- with d = y - y_old,
//     y(x) = y_old + x*(h f_old) + x^2*(3 d - 2 h f_old - h f) + x^3*(-2 d + h f_old + h f)
// -- so that `dense_output()` of SSV2stab and of tableaux without P copies nothing to
// the host until it is evaluated.
int esq_dense_create_vecs(esq_ctx *c, const int *vec_ids, int nvec, const double *W, int p,
                          int base_vec, esq_dense **out) {
    if (!c || !vec_ids || !W || !out) return ESQ_EINVAL;
    ENTER_KEEP(c);
    ENSURE_ROWS(c);
    if (nvec < 1 || nvec > kMaxTerms) return fail(c, ESQ_EINVAL, "bad vector count %d", nvec);
    std::vector<const double *> src(nvec);
    for (int j = 0; j < nvec; ++j)
        if (!(src[j] = dense_vec(c, vec_ids[j]))) return fail(c, ESQ_EINVAL, "bad vector id");
    const double *base = dense_vec(c, base_vec);
    if (!base) return fail(c, ESQ_EINVAL, "bad base vector id");
    return dense_build(c, src.data(), nvec, W, p, 1.0, base, out);
}
int esq_dense_eval(esq_dense *d, double x, double *host_out) {
    if (!d || !host_out) return ESQ_EINVAL;
    hipError_t e = hipSetDevice(d->device);
    if (e != hipSuccess) return (int)e;
    HornerArgs a;
    for (int k = 0; k < kMaxCols; ++k)
        a.q[k] = k < d->np ? d->mem + (size_t)k * d->len_pad : nullptr;
    double *base = d->mem + (size_t)d->np * d->len_pad;
    double *scratch = d->mem + (size_t)(d->np + 1) * d->len_pad;
    hipLaunchKernelGGL(k_horner, dim3(d->grid), dim3(kBlock), 0, d->stream,
                       scratch, base, a, d->np, x, d->len_pad / 2);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(host_out, scratch, d->len * sizeof(double),
                       hipMemcpyDeviceToHost, d->stream);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(d->stream);
}
int esq_dense_download(esq_dense *d, double *Qh_host) {
    if (!d || !Qh_host) return ESQ_EINVAL;
    hipError_t e = hipSetDevice(d->device);
    if (e != hipSuccess) return (int)e;
    for (int k = 0; k < d->np; ++k) {
        e = hipMemcpyAsync(Qh_host + (size_t)k * d->len, d->mem + (size_t)k * d->len_pad,
                           d->len * sizeof(double), hipMemcpyDeviceToHost, d->stream);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipStreamSynchronize(d->stream);
}
int esq_dense_destroy(esq_dense *d) {
    if (!d) return 0;
    (void)hipSetDevice(d->device);
    if (d->stream) { (void)hipStreamSynchronize(d->stream); (void)hipStreamDestroy(d->stream); }
    if (d->mem) dev_release(d->device, d->mem, d->mem_bytes);
    delete d;
    return 0;
}

// ---- RKC ----------------------------------------------------------------------
// vector ids of the esq_rkc_* / esq_vec_* family: r >= 0 is a PHYSICAL K row,
// ESQ_VEC_Y ... ESQ_VEC_WORK name the fixed slots
static double *vec_ptr(esq_ctx *c, int r) {
    if (r >= 0) return r < c->n_rows ? c->krow[r] : nullptr;
    switch (r) {
        case ESQ_VEC_Y: return c->y;
        case ESQ_VEC_YNEW: return c->ynew;
        case ESQ_VEC_YSTAGE: return c->ystage;
        case ESQ_VEC_WORK: return c->work;
        default: return nullptr;
    }
}
#define ROW(c, r) vec_ptr((c), (r))
static double *dense_vec(esq_ctx *c, int r) { return vec_ptr(c, r); }
// a vector id whose overwriting changes what the rows evaluated on demand
// (esq_rk_lazy_rows) would be computed from: they are evaluated first
#define ENSURE_ROWS_BEFORE_WRITE(c, id)                                   \
    do {                                                                  \
        if ((id) >= 0 || (id) == ESQ_VEC_Y || (id) == ESQ_VEC_YNEW) ENSURE_ROWS(c); \
    } while (0)

int esq_rkc_first_stage(esq_ctx *c, int dst, int yn, int fn, double hmus) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *d = ROW(c, dst), *a = ROW(c, yn), *f = ROW(c, fn);
    if (!d || !a || !f) return fail(c, ESQ_EINVAL, "bad row");
    Prof p(c, ESQ_PROF_RKC, "k_rkc_first", -1, 24.0 * (double)c->len);
    hipExtLaunchKernelGGL(k_rkc_first, dim3(c->grid_stream), dim3(kBlock), 0,
                          c->stream, p.start(), p.stop(), 0, d, a, f, hmus,
                          c->len_pad / 2);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int esq_rkc_stage(esq_ctx *c, int dst, int fy, int yjm1, int yjm2, int yn, int fn,
                  double mu, double nu, double hmus, double ajm1) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *d = ROW(c, dst), *f = ROW(c, fy), *a = ROW(c, yjm1), *b = ROW(c, yjm2),
           *y0 = ROW(c, yn), *g = ROW(c, fn);
    if (!d || !f || !a || !b || !y0 || !g) return fail(c, ESQ_EINVAL, "bad row");
    const double omn = (1.0 - mu) - nu;   // (1.0 - mu - nu), left to right
    Prof p(c, ESQ_PROF_RKC, "k_rkc_stage", -1, 48.0 * (double)c->len);
    hipExtLaunchKernelGGL(k_rkc_stage, dim3(c->grid_stream), dim3(kBlock), 0,
                          c->stream, p.start(), p.stop(), 0, d, f, a, b, y0, g,
                          mu, nu, omn, hmus, ajm1, c->len_pad / 2);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int esq_rkc_eval_rhs(esq_ctx *c, int dst, double t, int src) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *d = ROW(c, dst), *s = ROW(c, src);
    if (!d || !s) return fail(c, ESQ_EINVAL, "bad row");
    return call_rhs(c, t, s, d);
}
namespace {
// how many of the `left` remaining stages the next chain sweep takes (< 2: one stage by
// itself): never a single stage at the end -- 5 = 3 + 2 rather than 4 + 1 -- and no
// length the entry has declined
int rkc_chain_len(int left, int depth, unsigned refused) {
    int d = left < depth ? left : depth;
    if (left - d == 1 && d >= 3) --d;
    while (d >= 2 && ((refused >> d) & 1u)) --d;
    return d;
}
// the end of the step, wanted with the stages (esq_rkc_stages_end): f(t_end, y_{n+1})
// into a work row and the error estimate's sum of squares
struct RkcTail {
    double t_end, h;
    int fy_row = ESQ_VEC_NONE;
    double sumsq = 0.0;
    bool done = false;         // the last chain sweep took it along (LAST form)
};
int rkc_stages(esq_ctx *c, int yn, int fn, int w0, int w1, int w2, int w3, double hmus1,
               int m, const double *scalars, int *y_row_out, RkcTail *tail);
}  // namespace
// The next step's opening chain sweep (FIRST form: y_1 formed on the fly from y_{n+1} and
// its derivative), into the two free work rows -- behind the final sum of the step being
// finished, before the host waits for it.  Exactly the launch the next esq_rkc_stages_end
// would make first (rkc_stages below), only earlier.
extern "C++" bool esqi::rkc_launch_ahead_if_asked(esq_ctx *c) {
    esq_ctx::RkcAhead &ah = c->rkc_ahead;
    if (!ah.ask || !ah.armed) return false;
    ah.ask = ah.armed = false;
    ah.valid = false;
    const int m = ah.ask_m;
    if (!c->rhs_rkc_chain || !c->rkc_first || c->rkc_first_refused || c->rkc_depth < 2 ||
        m < 3 || c->comm || c->cplx)
        return false;
    const int d = rkc_chain_len(m - 1, c->rkc_depth, c->rkc_refused);
    if (d < 2 || 2 + d > m) return false;              // (not the chain that ends the step)
    if (ah.free_a == ESQ_VEC_NONE || ah.free_b == ESQ_VEC_NONE) return false;
    esq_rkc_chain ch;
    memset(&ch, 0, sizeof(ch));
    ch.depth = d;
    ch.yjm1 = nullptr;
    ch.hmus_first = ah.ask_hmus1;
    ch.yjm2 = ROW(c, ah.at_y);
    ch.yn = ROW(c, ah.at_y);
    ch.fn = ROW(c, ah.at_f);
    for (int k = 0; k < d; ++k) {
        const double *s5 = ah.ask_sc + 5 * (size_t)k;
        ch.mu[k] = s5[0]; ch.nu[k] = s5[1];
        ch.omn[k] = (1.0 - s5[0]) - s5[1];
        ch.hmus[k] = s5[2]; ch.ajm1[k] = s5[3]; ch.t[k] = s5[4];
    }
    ch.out = ROW(c, ah.free_a);
    ch.out_prev = ROW(c, ah.free_b);
    if (!ch.yn || !ch.fn || !ch.out || !ch.out_prev) return false;
    double amp = 1.0;
    ch.read_amplification = &amp;
    char label[24];
    snprintf(label, sizeof(label), "rkc_chain%d-first", d);
    Prof p(c, ESQ_PROF_RKC, label, -1, 64.0 * d * (double)c->len, false, 0.0);
    const int r = c->rhs_rkc_chain(c->rhs_user, &ch, c->len, (void *)c->stream,
                                   (void *)p.start(), (void *)p.stop());
    if (r != 0) {                          // declined: the next call finds it out again
        p.cancel();
        return false;
    }
    p.ev.moved = (2.0 * amp + 2.0) * 8.0 * (double)c->len;
    p.ev.floor = 4.0 * 8.0 * (double)c->len;
    ah.valid = true;
    ah.yn = ah.at_y; ah.fn = ah.at_f;
    ah.out = ah.free_a; ah.outp = ah.free_b;
    ah.d = d; ah.m = m; ah.hmus1 = ah.ask_hmus1;
    memcpy(ah.sc, ah.ask_sc, sizeof(double) * 5 * (size_t)d);
    return true;
}
int esq_rkc_guess_next(esq_ctx *c, double hmus1_next, int m_next, const double *scalars_next) {
    if (!c || m_next < 1 || (m_next > 1 && !scalars_next)) return ESQ_EINVAL;
    ENTER_KEEP(c);
    esq_ctx::RkcAhead &ah = c->rkc_ahead;
    ah.ask = true;
    ah.armed = false;
    ah.ask_m = m_next;
    ah.ask_hmus1 = hmus1_next;
    const int rows = m_next - 1 < ESQ_RKC_CHAIN_MAX_DEPTH ? m_next - 1 : ESQ_RKC_CHAIN_MAX_DEPTH;
    if (rows > 0) memcpy(ah.ask_sc, scalars_next, sizeof(double) * 5 * (size_t)rows);
    return 0;
}
int esq_rkc_stages(esq_ctx *c, int yn, int fn, int w0, int w1, int w2, int w3,
                   double hmus1, int m, const double *scalars, int *y_row_out) {
    if (!c || !y_row_out || m < 1 || (m > 1 && !scalars)) return ESQ_EINVAL;
    ENTER(c);
    return rkc_stages(c, yn, fn, w0, w1, w2, w3, hmus1, m, scalars, y_row_out, nullptr);
}
int esq_rkc_stages_end(esq_ctx *c, int yn, int fn, int w0, int w1, int w2, int w3,
                       double hmus1, int m, const double *scalars, double t_end, double h,
                       int *y_row_out, int *fy_row_out, double *sumsq_out) {
    if (!c || !y_row_out || !fy_row_out || !sumsq_out || m < 1 || (m > 1 && !scalars))
        return ESQ_EINVAL;
    // (the opening sweep of this step may be in the queue already: esq_rkc_guess_next)
    const bool ahead_valid = c->rkc_ahead.valid;
    ENTER(c);
    c->rkc_ahead.valid = ahead_valid;
    RkcTail tail;
    tail.t_end = t_end;
    tail.h = h;
    int r = rkc_stages(c, yn, fn, w0, w1, w2, w3, hmus1, m, scalars, y_row_out, &tail);
    c->rkc_ahead.ask = c->rkc_ahead.armed = false;
    if (r) return r;
    if (!tail.done) {
        // f(t_end, y) into a work row that does not hold y, and the estimate
        const int work[4] = {w0, w1, w2, w3};
        tail.fy_row = ESQ_VEC_NONE;
        for (int a = 0; a < 4 && tail.fy_row == ESQ_VEC_NONE; ++a)
            if (work[a] != ESQ_VEC_NONE && work[a] != *y_row_out) tail.fy_row = work[a];
        r = esq_rkc_end_error(c, *y_row_out, yn, fn, tail.fy_row, t_end, h, &tail.sumsq);
        if (r) return r;
    }
    *fy_row_out = tail.fy_row;
    *sumsq_out = tail.sumsq;
    return 0;
}
namespace {
int rkc_stages(esq_ctx *c, int yn, int fn, int w0, int w1, int w2, int w3, double hmus1,
               int m, const double *scalars, int *y_row_out, RkcTail *tail) {
    // rotation instead of the reference's two full copies per stage:
    //   jm1 = first-stage result, jm2 = yn; every stage writes into a free row
    const int work[4] = {w0, w1, w2, w3};
    const int nwork = w3 == ESQ_VEC_NONE ? 3 : 4;
    // the first iterate y_1 = yn + hmus1*fn: a sweep of its own, unless the chain
    // that opens the step forms it on the fly (FIRST; at least two more stages)
    bool first_pending = c->rhs_rkc_chain && c->rkc_first && !c->rkc_first_refused &&
                         nwork == 4 && c->rkc_depth >= 2 && m >= 3;
    int r = 0;
    if (!first_pending) {
        r = esq_rkc_first_stage(c, w0, yn, fn, hmus1);
        if (r) return r;
    }
    for (int a = 0; a < nwork; ++a) {
        if (!ROW(c, work[a]) || work[a] == yn || work[a] == fn)
            return fail(c, ESQ_EINVAL, "bad work row");
        for (int b = 0; b < a; ++b)
            if (work[a] == work[b]) return fail(c, ESQ_EINVAL, "work rows must differ");
    }
    // a work row that holds neither of the two live iterates
    auto free_row = [&](int jm1, int jm2, int not_this) -> int {
        for (int a = 0; a < nwork; ++a)
            if (work[a] != jm1 && work[a] != jm2 && work[a] != not_this) return work[a];
        return ESQ_VEC_NONE;
    };
    int jm1 = w0, jm2 = yn;
    int j = 2;
    // the step's opening chain sweep was launched behind the previous step's final sum
    // (esq_rkc_guess_next): taken up if this call asks for exactly that recursion on
    // exactly those rows, dropped otherwise (it wrote two work rows)
    if (c->rkc_ahead.valid) {
        esq_ctx::RkcAhead &ah = c->rkc_ahead;
        bool in_work_a = false, in_work_b = false;
        for (int a = 0; a < nwork; ++a) {
            in_work_a |= work[a] == ah.out;
            in_work_b |= work[a] == ah.outp;
        }
        const bool take = tail && first_pending && yn == ah.yn && fn == ah.fn && m == ah.m &&
                          hmus1 == ah.hmus1 && in_work_a && in_work_b && ah.d >= 2 &&
                          2 + ah.d <= m &&
                          memcmp(scalars, ah.sc, sizeof(double) * 5 * (size_t)ah.d) == 0;
        ah.valid = false;
        if (take) {
            ++c->ahead_used;
            first_pending = false;
            jm1 = ah.out;
            jm2 = ah.outp;
            j = 2 + ah.d;
        } else {
            ++c->ahead_dropped;
        }
    }
    while (j <= m) {
        const double *sc = scalars + 5 * (size_t)(j - 2);
        // ---- a chain of d stages: y_{j-1}, y_{j-2} in, y_{j+d-1}, y_{j+d-2} out
        if (c->rhs_rkc_chain && nwork == 4 && c->rkc_depth >= 2) {
            const int d = rkc_chain_len(m - j + 1, c->rkc_depth, c->rkc_refused);
            if (d >= 2) {
                const int o1 = free_row(jm1, jm2, ESQ_VEC_NONE);
                const int o2 = free_row(jm1, jm2, o1);
                esq_rkc_chain ch;
                memset(&ch, 0, sizeof(ch));
                ch.depth = d;
                ch.yjm1 = first_pending ? nullptr : ROW(c, jm1);
                ch.hmus_first = first_pending ? hmus1 : 0.0;
                ch.yjm2 = ROW(c, jm2);
                ch.yn = ROW(c, yn); ch.fn = ROW(c, fn);
                for (int k = 0; k < d; ++k) {
                    const double *s5 = sc + 5 * (size_t)k;
                    ch.mu[k] = s5[0]; ch.nu[k] = s5[1];
                    ch.omn[k] = (1.0 - s5[0]) - s5[1];    // (1.0 - mu - nu), left to right
                    ch.hmus[k] = s5[2]; ch.ajm1[k] = s5[3]; ch.t[k] = s5[4];
                }
                const bool last = j + d > m;              // nothing reads y_{m-1}
                ch.out = ROW(c, o1);
                ch.out_prev = last ? nullptr : ROW(c, o2);
                // the chain that ends the step takes f(t_end, y) and the error
                // estimate along (LAST; d + 1 stage slots)
                const bool with_end = last && tail && c->rkc_last &&
                                      !((c->rkc_last_refused >> d) & 1u) && !first_pending &&
                                      d + 1 <= ESQ_RKC_CHAIN_MAX_DEPTH && !c->cplx;
                if (with_end) {
                    ch.fy_out = ROW(c, o2);
                    ch.t_end = tail->t_end;
                    ch.h = tail->h;
                    ch.atol_vec = c->atol_is_vec ? c->atolv : nullptr;
                    ch.atol_s = c->atol_s;
                    ch.rtol = c->rtol;
                    ch.n_valid = c->n;
                    ch.partials = c->partials;
                    ch.partials_cap = kPartialsCap;
                    ch.partials_used = &c->red_count;
                }
                double amp = 1.0;
                ch.read_amplification = &amp;
                if (o1 == ESQ_VEC_NONE || o2 == ESQ_VEC_NONE || !ch.out)
                    return fail(c, ESQ_EINVAL, "bad row");
                char label[24];
                snprintf(label, sizeof(label), "rkc_chain%d%s", d,
                         first_pending ? "-first" : with_end ? "-end" : last ? "-last" : "");
                // booked: d x (RHS 16 B + recursion 48 B); moved: 4 inputs (x halo
                // factor, filled in by the plugin) + 2 (1) outputs
                Prof p(c, ESQ_PROF_RKC, label, -1, 64.0 * d * (double)c->len, false,
                       0.0);
                c->self_valid = false;
                r = c->rhs_rkc_chain(c->rhs_user, &ch, c->len, (void *)c->stream,
                                     (void *)p.start(), (void *)p.stop());
                if (r == 0) {
                    const double in = first_pending ? 2.0 : 4.0;
                    const double outw = (last && !with_end) ? 1.0 : 2.0;    // -end: y and f(y)
                    p.ev.moved = (in * amp + outw) * 8.0 * (double)c->len;
                    p.ev.floor = (in + outw) * 8.0 * (double)c->len;
                    first_pending = false;
                    if (with_end) {
                        tail->done = true;
                        tail->fy_row = o2;
                        *y_row_out = o1;
                        if (c->rkc_ahead.ask) {
                            // (two work rows that hold neither y_{n+1} nor its derivative)
                            esq_ctx::RkcAhead &ah = c->rkc_ahead;
                            ah.at_y = o1;
                            ah.at_f = o2;
                            ah.free_a = free_row(o1, o2, ESQ_VEC_NONE);
                            ah.free_b = free_row(o1, o2, ah.free_a);
                            ah.armed = true;
                        }
                        return finish_reduction(c, &tail->sumsq, false, c->partials,
                                                c->red_count);
                    }
                    jm2 = last ? jm1 : o2;
                    jm1 = o1;
                    j += d;
                    continue;
                }
                p.cancel();
                if (r != ESQ_ENOTSUP)
                    return fail(c, ESQ_ERHS, "RKC chain entry returned %d", r);
                if (with_end) {                            // declined in this form
                    c->rkc_last_refused |= 1u << d;
                    continue;
                }
                if (first_pending) {
                    // declined in this form: y_1 the plain way, then the same chain
                    c->rkc_first_refused = true;
                    first_pending = false;
                    r = esq_rkc_first_stage(c, w0, yn, fn, hmus1);
                    if (r) return r;
                    continue;
                }
                c->rkc_refused |= 1u << d;
                continue;                                  // a shorter chain, or one stage
            }
        }
        if (first_pending) {                               // no chain took the form
            first_pending = false;
            r = esq_rkc_first_stage(c, w0, yn, fn, hmus1);
            if (r) return r;
        }
        const int dst = free_row(jm1, jm2, ESQ_VEC_NONE);
        bool done = false;
        if (c->rhs_rkc) {
            // ONE sweep: derivative of yjm1 and the recursion, no fy in memory
            double *d = ROW(c, dst), *a = ROW(c, jm1), *b = ROW(c, jm2),
                   *y0 = ROW(c, yn), *g = ROW(c, fn);
            if (!d || !a || !b || !y0 || !g) return fail(c, ESQ_EINVAL, "bad row");
            const double omn = (1.0 - sc[0]) - sc[1];
            Prof p(c, ESQ_PROF_RKC, "rhs_rkc", -1, 64.0 * (double)c->len, false,
                   40.0 * (double)c->len);
            c->self_valid = false;
            r = c->rhs_rkc(c->rhs_user, sc[4], a, b, y0, g, sc[0], sc[1], omn,
                           sc[2], sc[3], d, c->len, (void *)c->stream,
                           (void *)p.start(), (void *)p.stop());
            if (r == 0) done = true;
            else if (r != ESQ_ENOTSUP)
                return fail(c, ESQ_ERHS, "RKC plugin entry returned %d", r);
            else p.cancel();
        }
        if (!done) {
            // fy = rhs(t_stage, yjm1) into dst, the combination overwrites it
            r = esq_rkc_eval_rhs(c, dst, sc[4], jm1);
            if (r) return r;
            r = esq_rkc_stage(c, dst, dst, jm1, jm2, yn, fn, sc[0], sc[1], sc[2], sc[3]);
            if (r) return r;
        }
        // shift: jm2 <- jm1, jm1 <- new; the old jm2 row becomes free (yn is
        // never recycled: it is not a work row)
        jm2 = jm1;
        jm1 = dst;
        ++j;
    }
    *y_row_out = jm1;
    return 0;
}
}  // namespace
// the same walk as rkc_stages, in words (tests/test_step_plans.py)
int esq_rkc_plan_describe(int m, int max_depth, int end_slots_max, char *buf, size_t buflen) {
    if (m < 1 || !buf || buflen < 64) return ESQ_EINVAL;
    const bool can_first = (max_depth & ESQ_RKC_CHAIN_FIRST) != 0;
    const bool can_last = (max_depth & ESQ_RKC_CHAIN_LAST) != 0;
    int depth = max_depth & 0xff;
    if (depth > ESQ_RKC_CHAIN_MAX_DEPTH) depth = ESQ_RKC_CHAIN_MAX_DEPTH;
    std::string out;
    int launches = 0;
    auto put = [&](const std::string &label) {
        out += out.empty() ? label : " " + label;
        ++launches;
    };
    bool first_pending = depth >= 2 && can_first && m >= 3;
    if (!first_pending) put("k_rkc_first");
    bool end_done = false;
    int j = 2;
    while (j <= m) {
        const int d = depth >= 2 ? rkc_chain_len(m - j + 1, depth, 0u) : 1;
        if (d >= 2) {
            const bool last = j + d > m;
            const bool with_end = last && can_last && !first_pending &&
                                  d + 1 <= end_slots_max && d + 1 <= ESQ_RKC_CHAIN_MAX_DEPTH;
            put("rkc_chain" + std::to_string(d) +
                (first_pending ? "-first" : with_end ? "-end" : last ? "-last" : ""));
            first_pending = false;
            end_done = with_end;
            j += d;
            continue;
        }
        if (first_pending) { first_pending = false; put("k_rkc_first"); }
        put("rhs_rkc");
        ++j;
    }
    if (!end_done) put("rhs+rkcerr");
    out += " | launches=" + std::to_string(launches);
    if (out.size() + 1 > buflen) return ESQ_EINVAL;
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}
int esq_rkc_error_norm(esq_ctx *c, int y, int yn, int fn, int fy, double h,
                       double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    double *a = ROW(c, y), *b = ROW(c, yn), *f = ROW(c, fn), *g = ROW(c, fy);
    if (!a || !b || !f || !g) return fail(c, ESQ_EINVAL, "bad row");
    if (c->cplx) return fail(c, ESQ_EINVAL, "RKC is real-only (sommeijer.py:98)");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_rkc_error", -1, 32.0 * (double)c->len);
        hipExtLaunchKernelGGL(k_rkc_error, dim3(c->grid_reduce), dim3(kBlock), 0,
                           c->stream, p.start(), p.stop(), 0, a, b, f, g, h,
                           c->atol_is_vec ? c->atolv : nullptr, c->atol_s, c->rtol,
                           c->len_pad / 2, c->n, c->partials);
        HIPCHK(c, hipGetLastError());
    }
    return finish_reduction(c, sumsq_out);
}
int esq_rkc_end_error(esq_ctx *c, int y, int yn, int fn, int fy, double t_end,
                      double h, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    double *a = ROW(c, y), *b = ROW(c, yn), *f = ROW(c, fn), *g = ROW(c, fy);
    if (!a || !b || !f || !g) return fail(c, ESQ_EINVAL, "bad row");
    if (c->cplx) return fail(c, ESQ_EINVAL, "RKC is real-only (sommeijer.py:98)");
    if (c->rhs_fused && ((c->fuse_mask >> ESQ_EPI_RKCERR) & 1)) {
        // fy = f(t_end, y) and the error partial sums in ONE sweep
        esq_epilogue e;
        memset(&e, 0, sizeof(e));
        e.kind = ESQ_EPI_RKCERR;
        e.rows[0] = b;
        e.rows[1] = f;
        e.h = h;
        e.atol_vec = c->atol_is_vec ? c->atolv : nullptr;
        e.atol_s = c->atol_s;
        e.rtol = c->rtol;
        e.n_valid = c->n;
        e.partials = c->partials;
        e.partials_cap = kPartialsCap;
        e.partials_used = &c->red_count;
        // booked: RHS 16 B + the error pass (y, yn, fn, fy); moved: y, yn, fn in,
        // fy out
        Prof p(c, ESQ_PROF_SOLERR, "rhs+rkcerr", -1, 48.0 * (double)c->len, false,
               32.0 * (double)c->len);
        c->self_valid = false;
        const int r = c->rhs_fused(c->rhs_user, t_end, a, g, &e, c->len,
                                   (void *)c->stream, (void *)p.start(),
                                   (void *)p.stop());
        if (r == 0) return finish_reduction(c, sumsq_out, false, c->partials,
                                            c->red_count);
        p.cancel();
        if (r != ESQ_ENOTSUP) return fail(c, ESQ_ERHS, "fused RHS entry returned %d", r);
    }
    int r = call_rhs(c, t_end, a, g);
    if (r) return r;
    return esq_rkc_error_norm(c, y, yn, fn, fy, h, sumsq_out);
}
int esq_vec_sumsq(esq_ctx *c, int x, int y, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    double *a = ROW(c, x), *b = y != ESQ_VEC_NONE ? ROW(c, y) : nullptr;
    if (!a || (y != ESQ_VEC_NONE && !b)) return fail(c, ESQ_EINVAL, "bad row");
    hipLaunchKernelGGL(k_sumsq, dim3(c->grid_reduce), dim3(kBlock), 0, c->stream,
                       a, b, c->len_pad / 2, c->partials);
    HIPCHK(c, hipGetLastError());
    return finish_reduction(c, sumsq_out);
}
int esq_vec_axpbmc(esq_ctx *c, int dst, int a, double alpha, int b, int cc) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *d = ROW(c, dst), *pa = a != ESQ_VEC_NONE ? ROW(c, a) : nullptr,
           *pb = ROW(c, b), *pc = cc != ESQ_VEC_NONE ? ROW(c, cc) : nullptr;
    if (!d || !pb || (a != ESQ_VEC_NONE && !pa) || (cc != ESQ_VEC_NONE && !pc))
        return fail(c, ESQ_EINVAL, "bad row");
    hipLaunchKernelGGL(k_axpbmc, dim3(c->grid_stream), dim3(kBlock), 0, c->stream,
                       d, pa, alpha, pb, pc, c->len_pad / 2);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int esq_vec_wdiff_sumsq(esq_ctx *c, int a, int b, int w, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    double *pa = ROW(c, a), *pb = ROW(c, b), *pw = ROW(c, w);
    if (!pa || !pb || !pw) return fail(c, ESQ_EINVAL, "bad row");
    hipLaunchKernelGGL(k_wdiff_sumsq, dim3(c->grid_reduce), dim3(kBlock), 0,
                       c->stream, pa, pb, pw, c->atol_is_vec ? c->atolv : nullptr,
                       c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    HIPCHK(c, hipGetLastError());
    return finish_reduction(c, sumsq_out);
}

int esq_vec_wdot(esq_ctx *c, int a, int b, int y1, int y2, double floor_,
                 double *out) {
    if (!c || !out) return ESQ_EINVAL;
    ENTER(c);
    double *pa = ROW(c, a), *pb = ROW(c, b), *p1 = ROW(c, y1), *p2 = ROW(c, y2);
    if (!pa || !pb || !p1 || !p2) return fail(c, ESQ_EINVAL, "bad vector id");
    if (c->cplx)
        hipLaunchKernelGGL(k_wdot<true>, dim3(c->grid_reduce), dim3(kBlock), 0,
                           c->stream, pa, pb, p1, p2, floor_, c->len_pad / 2, c->n,
                           c->partials);
    else
        hipLaunchKernelGGL(k_wdot<false>, dim3(c->grid_reduce), dim3(kBlock), 0,
                           c->stream, pa, pb, p1, p2, floor_, c->len_pad / 2, c->n,
                           c->partials);
    HIPCHK(c, hipGetLastError());
    return finish_reduction(c, out);
}
int esq_vec_fill(esq_ctx *c, int dst, double value, double value_im) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    ENSURE_ROWS_BEFORE_WRITE(c, dst);
    double *d = ROW(c, dst);
    if (!d) return fail(c, ESQ_EINVAL, "bad vector id %d", dst);
    if (c->cplx)
        hipLaunchKernelGGL(k_fill<true>, dim3(c->grid_stream), dim3(kBlock), 0,
                           c->stream, d, value, value_im, c->len_pad / 2, c->n);
    else
        hipLaunchKernelGGL(k_fill<false>, dim3(c->grid_stream), dim3(kBlock), 0,
                           c->stream, d, value, value_im, c->len_pad / 2, c->n);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int esq_vec_copy(esq_ctx *c, int dst, int src) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    ENSURE_ROWS_BEFORE_WRITE(c, dst);
    if (src >= 0) ENSURE_ROWS(c);
    double *d = ROW(c, dst), *s = ROW(c, src);
    if (!d || !s) return fail(c, ESQ_EINVAL, "bad vector id");
    HIPCHK(c, hipMemcpyAsync(d, s, c->len_pad * sizeof(double),
                             hipMemcpyDefault, c->stream));
    return 0;
}
int esq_vec_eval_rhs(esq_ctx *c, int dst, double t, int src) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *d = ROW(c, dst), *s = ROW(c, src);
    if (!d || !s) return fail(c, ESQ_EINVAL, "bad vector id");
    return call_rhs(c, t, s, d);
}
int esq_vec_upload(esq_ctx *c, int dst, const double *host) {
    if (!c || !host) return ESQ_EINVAL;
    ENSURE_ROWS_BEFORE_WRITE(c, dst);
    const bool was_idle = c->idle;
    ENTER(c);
    double *d = ROW(c, dst);
    if (!d) return fail(c, ESQ_EINVAL, "bad vector id %d", dst);
    return h2d(c, d, host, c->len * sizeof(double), was_idle);
}
int esq_vec_download(esq_ctx *c, int src, double *host) {
    if (!c || !host) return ESQ_EINVAL;
    const bool was_idle = c->idle;
    ENTER_KEEP(c);
    double *s = ROW(c, src);
    if (!s) return fail(c, ESQ_EINVAL, "bad vector id %d", src);
    return d2h(c, host, s, c->len * sizeof(double), was_idle);
}
int esq_hs_log_etol(esq_ctx *c, int y, double *sum_out, double *min_out) {
    if (!c || !sum_out || !min_out) return ESQ_EINVAL;
    ENTER(c);
    double *py = ROW(c, y);
    if (!py) return fail(c, ESQ_EINVAL, "bad vector id %d", y);
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipLaunchKernelGGL(k_log_etol<true>, dim3(c->grid_reduce), dim3(kBlock), 0,
                           c->stream, py, av, c->atol_s, c->rtol, c->len_pad / 2,
                           c->n, c->partials, c->partials2);
    else
        hipLaunchKernelGGL(k_log_etol<false>, dim3(c->grid_reduce), dim3(kBlock), 0,
                           c->stream, py, av, c->atol_s, c->rtol, c->len_pad / 2,
                           c->n, c->partials, c->partials2);
    HIPCHK(c, hipGetLastError());
    int r = finish_reduction(c, sum_out);
    if (r) return r;
    return finish_reduction(c, min_out, /*take_min=*/true, c->partials2);
}
int esq_hs_select(esq_ctx *c, int yp, int spy, int src, double fill) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *a = ROW(c, yp), *b = ROW(c, spy), *s = ROW(c, src);
    if (!a || !b || !s) return fail(c, ESQ_EINVAL, "bad vector id");
    if (c->cplx)
        hipLaunchKernelGGL(k_hs_select<true>, dim3(c->grid_stream), dim3(kBlock), 0,
                           c->stream, a, b, s, fill, c->len_pad / 2, c->n);
    else
        hipLaunchKernelGGL(k_hs_select<false>, dim3(c->grid_stream), dim3(kBlock), 0,
                           c->stream, a, b, s, fill, c->len_pad / 2, c->n);
    HIPCHK(c, hipGetLastError());
    return 0;
}

}  // extern "C"
