// esq_abi.hip -- the C ABI of libextensisq_amd.so (see include/extensisq_amd.h).
//
// Host side of the device path: a context owns one HIP stream and ONE HBM slab
// holding every vector of the step (K rows, y, y_new, y_stage, atol, work), all
// 4-KiB aligned and padded to a multiple of 512 doubles so that kernels run
// without tail code.  Rotation of K rows and the y <-> y_new exchange are
// pointer swaps on the host; kernels receive row pointers and coefficients by
// value in their kernel arguments.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "../../include/extensisq_amd.h"
#include "esq_kernels.hpp"

using namespace esq;

namespace {

constexpr size_t kPadDoubles = 512;   // 4 KiB
constexpr int kFixedSlots = 5;        // Y, YNEW, YSTAGE, ATOL, WORK
constexpr int kPartialsCap = 1 << 17; // one partial per workgroup of a sweep

struct ProfEvent {
    hipEvent_t start, stop;
    int klass;
    double bytes;      // algorithmic bytes (SURVEY.md §8d definition)
    double moved;      // bytes the launch is designed to move
    char name[40];     // kernel label for the per-kernel table (esq_profile_kernels)
};
struct ProfKernel {    // per-label totals since the last reset
    int klass = 0;
    long launches = 0;
    double ms = 0.0, bytes = 0.0, moved = 0.0;
};
// pinned host slot a reduction's result lands in: the value, then the sequence
// number of the reduction (system-scope release), polled by the host
struct HostSlot {
    double value;
    unsigned long long seq;
};

}  // namespace

struct esq_ctx {
    int device = 0;
    size_t n = 0;          // state dimension as the user counts it
    size_t len = 0;        // doubles per vector (n or 2n)
    size_t len_pad = 0;    // padded doubles per vector
    size_t stride = 0;     // doubles between consecutive vectors in the slab
    int n_rows = 0;
    bool cplx = false;
    hipStream_t stream = nullptr;
    double *slab = nullptr;
    // small host-RHS problems: the slab is pinned, device-mapped HOST memory --
    // kernels read and write it over PCIe, uploads and downloads are plain
    // memcpy calls (no copy engine, no stream synchronisation)
    bool host_slab = false;
    double *slab_host = nullptr;      // host address of slab[0]
    size_t slab_doubles = 0;
    bool idle = true;                 // nothing enqueued since the last wait
    bool self_valid = false;          // the last kernel enqueued publishes self_seq
    unsigned long long self_seq = 0;
    std::vector<double *> aux_slabs;   // lazily added work rows (esq_aux_rows)
    std::vector<double *> krow;       // physical K rows
    std::vector<int> kmap;            // logical -> physical (step in flight)
    std::vector<int> kmap_last;       // mapping of the step just accepted
    double *y = nullptr, *ynew = nullptr, *ystage = nullptr, *atolv = nullptr,
           *work = nullptr;
    double *partials = nullptr;       // kPartialsCap doubles (own kernels use
                                      // <= kMaxPartials, fused sweeps their grid)
    double *partials2 = nullptr;      // second set (min reductions)
    double *d_result = nullptr;       // 8 doubles (device)
    HostSlot *h_slot = nullptr;       // pinned, device-visible host memory
    unsigned long long red_seq = 0;   // reductions issued so far
    double comm_timeout_s = 120.0;    // bounded wait of a lock-step all-reduce
    // method
    int s = 0, fsal = 0;
    std::vector<double> A, B, C, E;
    bool have_tab = false;
    double rtol = 1e-3, atol_s = 1e-6;
    bool atol_is_vec = false;
    esq_rhs_fn rhs = nullptr;
    void *rhs_user = nullptr;
    esq_rhs_fused_fn rhs_fused = nullptr;   // optional RHS + epilogue entry
    int fuse_mask = 0;                      // epilogue kinds the library may request
    bool src_declined = false;              // the plugin returned ENOTSUP for ESQ_FUSE_SRC
    bool src_pays = false;                  // working set inside the Infinity Cache
    esq_rhs_rkc_fn rhs_rkc = nullptr;       // optional RHS + Chebyshev recursion entry
    bool ynew_ready = false;     // YNEW already formed by the last stage's sweep
    bool solerr_ready = false;   // ... and the error partial sums too
    int red_count = 0;           // partials written by the last reducing sweep
    // first stage argument of the NEXT step, formed at accept time
    bool pre_valid = false;
    double pre_h = 0.0;
    // which sweeps stream the fresh derivative out with non-temporal stores
    // (ESQ_EPI_NT bits: 0 stage, 1 block, 2 solerr, 3 end-point, 4 FSAL errnorm)
    unsigned epi_nt = 0x3;
    // blocked accumulation plan (esq_rk_set_tableau)
    struct Block {
        int J = 0, prev = 0;              // columns [prev, J) of A
        std::vector<int> cols;            // non-zero columns of the block
        std::vector<int> stages;          // later stages that use them
        std::vector<int> out_vec;         // physical row of each stage's partial
        std::vector<int> in_vec;          // previous-level partial (-1: none)
    };
    std::vector<Block> blocks;
    int block_rows_first = -1, block_rows_count = 0;   // aux rows of the plan
    std::vector<int> stage_init;          // per stage: row of its partial or -1
    std::vector<int> stage_from;          // per stage: first column still to add
    // launch geometry
    unsigned grid_stream = 0;         // grid for streaming kernels
    unsigned grid_reduce = 0;
    unsigned grid_block = 0;          // grid of the (write-heavy) block kernel
    int stage_policy = 0;             // cache policy of k_lincomb (tuning knob)
    // lock-step
    void *comm = nullptr;
    // profiling
    unsigned prof_mask = 0;           // bit k: time launches of class k
    unsigned prof_every = 1;          // time every prof_every-th launch of a class
    unsigned long prof_seen[ESQ_PROF_NCLASS] = {0};
    std::vector<ProfEvent> prof_live;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[ESQ_PROF_NCLASS] = {0};
    long prof_cnt[ESQ_PROF_NCLASS] = {0};
    double prof_bytes[ESQ_PROF_NCLASS] = {0};
    double prof_moved[ESQ_PROF_NCLASS] = {0};
    std::map<std::string, ProfKernel> prof_kernels;
    char err[512] = {0};
};

namespace {

int fail(esq_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}
#define HIPCHK(c, call)                                                         \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess)                                                   \
            return fail((c), (int)e_, "%s failed: %s (%s:%d)", #call,           \
                        hipGetErrorString(e_), __FILE__, __LINE__);             \
    } while (0)

double *slot_ptr(esq_ctx *c, int slot, int row, bool logical = true) {
    switch (slot) {
        case ESQ_SLOT_K:
            if (row < 0 || row >= c->n_rows) return nullptr;
            return c->krow[logical ? c->kmap[row] : row];
        case ESQ_SLOT_Y: return c->y;
        case ESQ_SLOT_YNEW: return c->ynew;
        case ESQ_SLOT_YSTAGE: return c->ystage;
        case ESQ_SLOT_ATOL: return c->atolv;
        case ESQ_SLOT_WORK: return c->work;
        default: return nullptr;
    }
}

// ---- profiling -------------------------------------------------------------
// Own kernels are launched with hipExtLaunchKernelGGL(start, stop): the events
// take the begin/end timestamps of THAT dispatch packet, no extra barrier
// packets enter the queue (a hipEventRecord pair around each launch cost ~10 %
// of a Pr8 step).  Opaque RHS plugins are bracketed with hipEventRecord.
struct Prof {
    esq_ctx *c;
    bool on, recorded;
    ProfEvent ev;
    Prof(esq_ctx *ctx, int klass, const char *name, int nt, double bytes,
         bool record_now = false, double moved = -1.0)
        : c(ctx), on((ctx->prof_mask >> klass) & 1u), recorded(record_now) {
        ev.start = ev.stop = nullptr;
        ev.name[0] = 0;
        if (on && ctx->prof_every > 1) {
            // pseudo-random 1-in-`every` sampling: a fixed stride would alias
            // with the number of launches per step (e.g. 14 for Pr8, stride 7)
            unsigned x = (unsigned)(ctx->prof_seen[klass]++) * 2654435761u;
            x ^= x >> 15;
            x *= 2246822519u;
            x ^= x >> 13;
            on = (x % ctx->prof_every) == 0;
        }
        if (!on) return;
        auto take = [&]() {
            hipEvent_t e;
            if (!c->prof_pool.empty()) {
                e = c->prof_pool.back();
                c->prof_pool.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        ev.start = take();
        ev.stop = take();
        ev.klass = klass;
        ev.bytes = bytes;
        ev.moved = moved < 0.0 ? bytes : moved;
        if (nt >= 0) snprintf(ev.name, sizeof(ev.name), "%s<%d>", name, nt);
        else snprintf(ev.name, sizeof(ev.name), "%s", name);
        if (recorded) (void)hipEventRecord(ev.start, c->stream);
    }
    void cancel() {            // the launch did not happen: return the events
        if (!on) return;
        c->prof_pool.push_back(ev.start);
        c->prof_pool.push_back(ev.stop);
        on = false;
    }
    hipEvent_t start() const { return on && !recorded ? ev.start : nullptr; }
    hipEvent_t stop() const { return on && !recorded ? ev.stop : nullptr; }
    ~Prof() {
        if (!on) return;
        if (recorded) (void)hipEventRecord(ev.stop, c->stream);
        c->prof_live.push_back(ev);
    }
};

void prof_drain(esq_ctx *c) {
    if (c->prof_live.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto &ev : c->prof_live) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess) {
            c->prof_ms[ev.klass] += ms;
            c->prof_cnt[ev.klass] += 1;
            c->prof_bytes[ev.klass] += ev.bytes;
            c->prof_moved[ev.klass] += ev.moved;
            ProfKernel &pk = c->prof_kernels[ev.name];
            pk.klass = ev.klass;
            pk.launches += 1;
            pk.ms += ms;
            pk.bytes += ev.bytes;
            pk.moved += ev.moved;
        }
        c->prof_pool.push_back(ev.start);
        c->prof_pool.push_back(ev.stop);
    }
    c->prof_live.clear();
}

// ---- launch helpers ----------------------------------------------------------
template <int NT, int LDP, int STP>
void launch_lincomb_p(esq_ctx *c, double *out, const double *base,
                      const double *init, const Terms &tm, double h,
                      const Prof *p) {
    hipExtLaunchKernelGGL((k_lincomb<NT, LDP, STP>), dim3(c->grid_stream),
                          dim3(kBlock), 0, c->stream, p ? p->start() : nullptr,
                          p ? p->stop() : nullptr, 0, out, base, init, tm, h,
                          c->len_pad / 2);
}
template <int NT>
void launch_lincomb_n(esq_ctx *c, double *out, const double *base,
                      const double *init, const Terms &tm, double h,
                      const Prof *p) {
    switch (c->stage_policy) {      // ESQ_STAGE_POLICY = <load><store>
        case 1:  launch_lincomb_p<NT, 0, 1>(c, out, base, init, tm, h, p); break;
        case 10: launch_lincomb_p<NT, 1, 0>(c, out, base, init, tm, h, p); break;
        case 11: launch_lincomb_p<NT, 1, 1>(c, out, base, init, tm, h, p); break;
        case 20: launch_lincomb_p<NT, 2, 0>(c, out, base, init, tm, h, p); break;
        case 21: launch_lincomb_p<NT, 2, 1>(c, out, base, init, tm, h, p); break;
        default: launch_lincomb_p<NT, 0, 0>(c, out, base, init, tm, h, p); break;
    }
}
template <int NT>
void launch_lincomb_small(esq_ctx *c, double *out, const double *base,
                          const double *init, const Terms &tm, double h,
                          const Prof *p) {
    ResultSink rs;
    rs.seq = ++c->red_seq;
    rs.dev = nullptr;
    rs.host_value = &c->h_slot->value;
    rs.host_seq = &c->h_slot->seq;
    hipExtLaunchKernelGGL((k_lincomb_small<NT>), dim3(1), dim3(kBlock), 0, c->stream,
                          p ? p->start() : nullptr, p ? p->stop() : nullptr, 0, out,
                          base, init, tm, h, c->len_pad / 2, rs);
    c->self_seq = rs.seq;
    c->self_valid = true;
}
int launch_lincomb(esq_ctx *c, double *out, const double *base, const Terms &tm,
                   int nt, double h, const Prof *p = nullptr,
                   const double *init = nullptr) {
    if (c->host_slab && c->len_pad / 2 <= 4096) {
        // small host-RHS problem: one workgroup, completion signalled in-kernel
#define CASE(N) case N: launch_lincomb_small<N>(c, out, base, init, tm, h, p); break;
        switch (nt) {
            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
            CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
            CASE(17) CASE(18) CASE(19) CASE(20)
            default: return fail(c, ESQ_EINVAL, "too many terms: %d", nt);
        }
#undef CASE
        HIPCHK(c, hipGetLastError());
        return 0;
    }
#define CASE(N) case N: launch_lincomb_n<N>(c, out, base, init, tm, h, p); break;
    switch (nt) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
        CASE(17) CASE(18) CASE(19) CASE(20)
        default: return fail(c, ESQ_EINVAL, "too many terms: %d", nt);
    }
#undef CASE
    HIPCHK(c, hipGetLastError());
    return 0;
}

template <int NT>
void launch_solerr_n(esq_ctx *c, const Terms2 &tm, double h, const Prof &p) {
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipExtLaunchKernelGGL((k_solution_error<NT, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->ynew, c->y, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    else if (c->stage_policy >= 10)
        hipExtLaunchKernelGGL((k_solution_error<NT, false, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->ynew, c->y, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    else
        hipExtLaunchKernelGGL((k_solution_error<NT, false>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->ynew, c->y, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
}
template <int NT>
void launch_errnorm_n(esq_ctx *c, const Terms &tm, double h, const Prof &p) {
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipExtLaunchKernelGGL((k_error_norm<NT, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, c->ynew, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
    else
        hipExtLaunchKernelGGL((k_error_norm<NT, false>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, c->ynew, tm, h, av,
                           c->atol_s, c->rtol, c->len_pad / 2, c->n, c->partials);
}
template <int NT>
void launch_preerr_n(esq_ctx *c, const Terms2 &tm, double h, const Prof &p) {
    const double *av = c->atol_is_vec ? c->atolv : nullptr;
    if (c->cplx)
        hipExtLaunchKernelGGL((k_pre_error<NT, true>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, tm, h, av, c->atol_s,
                           c->rtol, c->len_pad / 2, c->n, c->partials);
    else
        hipExtLaunchKernelGGL((k_pre_error<NT, false>), dim3(c->grid_reduce), dim3(kBlock), 0, c->stream, p.start(), p.stop(), 0, c->y, tm, h, av, c->atol_s,
                           c->rtol, c->len_pad / 2, c->n, c->partials);
}
#define DISPATCH_1_20(FN, nt, ...)                                              \
    switch (nt) {                                                               \
        case 1: FN<1>(__VA_ARGS__); break;   case 2: FN<2>(__VA_ARGS__); break;   \
        case 3: FN<3>(__VA_ARGS__); break;   case 4: FN<4>(__VA_ARGS__); break;   \
        case 5: FN<5>(__VA_ARGS__); break;   case 6: FN<6>(__VA_ARGS__); break;   \
        case 7: FN<7>(__VA_ARGS__); break;   case 8: FN<8>(__VA_ARGS__); break;   \
        case 9: FN<9>(__VA_ARGS__); break;   case 10: FN<10>(__VA_ARGS__); break; \
        case 11: FN<11>(__VA_ARGS__); break; case 12: FN<12>(__VA_ARGS__); break; \
        case 13: FN<13>(__VA_ARGS__); break; case 14: FN<14>(__VA_ARGS__); break; \
        case 15: FN<15>(__VA_ARGS__); break; case 16: FN<16>(__VA_ARGS__); break; \
        case 17: FN<17>(__VA_ARGS__); break; case 18: FN<18>(__VA_ARGS__); break; \
        case 19: FN<19>(__VA_ARGS__); break; case 20: FN<20>(__VA_ARGS__); break; \
        default: return fail(c, ESQ_EINVAL, "bad term count %d", nt);           \
    }

// ---- RCCL, loaded lazily so that single-GPU use never pays for it -----------
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    void *CommInitRank = nullptr;   // int (*)(ncclComm_t*, int, ncclUniqueId, int)
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct UniqueId { char bytes[128]; };
typedef int (*init_rank_fn)(void **, int, UniqueId, int);

Rccl g_rccl;
int rccl_load() {
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return ESQ_ESTATE;
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = dlsym(lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *,
                                hipStream_t))dlsym(lib, "ncclAllReduce");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    g_rccl.CommAbort = (int (*)(void *))dlsym(lib, "ncclCommAbort");
    g_rccl.CommCount = (int (*)(void *, int *))dlsym(lib, "ncclCommCount");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy ||
        !g_rccl.AllReduce)
        return ESQ_ESTATE;
    g_rccl.lib = lib;
    return 0;
}
// RCCL prints a version banner on stdout; callers (bench.py) own stdout, so the
// banner is sent to stderr instead
struct StdoutToStderr {
    int saved;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) { dup2(saved, 1); close(saved); }
    }
};
constexpr int kNcclFloat64 = 8;   // ncclDouble
constexpr int kNcclSum = 0;       // ncclSum
constexpr int kNcclMax = 2;       // ncclMax
constexpr int kNcclMin = 3;       // ncclMin

// Wait until the reduction numbered `seq` has landed in the pinned host slot.
// The GPU writes the slot itself (no copy engine, no stream-sync wake-up: the
// 8-byte D2H copy + hipStreamSynchronize pair cost ~15 us of every step), the
// host spins on it.  The stream is queried now and then so that a faulted
// kernel surfaces as an error instead of a hang; `timeout_s` > 0 bounds the
// wait (lock-step: a peer that died never arrives at the all-reduce).
int wait_slot(esq_ctx *c, unsigned long long seq, double timeout_s) {
    volatile unsigned long long *flag = &c->h_slot->seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spins = 1;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spins & 0xfff) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                // everything on the stream has finished: the slot is written
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
                return fail(c, ESQ_ESTATE, "reduction %llu finished without a result", seq);
            }
            if (q != hipErrorNotReady)
                return fail(c, (int)q, "stream failed while waiting for a reduction: %s",
                            hipGetErrorString(q));
            if (timeout_s > 0.0) {
                const double el = std::chrono::duration<double>(
                    std::chrono::steady_clock::now() - t0).count();
                if (el > timeout_s) return ESQ_ETIMEOUT;
            }
        }
    }
}

// partials -> one double on the host (all-reduced over the communicator if set)
int finish_reduction(esq_ctx *c, double *out, bool take_min = false,
                     const double *partials = nullptr, int count = -1) {
    if (!partials) partials = c->partials;
    if (count < 0) count = (int)c->grid_reduce;
    ResultSink rs;
    rs.seq = ++c->red_seq;
    rs.dev = c->comm ? c->d_result : nullptr;
    rs.host_value = c->comm ? nullptr : &c->h_slot->value;
    rs.host_seq = &c->h_slot->seq;
    if (take_min)
        hipLaunchKernelGGL(k_final_min, dim3(1), dim3(1024), 0, c->stream,
                           partials, count, rs);
    else
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(1024), 0, c->stream,
                           partials, count, rs);
    HIPCHK(c, hipGetLastError());
    if (c->comm) {
        int r = g_rccl.AllReduce(c->d_result, c->d_result, 1, kNcclFloat64,
                                 take_min ? kNcclMin : kNcclSum, c->comm,
                                 c->stream);
        if (r != 0)
            return fail(c, 1000 + r, "ncclAllReduce failed: %s",
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        rs.host_value = &c->h_slot->value;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream,
                           c->d_result, rs);
        HIPCHK(c, hipGetLastError());
    }
    int w = wait_slot(c, rs.seq, c->comm ? c->comm_timeout_s : 0.0);
    if (w == ESQ_ETIMEOUT) {
        // a peer never reached the collective: abort the communicator so that
        // this rank (and, through RCCL, the others) fails instead of hanging
        if (c->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
        c->comm = nullptr;
        return fail(c, ESQ_ETIMEOUT, "lock-step all-reduce did not complete within "
                    "%.0f s (a peer rank failed?); communicator aborted",
                    c->comm_timeout_s);
    }
    if (w) return w;
    c->idle = true;        // the final-sum kernel was the last thing enqueued
    if (out) *out = c->h_slot->value;
    return 0;
}

int call_rhs(esq_ctx *c, double t, const double *src, double *dst) {
    if (!c->rhs) return fail(c, ESQ_ESTATE, "no device RHS set (esq_set_rhs)");
    Prof p(c, ESQ_PROF_RHS, "rhs_plugin", -1, 16.0 * (double)c->len, /*record_now=*/true);
    int r = c->rhs(c->rhs_user, t, src, dst, c->len, (void *)c->stream);
    if (r != 0) return fail(c, ESQ_ERHS, "RHS plugin returned %d", r);
    return 0;
}

int build_row_terms(esq_ctx *c, const double *coef, int count, Terms &tm,
                    const std::vector<int> &map) {
    int nt = 0;
    for (int j = 0; j < count; ++j) {
        if (coef[j] == 0.0) continue;
        if (nt >= kMaxTerms) return -1;
        tm.p[nt] = c->krow[map[j]];
        tm.c[nt] = coef[j];
        ++nt;
    }
    for (int j = nt; j < kMaxTerms; ++j) { tm.p[j] = nullptr; tm.c[j] = 0.0; }
    return nt;
}
int build_row_terms2(esq_ctx *c, const double *b, int nb, const double *e, int ne,
                     Terms2 &tm, const std::vector<int> &map) {
    int nt = 0;
    const int count = nb > ne ? nb : ne;
    for (int j = 0; j < count; ++j) {
        const double bj = j < nb ? b[j] : 0.0, ej = j < ne ? e[j] : 0.0;
        if (bj == 0.0 && ej == 0.0) continue;
        if (nt >= kMaxTerms) return -1;
        tm.p[nt] = c->krow[map[j]];
        tm.b[nt] = bj;
        tm.e[nt] = ej;
        ++nt;
    }
    for (int j = nt; j < kMaxTerms; ++j) { tm.p[j] = nullptr; tm.b[j] = tm.e[j] = 0.0; }
    return nt;
}

// ---- blocked accumulation plan ------------------------------------------------
// words per element and step moved by the stage kernels (+ block kernels) for a
// set of column boundaries; returns -1 if a block needs too many outputs/rows
int plan_words(const std::vector<double> &A, int s, const std::vector<int> &bounds,
               std::vector<esq_ctx::Block> *out) {
    auto nz = [&](int i, int j) { return A[(size_t)i * s + j] != 0.0; };
    std::vector<char> has(s, 0);
    int total = 0, prev = 0;
    if (out) out->clear();
    for (int J : bounds) {
        esq_ctx::Block b;
        b.J = J;
        b.prev = prev;
        std::vector<char> col(s, 0);
        for (int i = J; i < s; ++i) {
            bool any = false;
            for (int j = prev; j < J; ++j)
                if (nz(i, j)) { any = true; col[j] = 1; }
            if (any) b.stages.push_back(i);
        }
        for (int j = prev; j < J; ++j)
            if (col[j]) b.cols.push_back(j);
        if ((int)b.stages.size() > kMaxOut || (int)b.cols.size() > kMaxTerms) return -1;
        if (b.stages.empty()) return -1;
        total += (int)b.cols.size();
        for (int i : b.stages) {
            total += 1 + (has[i] ? 1 : 0);
            has[i] = 1;
        }
        if (out) out->push_back(b);
        prev = J;
    }
    for (int i = 1; i < s; ++i) {
        int last = 0;
        for (int J : bounds)
            if (J <= i) last = J;
        int c = 2;
        for (int j = last; j < i; ++j) c += nz(i, j);
        if (last > 0 && has[i]) c += 1;
        total += c;
    }
    return total;
}

// One process per GPU is the intended use, but a process MAY hold contexts on
// several devices and drive a context from any thread: hipSetDevice is
// per-thread state and costs well under a microsecond, so every entry point
// selects the context's device unconditionally.
// The first stage argument formed ahead of time by esq_rk_accept lives in
// YSTAGE until the next esq_rk_stages: any entry point that may write a vector
// drops it (ENTER); the read-only ones keep it (ENTER_KEEP).
#define ENTER_KEEP(c) (void)hipSetDevice((c)->device)
#define ENTER(c)                             \
    do {                                     \
        (void)hipSetDevice((c)->device);     \
        (c)->pre_valid = false;              \
        (c)->idle = false;                   \
        (c)->self_valid = false;             \
    } while (0)

// ---- host-slab mode ------------------------------------------------------------
bool in_host_slab(const esq_ctx *c, const void *dev) {
    const double *p = (const double *)dev;
    return c->host_slab && p >= c->slab && p < c->slab + c->slab_doubles;
}
double *host_of(const esq_ctx *c, const void *dev) {
    return c->slab_host + ((const double *)dev - c->slab);
}
// every kernel enqueued so far has finished (its writes to the pinned slab are
// visible to the host): a one-thread kernel bumps the pinned sequence number
// behind them, the host spins on it
int host_wait(esq_ctx *c, bool already_idle) {
    if (already_idle) return 0;
    if (c->self_valid) {               // the last kernel signals its own completion
        c->self_valid = false;
        const int w = wait_slot(c, c->self_seq, 0.0);
        if (w) return w;
        c->idle = true;
        return 0;
    }
    ResultSink rs;
    rs.seq = ++c->red_seq;
    rs.dev = nullptr;
    rs.host_value = &c->h_slot->value;
    rs.host_seq = &c->h_slot->seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, c->d_result, rs);
    HIPCHK(c, hipGetLastError());
    const int w = wait_slot(c, rs.seq, 0.0);
    if (w) return w;
    c->idle = true;
    return 0;
}

// Device-to-host copy into a caller's (pageable) buffer.  Large copies pin the
// destination for the duration of the call: measured for 80 MB into a fresh
// NumPy array 4.3 ms (register 2.8 + copy 1.5 at 54 GB/s) against 6.5-7 ms for
// the staged pageable copy.
int d2h(esq_ctx *c, void *host, const void *dev, size_t bytes, bool was_idle = false) {
    if (in_host_slab(c, dev)) {
        const int w = host_wait(c, was_idle);
        if (w) return w;
        memcpy(host, host_of(c, dev), bytes);
        return 0;
    }
    bool pinned = false;
    if (bytes >= ((size_t)8 << 20))
        pinned = hipHostRegister(host, bytes, hipHostRegisterDefault) == hipSuccess;
    hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (pinned) (void)hipHostUnregister(host);
    if (e != hipSuccess)
        return fail(c, (int)e, "device-to-host copy failed: %s", hipGetErrorString(e));
    c->idle = true;
    return 0;
}
// host-to-device copy of a caller's buffer, synchronous
int h2d(esq_ctx *c, void *dev, const void *host, size_t bytes, bool was_idle) {
    if (in_host_slab(c, dev)) {
        // no kernel may still be reading the row that is overwritten
        const int w = host_wait(c, was_idle);
        if (w) return w;
        memcpy(host_of(c, dev), host, bytes);
        c->idle = true;
        return 0;
    }
    HIPCHK(c, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->idle = true;
    return 0;
}

unsigned env_uint(const char *name, unsigned dflt) {
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    char *end = nullptr;
    long v = strtol(s, &end, 10);
    return (end != s && v >= 0) ? (unsigned)v : dflt;
}

}  // namespace

template <int NT>
void launch_block_n(esq_ctx *c, const BlockArgs &a, int no, const Prof &p) {
    hipExtLaunchKernelGGL(k_block_acc<NT>, dim3(c->grid_block), dim3(kBlock), 0,
                          c->stream, p.start(), p.stop(), 0, a, no,
                          c->len_pad / 2);
}
// leading parts of the sums of all later stages, one pass over the block's rows
// returns 0 on success; *made_ystage = true if the block also wrote the
// boundary stage's argument into YSTAGE
static int run_block(esq_ctx *c, const esq_ctx::Block &b, double h,
                     bool *made_ystage) {
    BlockArgs a;
    const int nt = (int)b.cols.size(), no = (int)b.stages.size();
    for (int j = 0; j < kMaxTerms; ++j) {
        a.p[j] = j < nt ? c->krow[c->kmap[b.cols[j]]] : nullptr;
        for (int o = 0; o < kMaxOut; ++o)
            a.w[j][o] = (j < nt && o < no)
                            ? c->A[(size_t)b.stages[o] * c->s + b.cols[j]] : 0.0;
    }
    double reads = nt;
    for (int o = 0; o < kMaxOut; ++o) {
        a.out[o] = o < no ? c->krow[b.out_vec[o]] : nullptr;
        a.init[o] = (o < no && b.in_vec[o] >= 0) ? c->krow[b.in_vec[o]] : nullptr;
        if (a.init[o]) reads += 1;
    }
    // the boundary stage J itself (always output 0 when it uses the block) has
    // no later column to add: write its argument y + h*sum straight to YSTAGE
    a.y = nullptr;
    a.h = h;
    double alg = 0.0;
    *made_ystage = false;
    static const bool fold = env_uint("ESQ_BLOCK_FOLD", 1) != 0;
    if (fold && no > 0 && b.stages[0] == b.J && c->stage_init[b.J] == b.out_vec[0] &&
        c->stage_from[b.J] == b.J) {
        a.y = c->y;
        a.out[0] = c->ystage;
        reads += 1;
        int nnz_all = 0;
        for (int j = 0; j < b.J; ++j) nnz_all += c->A[(size_t)b.J * c->s + j] != 0.0;
        alg = 8.0 * (nnz_all + 2) * (double)c->len;   // that stage's booking
        *made_ystage = true;
    }
    // algorithmic bytes: only the folded-in stage (the other partial sums are
    // booked on the stages they serve); moved bytes: its real traffic
    Prof p(c, ESQ_PROF_STAGE, "k_block_acc", nt, alg, false,
           8.0 * (reads + no) * (double)c->len);
    DISPATCH_1_20(launch_block_n, nt, c, a, no, p)
    HIPCHK(c, hipGetLastError());
    return 0;
}

extern "C" {

int esq_abi_version(void) { return ESQ_ABI_VERSION; }

int esq_device_count(int *count_out) {
    if (!count_out) return ESQ_EINVAL;
    const hipError_t e = hipGetDeviceCount(count_out);
    return e == hipSuccess ? 0 : (int)e;
}

int esq_create(esq_ctx **out, int device, size_t n, int n_rows, int is_complex) {
    return esq_create2(out, device, n, n_rows, is_complex, 0);
}
int esq_create2(esq_ctx **out, int device, size_t n, int n_rows, int is_complex,
                int flags) {
    if (!out || n_rows < 1 || n_rows > 64) return ESQ_EINVAL;
    esq_ctx *c = new (std::nothrow) esq_ctx();
    if (!c) return ESQ_ENOMEM;
    *out = c;   // returned even on failure so the caller can read the message
    c->device = device;
    c->n = n;
    c->cplx = is_complex != 0;
    c->len = c->cplx ? 2 * n : n;
    c->len_pad = ((c->len + kPadDoubles - 1) / kPadDoubles) * kPadDoubles;
    if (c->len_pad == 0) c->len_pad = kPadDoubles;
    // ESQ_ROW_STAGGER: extra bytes between consecutive vectors (HBM channel
    // de-aliasing experiments); must be a multiple of 16
    const size_t stagger = (env_uint("ESQ_ROW_STAGGER", 0) / 16) * 2;
    c->stride = c->len_pad + stagger;
    c->n_rows = n_rows;
    HIPCHK(c, hipSetDevice(device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t nvec = (size_t)n_rows + kFixedSlots;
    const size_t slab_doubles = nvec * c->stride + kPartialsCap + kMaxPartials + 64;
    c->slab_doubles = slab_doubles;
    c->host_slab = (flags & ESQ_CREATE_HOST_SLAB) != 0;
    if (c->host_slab) {
        HIPCHK(c, hipHostMalloc((void **)&c->slab_host, slab_doubles * sizeof(double),
                                hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->slab_host, 0, slab_doubles * sizeof(double));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->slab, c->slab_host, 0));
    } else {
        HIPCHK(c, hipMalloc(&c->slab, slab_doubles * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->slab, 0, slab_doubles * sizeof(double), c->stream));
    }
    c->krow.resize(n_rows);
    c->kmap.resize(n_rows);
    for (int r = 0; r < n_rows; ++r) {
        c->krow[r] = c->slab + (size_t)r * c->stride;
        c->kmap[r] = r;
    }
    c->kmap_last = c->kmap;
    double *base = c->slab + (size_t)n_rows * c->stride;
    c->y = base;
    c->ynew = base + c->stride;
    c->ystage = base + 2 * c->stride;
    c->atolv = base + 3 * c->stride;
    c->work = base + 4 * c->stride;
    c->partials = base + 5 * c->stride;
    c->partials2 = c->partials + kPartialsCap;
    c->d_result = c->partials2 + kMaxPartials;
    HIPCHK(c, hipHostMalloc((void **)&c->h_slot, 64,
                            hipHostMallocMapped | hipHostMallocCoherent));
    c->h_slot->value = 0.0;
    c->h_slot->seq = 0;
    c->comm_timeout_s = (double)env_uint("ESQ_COMM_TIMEOUT_S", 120);
    c->epi_nt = env_uint("ESQ_EPI_NT", 0x3);
    // launch geometry: grid-stride kernels, a few resident blocks per CU
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    const unsigned cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t n2 = c->len_pad / 2;
    const size_t need = (n2 + kBlock - 1) / kBlock;
    // Cache policy (measured): a working set beyond the 256 MiB
    // Infinity Cache streams its K rows with non-temporal loads so that y and
    // the stage argument stay on-die (2 workgroups per CU); a working set that
    // fits is left to the cache (plain loads, 8 workgroups per CU).
    const bool fits_mall = slab_doubles * sizeof(double) <= (size_t)160 << 20;
    // The first sweep of a step may form its own input from y and K[0]
    // (ESQ_FUSE_SRC): one launch and 16 B per element less, but a second row
    // window through L2.  Measured (profiles/r02_experiments.md): Ts5 at n = 1e6
    // 0.0834 -> 0.0778 ms/step, Pr8 at n = 1e7 unchanged, Pr9 at n = 5e6 +0.8 %:
    // used where the working set is cache-resident.  ESQ_SRC=0|1 overrides.
    c->src_pays = env_uint("ESQ_SRC", fits_mall ? 1 : 0) != 0;
    const unsigned per_cu = env_uint("ESQ_BLOCKS_PER_CU", fits_mall ? 8 : 2);
    c->stage_policy = (int)env_uint("ESQ_STAGE_POLICY", fits_mall ? 0 : 10);
    size_t g = (size_t)cus * per_cu;
    if (g > need) g = need;
    if (g < 1) g = 1;
    c->grid_stream = (unsigned)g;
    c->grid_reduce = (unsigned)(g > (size_t)kMaxPartials ? kMaxPartials : g);
    size_t gb = (size_t)cus * env_uint("ESQ_BLOCK_BPC", 16);
    if (gb > need) gb = need;
    c->grid_block = (unsigned)(gb < 1 ? 1 : gb);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int esq_destroy(esq_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &ev : c->prof_live) { (void)hipEventDestroy(ev.start); (void)hipEventDestroy(ev.stop); }
    for (auto &e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->host_slab) {
        if (c->slab_host) (void)hipHostFree(c->slab_host);
    } else if (c->slab) {
        (void)hipFree(c->slab);
    }
    for (double *p : c->aux_slabs) (void)hipFree(p);
    if (c->h_slot) (void)hipHostFree(c->h_slot);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

const char *esq_last_error(const esq_ctx *c) { return c ? c->err : "null context"; }

int esq_synchronize(esq_ctx *c) {
    if (!c) return ESQ_EINVAL;
    ENTER_KEEP(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->idle = true;
    return 0;
}
size_t esq_vector_len(const esq_ctx *c) { return c ? c->len : 0; }

int esq_upload(esq_ctx *c, int slot, int row, const double *host) {
    if (!c || !host) return ESQ_EINVAL;
    const bool was_idle = c->idle;
    ENTER(c);
    double *d = slot_ptr(c, slot, row);
    if (!d) return fail(c, ESQ_EINVAL, "bad slot/row %d/%d", slot, row);
    const size_t cnt = (slot == ESQ_SLOT_ATOL) ? c->n : c->len;
    return h2d(c, d, host, cnt * sizeof(double), was_idle);
}
int esq_download(esq_ctx *c, int slot, int row, double *host) {
    if (!c || !host) return ESQ_EINVAL;
    ENTER_KEEP(c);
    double *d = slot_ptr(c, slot, row);
    if (!d) return fail(c, ESQ_EINVAL, "bad slot/row %d/%d", slot, row);
    const size_t cnt = (slot == ESQ_SLOT_ATOL) ? c->n : c->len;
    return d2h(c, host, d, cnt * sizeof(double), c->idle);
}
int esq_copy(esq_ctx *c, int dst_slot, int dst_row, int src_slot, int src_row) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *d = slot_ptr(c, dst_slot, dst_row), *s = slot_ptr(c, src_slot, src_row);
    if (!d || !s) return fail(c, ESQ_EINVAL, "bad slot/row");
    HIPCHK(c, hipMemcpyAsync(d, s, c->len_pad * sizeof(double),
                             hipMemcpyDefault, c->stream));
    return 0;
}

int esq_rk_set_tableau(esq_ctx *c, int s, const double *A, const double *B,
                       const double *C, const double *E, int fsal) {
    if (!c || !A || !B || !C || !E || s < 1) return ESQ_EINVAL;
    ENTER(c);
    if (s + 1 > c->n_rows)
        return fail(c, ESQ_EINVAL, "tableau needs %d rows, context has %d", s + 1,
                    c->n_rows);
    for (int i = 0; i < s; ++i) {
        int nz = 0;
        for (int j = 0; j < s; ++j) {
            if (j >= i && A[i * s + j] != 0.0)
                return fail(c, ESQ_EINVAL, "A must be strictly lower triangular");
            nz += A[i * s + j] != 0.0;
        }
        if (nz > kMaxTerms)
            return fail(c, ESQ_EINVAL, "row %d of A has %d > %d nonzeros", i, nz, kMaxTerms);
    }
    c->s = s;
    c->fsal = fsal ? 1 : 0;
    c->A.assign(A, A + (size_t)s * s);
    c->B.assign(B, B + s);
    c->C.assign(C, C + s);
    c->E.assign(E, E + s + 1);
    c->have_tab = true;
    // ---- blocked accumulation plan (up to 3 column boundaries, exhaustive)
    c->blocks.clear();
    c->stage_init.assign(s, -1);
    c->stage_from.assign(s, 0);
    if (env_uint("ESQ_BLOCK_ACC", 1) != 0 && s >= 4) {
        std::vector<int> best;
        int best_words = plan_words(c->A, s, best, nullptr);
        // fewest boundaries first: a plan with more boundaries must be strictly
        // better (every boundary is one more launch)
        for (int b1 = 2; b1 < s; ++b1) {
            const int w = plan_words(c->A, s, {b1}, nullptr);
            if (w >= 0 && w < best_words) { best_words = w; best = {b1}; }
        }
        for (int b1 = 2; b1 < s; ++b1)
            for (int b2 = b1 + 1; b2 < s; ++b2) {
                const int w = plan_words(c->A, s, {b1, b2}, nullptr);
                if (w >= 0 && w < best_words) { best_words = w; best = {b1, b2}; }
            }
        for (int b1 = 2; b1 < s; ++b1)
            for (int b2 = b1 + 1; b2 < s; ++b2)
                for (int b3 = b2 + 1; b3 < s; ++b3) {
                    const int w = plan_words(c->A, s, {b1, b2, b3}, nullptr);
                    if (w >= 0 && w < best_words) { best_words = w; best = {b1, b2, b3}; }
                }
        // ESQ_BLOCK_BOUNDS="6,10": override the boundaries (tuning experiments)
        if (const char *ov = getenv("ESQ_BLOCK_BOUNDS")) {
            std::vector<int> forced;
            for (const char *q = ov; *q;) {
                char *end = nullptr;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                if (v >= 2 && v < s) forced.push_back((int)v);
                q = *end ? end + 1 : end;
            }
            if (plan_words(c->A, s, forced, nullptr) >= 0) best = forced;
        }
        if (!best.empty()) {
            std::vector<esq_ctx::Block> blocks;
            plan_words(c->A, s, best, &blocks);
            int count = 0;
            for (auto &bl : blocks) count += (int)bl.stages.size();
            // partial-sum rows: those of an earlier plan on this context are
            // reused (a second esq_rk_set_tableau must not leak a slab)
            int first = c->block_rows_first;
            if (count > c->block_rows_count) {
                int r = esq_aux_rows(c, count, &first);
                if (r) return r;
                c->block_rows_first = first;
                c->block_rows_count = count;
            }
            std::vector<int> cur(s, -1);
            for (auto &bl : blocks) {
                for (int i : bl.stages) {
                    bl.in_vec.push_back(cur[i]);
                    bl.out_vec.push_back(first);
                    cur[i] = first++;
                }
            }
            c->blocks = blocks;
            for (int i = 1; i < s; ++i) {
                int last = 0;
                for (int J : best)
                    if (J <= i) last = J;
                c->stage_from[i] = cur[i] >= 0 ? last : 0;
                c->stage_init[i] = cur[i] >= 0 && last > 0 ? cur[i] : -1;
                if (c->stage_init[i] < 0) c->stage_from[i] = 0;
            }
        }
    }
    return 0;
}

int esq_set_tol(esq_ctx *c, double rtol, const double *atol, size_t n_atol) {
    if (!c || !atol) return ESQ_EINVAL;
    ENTER(c);
    c->rtol = rtol;
    if (n_atol == 1) {
        c->atol_s = atol[0];
        c->atol_is_vec = false;
        return 0;
    }
    if (n_atol != c->n) return fail(c, ESQ_EINVAL, "atol has %zu entries, n = %zu", n_atol, c->n);
    c->atol_is_vec = true;
    return h2d(c, c->atolv, atol, c->n * sizeof(double), false);
}

int esq_set_rhs(esq_ctx *c, esq_rhs_fn fn, void *user) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs = fn;
    c->rhs_user = user;
    c->rhs_fused = nullptr;
    c->fuse_mask = 0;
    c->rhs_rkc = nullptr;
    return 0;
}
int esq_set_rhs_rkc(esq_ctx *c, esq_rhs_rkc_fn fn) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs_rkc = fn;
    return 0;
}
int esq_set_rhs_fused(esq_ctx *c, esq_rhs_fused_fn fn, int fuse_mask) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs_fused = fn;
    c->fuse_mask = fn ? fuse_mask : 0;
    c->src_declined = false;
    return 0;
}

int esq_rk_stage_accumulate(esq_ctx *c, int i, double h) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (i < 1 || i >= c->s) return fail(c, ESQ_EINVAL, "stage %d out of range", i);
    for (const auto &b : c->blocks)
        if (b.J == i) {
            bool made = false;
            const int r = run_block(c, b, h, &made);
            if (r) return r;
            if (made) return 0;        // YSTAGE already holds this stage's argument
        }
    const int from = c->stage_from[i];
    const double *init = c->stage_init[i] >= 0 ? c->krow[c->stage_init[i]] : nullptr;
    Terms tm;
    // columns [from, i): the chain resumes from the stored partial sum
    std::vector<double> row(c->A.begin() + (size_t)i * c->s,
                            c->A.begin() + (size_t)i * c->s + i);
    int nnz_all = 0;
    for (int j = 0; j < i; ++j) {
        nnz_all += row[j] != 0.0;
        if (j < from) row[j] = 0.0;
    }
    const int nt = build_row_terms(c, row.data(), i, tm, c->kmap);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    Prof p(c, ESQ_PROF_STAGE, "k_lincomb", nt, 8.0 * (nnz_all + 2) * (double)c->len,
           false, 8.0 * (nt + 2 + (init ? 1 : 0)) * (double)c->len);
    return launch_lincomb(c, c->ystage, c->y, tm, nt, h, &p, init);
}

int esq_rk_block_plan(esq_ctx *c, int *boundaries, int max_boundaries,
                      int *words_plain, int *words_blocked) {
    if (!c || !c->have_tab) return ESQ_EINVAL;
    std::vector<int> b;
    for (const auto &bl : c->blocks) b.push_back(bl.J);
    if (words_plain) *words_plain = plan_words(c->A, c->s, {}, nullptr);
    if (words_blocked) *words_blocked = plan_words(c->A, c->s, b, nullptr);
    for (int k = 0; k < (int)b.size() && k < max_boundaries; ++k)
        if (boundaries) boundaries[k] = b[k];
    return (int)b.size();
}

int esq_rk_eval_rhs(esq_ctx *c, int dst_row, double t, int src_slot, int src_row) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    double *dst = slot_ptr(c, ESQ_SLOT_K, dst_row);
    double *src = slot_ptr(c, src_slot, src_row);
    if (!dst || !src) return fail(c, ESQ_EINVAL, "bad row/slot");
    return call_rhs(c, t, src, dst);
}

}  // extern "C"

namespace {

// coefficient row of stage i as the stage kernels use it: columns below the
// stage's blocked-accumulation boundary are in its stored partial sum, column
// `skip` (if >= 0) comes from registers.  Returns the number of rows to read.
int stage_terms(esq_ctx *c, int i, int skip, Terms &tm, const double **init,
                int *nnz_all, double *c_skip) {
    const int from = c->stage_from[i];
    *init = c->stage_init[i] >= 0 ? c->krow[c->stage_init[i]] : nullptr;
    const double *row = &c->A[(size_t)i * c->s];
    int nt = 0, nnz = 0;
    if (c_skip) *c_skip = 0.0;
    for (int j = 0; j < i; ++j) {
        if (row[j] == 0.0) continue;
        ++nnz;
        if (j < from) continue;
        if (j == skip) { if (c_skip) *c_skip = row[j]; continue; }
        if (nt >= kMaxTerms) return -1;
        tm.p[nt] = c->krow[c->kmap[j]];
        tm.c[nt] = row[j];
        ++nt;
    }
    for (int j = nt; j < kMaxTerms; ++j) { tm.p[j] = nullptr; tm.c[j] = 0.0; }
    *nnz_all = nnz;
    return nt;
}

void epi_common(esq_ctx *c, esq_epilogue &e, int kind) {
    memset(&e, 0, sizeof(e));
    e.kind = kind;
    e.atol_vec = c->atol_is_vec ? c->atolv : nullptr;
    e.atol_s = c->atol_s;
    e.rtol = c->rtol;
    e.n_valid = c->n;
    e.partials = c->partials;
    e.partials_cap = kPartialsCap;
    e.partials_used = &c->red_count;
}

// one fused sweep; returns 0, ESQ_ENOTSUP (caller falls back) or an error
int run_fused(esq_ctx *c, double t, const double *y_in, double *f_out,
              const esq_epilogue &e, Prof &p) {
    const int r = c->rhs_fused(c->rhs_user, t, y_in, f_out, &e, c->len,
                               (void *)c->stream, (void *)p.start(),
                               (void *)p.stop());
    if (r == ESQ_ENOTSUP) { p.cancel(); return r; }
    if (r != 0) { p.cancel(); return fail(c, ESQ_ERHS, "fused RHS entry returned %d", r); }
    return 0;
}
bool may_fuse(const esq_ctx *c, int kind) {
    return c->rhs_fused && ((c->fuse_mask >> kind) & 1);
}

// RHS sweep of stage i + the accumulate of stage nx = i + 1 (ESQ_EPI_STAGE).
// from_state (stage 1 only): the sweep forms its own input y + h*a_10*K[0] on
// the fly instead of reading YSTAGE (ESQ_FUSE_SRC)
int sweep_next_stage(esq_ctx *c, int i, double t, double h, bool from_state = false) {
    const int nx = i + 1;
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_STAGE);
    Terms tm;
    int nnz_all = 0;
    const int nt = stage_terms(c, nx, i, tm, &e.init, &nnz_all, &e.c_self);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    e.nt = nt;
    for (int j = 0; j < nt; ++j) { e.rows[j] = tm.p[j]; e.c[j] = tm.c[j]; }
    e.y = c->y;
    e.h = h;
    e.out = c->work;
    e.f_store_nt = c->epi_nt & 1;   // K_i is consumed from registers, not re-read soon
    if (from_state) {
        if (nt > 1 || e.init) return ESQ_ENOTSUP;
        e.in_base = c->y;
        e.in_row = c->krow[c->kmap[0]];
        e.in_c = c->A[(size_t)c->s];                 // A[1][0]
        e.in_h = h;
        // booked: stage 1's accumulate (1 + 2 words) + the RHS + stage 2's
        // accumulate; moved: y and K[0] in (stage 2's row K[0] is the same
        // vector), K[1] and the argument of stage 2 out
        Prof p(c, ESQ_PROF_STAGE, "rhs1+stage", nt,
               8.0 * (3 + nnz_all + 4) * (double)c->len, false, 8.0 * 4 * (double)c->len);
        const int r = run_fused(c, t + c->C[i] * h, nullptr, c->krow[c->kmap[i]], e, p);
        if (r == 0) std::swap(c->ystage, c->work);
        return r;
    }
    // booked on the stage class: next stage's algorithmic bytes + the RHS's
    // 16 B; moved: ys_in, rows, init, y in; K[i], ys_out out
    Prof p(c, ESQ_PROF_STAGE, "rhs+stage", nt, 8.0 * (nnz_all + 4) * (double)c->len,
           false, 8.0 * (nt + 4 + (e.init ? 1 : 0)) * (double)c->len);
    const int r = run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
    if (r == 0) std::swap(c->ystage, c->work);   // double buffer
    return r;
}
bool may_use_src(const esq_ctx *c) {
    return c->src_pays && may_fuse(c, ESQ_EPI_STAGE) && (c->fuse_mask & ESQ_FUSE_SRC) &&
           !c->src_declined;
}

// RHS sweep of stage i = J - 1 + the blocked accumulation at boundary J with
// K_i as the block's last column (ESQ_EPI_BLOCK)
int sweep_block(esq_ctx *c, const esq_ctx::Block &b, int i, double t, double h,
                bool *made_ystage) {
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_BLOCK);
    const int no = (int)b.stages.size();
    int nt = 0;
    for (int col : b.cols) {
        if (col == i) continue;
        if (nt >= ESQ_EPI_MAX_ROWS) return ESQ_ENOTSUP;
        e.rows[nt] = c->krow[c->kmap[col]];
        for (int o = 0; o < no; ++o)
            e.w[nt][o] = c->A[(size_t)b.stages[o] * c->s + col];
        ++nt;
    }
    e.nt = nt;
    e.no = no;
    double reads = nt + 1;                       // rows + the sweep's input
    for (int o = 0; o < no; ++o) {
        e.w_self[o] = c->A[(size_t)b.stages[o] * c->s + i];
        e.out_o[o] = c->krow[b.out_vec[o]];
        e.init_o[o] = b.in_vec[o] >= 0 ? c->krow[b.in_vec[o]] : nullptr;
        if (e.init_o[o]) reads += 1;
    }
    e.h = h;
    double alg = 16.0 * (double)c->len;          // the RHS itself
    *made_ystage = false;
    if (no > 0 && b.stages[0] == b.J && c->stage_init[b.J] == b.out_vec[0] &&
        c->stage_from[b.J] == b.J) {
        e.y = c->y;
        e.out_o[0] = c->work;
        reads += 1;
        int nnz_all = 0;
        for (int j = 0; j < b.J; ++j) nnz_all += c->A[(size_t)b.J * c->s + j] != 0.0;
        alg += 8.0 * (nnz_all + 2) * (double)c->len;   // the boundary stage's booking
        *made_ystage = true;
    }
    e.f_store_nt = (c->epi_nt >> 1) & 1;
    Prof p(c, ESQ_PROF_STAGE, "rhs+block", nt, alg, false,
           8.0 * (reads + no + 1) * (double)c->len);
    const int r = run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
    if (r == 0 && *made_ystage) std::swap(c->ystage, c->work);
    if (r != 0) *made_ystage = false;
    return r;
}

// FSAL pairs: RHS sweep of the last stage also forms y_new (ESQ_EPI_STAGE)
int sweep_ynew(esq_ctx *c, int i, double t, double h) {
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_STAGE);
    int nt = 0, nnz_all = 0;
    for (int j = 0; j < c->s; ++j) {
        if (c->B[j] == 0.0) continue;
        ++nnz_all;
        if (j == i) { e.c_self = c->B[j]; continue; }
        e.rows[nt] = c->krow[c->kmap[j]];
        e.c[nt] = c->B[j];
        ++nt;
    }
    e.nt = nt;
    e.y = c->y;
    e.h = h;
    e.out = c->ynew;
    e.f_store_nt = c->epi_nt & 1;
    Prof p(c, ESQ_PROF_STAGE, "rhs+stage", nt, 8.0 * (nnz_all + 4) * (double)c->len,
           false, 8.0 * (nt + 4) * (double)c->len);
    return run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
}

// non-FSAL pairs: RHS sweep of the last stage + y_new + error partial sums
int sweep_solerr(esq_ctx *c, int i, double t, double h) {
    esq_epilogue e;
    epi_common(c, e, ESQ_EPI_SOLERR);
    int nt = 0;
    for (int j = 0; j < c->s; ++j) {
        const double bj = c->B[j], ej = c->E[j];
        if (j == i) { e.c_self = bj; e.e_self = ej; continue; }
        if (bj == 0.0 && ej == 0.0) continue;
        if (nt >= ESQ_EPI_MAX_ROWS) return ESQ_ENOTSUP;
        e.rows[nt] = c->krow[c->kmap[j]];
        e.c[nt] = bj;
        e.e[nt] = ej;
        ++nt;
    }
    e.nt = nt;
    e.y = c->y;
    e.h = h;
    e.out = c->ynew;
    e.f_store_nt = (c->epi_nt >> 2) & 1;   // K_{s-1}: next read by the dense output
    // booked: the RHS's 16 B + the fused solution/error pass (rows incl. the
    // fresh one + y + y_new); moved: ys_in, rows, y in; K_i, y_new out
    Prof p(c, ESQ_PROF_SOLERR, "rhs+solerr", nt, 8.0 * (nt + 1 + 2 + 2) * (double)c->len,
           false, 8.0 * (nt + 4) * (double)c->len);
    return run_fused(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]], e, p);
}

}  // namespace

extern "C" {

int esq_rk_stages(esq_ctx *c, int i_from, int i_to, double t, double h) {
    if (!c) return ESQ_EINVAL;
    // YSTAGE may already hold the first stage's argument (esq_rk_accept)
    bool ready = i_from == 1 && c->pre_valid && c->pre_h == h;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (i_from < 1 || i_to > c->s || i_from > i_to)
        return fail(c, ESQ_EINVAL, "bad stage range [%d, %d)", i_from, i_to);
    c->ynew_ready = false;
    c->solerr_ready = false;
    bool block_done = false;   // the block at boundary i already ran in a sweep
    for (int i = i_from; i < i_to; ++i) {
        if (i == 1 && !ready && i + 1 < i_to && may_use_src(c)) {
            // the first sweep forms its own input from y and K[0]: no stage-1
            // kernel, no stage argument in memory
            bool boundary = false;
            for (const auto &b : c->blocks) boundary |= (b.J == 2);
            if (!boundary) {
                const int r = sweep_next_stage(c, 1, t, h, /*from_state=*/true);
                if (r == 0) { ready = true; continue; }
                if (r != ESQ_ENOTSUP) return r;
                c->src_declined = true;
            }
        }
        if (!ready) {
            if (block_done) {
                // partial sums are in place; only the stage kernel is left
                const double *init = nullptr;
                Terms tm;
                int nnz_all = 0;
                const int nt = stage_terms(c, i, -1, tm, &init, &nnz_all, nullptr);
                if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
                Prof p(c, ESQ_PROF_STAGE, "k_lincomb", nt,
                       8.0 * (nnz_all + 2) * (double)c->len, false,
                       8.0 * (nt + 2 + (init ? 1 : 0)) * (double)c->len);
                const int r = launch_lincomb(c, c->ystage, c->y, tm, nt, h, &p, init);
                if (r) return r;
            } else {
                const int r = esq_rk_stage_accumulate(c, i, h);
                if (r) return r;
            }
        }
        ready = false;
        block_done = false;
        const esq_ctx::Block *bnext = nullptr;
        for (const auto &b : c->blocks)
            if (b.J == i + 1) bnext = &b;
        if (i + 1 < i_to && !bnext && may_fuse(c, ESQ_EPI_STAGE)) {
            // this stage's RHS sweep also forms the NEXT stage's argument
            const int r = sweep_next_stage(c, i, t, h);
            if (r == 0) { ready = true; continue; }
            if (r != ESQ_ENOTSUP) return r;
        }
        if (i + 1 < i_to && bnext && may_fuse(c, ESQ_EPI_BLOCK)) {
            // ... or runs the blocked accumulation at the column boundary
            bool made = false;
            const int r = sweep_block(c, *bnext, i, t, h, &made);
            if (r == 0) { ready = made; block_done = !made; continue; }
            if (r != ESQ_ENOTSUP) return r;
        }
        if (i == c->s - 1 && i_to == c->s) {
            if (c->fsal && may_fuse(c, ESQ_EPI_STAGE)) {
                // FSAL pairs: the LAST stage's sweep also forms y_new
                const int r = sweep_ynew(c, i, t, h);
                if (r == 0) { c->ynew_ready = true; continue; }
                if (r != ESQ_ENOTSUP) return r;
            }
            if (!c->fsal && !c->cplx && may_fuse(c, ESQ_EPI_SOLERR)) {
                // others: ... y_new and the error partial sums
                const int r = sweep_solerr(c, i, t, h);
                if (r == 0) { c->ynew_ready = c->solerr_ready = true; continue; }
                if (r != ESQ_ENOTSUP) return r;
            }
        }
        const int r = call_rhs(c, t + c->C[i] * h, c->ystage, c->krow[c->kmap[i]]);
        if (r) return r;
    }
    return 0;
}

int esq_rk_solution(esq_ctx *c, double h) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    Terms tm;
    const int nt = build_row_terms(c, c->B.data(), c->s, tm, c->kmap);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    Prof p(c, ESQ_PROF_SOLERR, "k_lincomb", nt, 8.0 * (nt + 2) * (double)c->len);
    return launch_lincomb(c, c->ynew, c->y, tm, nt, h, &p);
}

int esq_rk_error_norm(esq_ctx *c, double h, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    Terms tm;
    const int nt = build_row_terms(c, c->E.data(), c->s + c->fsal, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "error weights are all zero");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_error_norm", nt,
               8.0 * (nt + 2) * (double)c->len);
        DISPATCH_1_20(launch_errnorm_n, nt, c, tm, h, p)
        HIPCHK(c, hipGetLastError());
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_solution_error(esq_ctx *c, double t, double h, double *sumsq_out) {
    if (!c || !sumsq_out) return ESQ_EINVAL;
    const bool ynew_ready = c->ynew_ready, solerr_ready = c->solerr_ready;
    ENTER(c);
    c->ynew_ready = c->solerr_ready = false;
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    if (c->fsal) {
        int r = 0;
        if (!ynew_ready) r = esq_rk_solution(c, h);
        if (r) return r;
        if (!c->cplx && may_fuse(c, ESQ_EPI_ERRNORM)) {
            // K[s] = f(t + h, y_new) and the error norm in ONE sweep
            esq_epilogue e;
            epi_common(c, e, ESQ_EPI_ERRNORM);
            int nt = 0;
            bool ok = true;
            for (int j = 0; j < c->s; ++j) {
                if (c->E[j] == 0.0) continue;
                if (nt >= ESQ_EPI_MAX_ROWS) { ok = false; break; }
                e.rows[nt] = c->krow[c->kmap[j]];
                e.e[nt] = c->E[j];
                ++nt;
            }
            if (ok) {
                e.nt = nt;
                e.e_self = c->E[c->s];
                e.y = c->y;
                e.h = h;
                e.f_store_nt = (c->epi_nt >> 4) & 1;   // K[s] is the next step's K[0]
                // booked: RHS 16 B + error pass (rows incl. K[s], y, y_new);
                // moved: y_new, rows, y in; K[s] out
                Prof p(c, ESQ_PROF_SOLERR, "rhs+errnorm", nt,
                       8.0 * (nt + 1 + 2 + 2) * (double)c->len, false,
                       8.0 * (nt + 3) * (double)c->len);
                r = run_fused(c, t + h, c->ynew, c->krow[c->kmap[c->s]], e, p);
                if (r == 0) return finish_reduction(c, sumsq_out, false, c->partials,
                                                    c->red_count);
                if (r != ESQ_ENOTSUP) return r;
            }
        }
        r = call_rhs(c, t + h, c->ynew, c->krow[c->kmap[c->s]]);
        if (r) return r;
        return esq_rk_error_norm(c, h, sumsq_out);
    }
    if (solerr_ready)
        return finish_reduction(c, sumsq_out, false, c->partials, c->red_count);
    Terms2 tm;
    const int nt = build_row_terms2(c, c->B.data(), c->s, c->E.data(), c->s, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_solution_error", nt,
               8.0 * (nt + 2) * (double)c->len);
        DISPATCH_1_20(launch_solerr_n, nt, c, tm, h, p)
        HIPCHK(c, hipGetLastError());
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_pre_error(esq_ctx *c, double h, const double *e_pre,
                     const double *b_scale_pre, int rows, double *sumsq_out) {
    if (!c || !e_pre || !b_scale_pre || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    if (rows < 1 || rows > c->n_rows) return fail(c, ESQ_EINVAL, "bad rows %d", rows);
    Terms2 tm;
    const int nt = build_row_terms2(c, b_scale_pre, rows, e_pre, rows, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_pre_error", nt,
               8.0 * (nt + 1) * (double)c->len);
        DISPATCH_1_20(launch_preerr_n, nt, c, tm, h, p)
        HIPCHK(c, hipGetLastError());
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_custom_sol_err(esq_ctx *c, double h, const double *b, const double *e,
                          int rows, int store_ynew, double *sumsq_out) {
    if (!store_ynew) return esq_rk_pre_error(c, h, e, b, rows, sumsq_out);
    if (!c || !b || !e || !sumsq_out) return ESQ_EINVAL;
    ENTER(c);
    if (rows < 1 || rows > c->n_rows) return fail(c, ESQ_EINVAL, "bad rows %d", rows);
    Terms2 tm;
    const int nt = build_row_terms2(c, b, rows, e, rows, tm, c->kmap);
    if (nt < 1) return fail(c, ESQ_EINVAL, "bad weights");
    {
        Prof p(c, ESQ_PROF_SOLERR, "k_solution_error", nt,
               8.0 * (nt + 2) * (double)c->len);
        DISPATCH_1_20(launch_solerr_n, nt, c, tm, h, p)
        HIPCHK(c, hipGetLastError());
    }
    return finish_reduction(c, sumsq_out);
}

int esq_rk_accept(esq_ctx *c, double t_new, int with_end_eval, double h_next) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    // the next step's first stage argument can be formed now: stage 1 reads
    // nothing but y and K[0]
    const bool want_pre = h_next != 0.0 && c->s >= 2 && c->rhs != nullptr &&
                          !(may_use_src(c) && c->s >= 3);
    bool pre_done = false;
    if (!c->fsal && with_end_eval) {
        int r = ESQ_ENOTSUP;
        if (want_pre && may_fuse(c, ESQ_EPI_STAGE)) {
            // K[s] = f(t_new, y_new) and YSTAGE = y_new + h_next*a_10*K[s] in
            // ONE sweep (K[s] becomes K[0], y_new becomes y below)
            esq_epilogue e;
            epi_common(c, e, ESQ_EPI_STAGE);
            e.nt = 0;
            e.c_self = c->A[(size_t)c->s];          // A[1][0]
            e.y = nullptr;                          // base = the sweep's input
            e.h = h_next;
            e.out = c->ystage;
            e.f_store_nt = (c->epi_nt >> 3) & 1;    // K[0] of the next step
            const int nnz = e.c_self != 0.0 ? 1 : 0;
            Prof p(c, ESQ_PROF_STAGE, "rhs+stage", 0, 8.0 * (nnz + 4) * (double)c->len,
                   false, 8.0 * 3 * (double)c->len);
            r = run_fused(c, t_new, c->ynew, c->krow[c->kmap[c->s]], e, p);
            if (r == 0) pre_done = true;
            else if (r != ESQ_ENOTSUP) return r;
        }
        if (r == ESQ_ENOTSUP) {
            r = call_rhs(c, t_new, c->ynew, c->krow[c->kmap[c->s]]);
            if (r) return r;
        }
    }
    c->kmap_last = c->kmap;
    std::swap(c->kmap[0], c->kmap[c->s]);
    std::swap(c->y, c->ynew);
    if (want_pre && !pre_done) {
        // stage 1's accumulate, launched now: it runs while the host controller
        // is between steps
        const int r = esq_rk_stage_accumulate(c, 1, h_next);
        if (r) return r;
        pre_done = true;
    }
    c->pre_valid = pre_done;
    c->pre_h = h_next;
    return 0;
}

int esq_rk_error_vector(esq_ctx *c, double h, int last_step) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (!c->have_tab) return fail(c, ESQ_ESTATE, "no tableau set");
    Terms tm;
    const int nt = build_row_terms(c, c->E.data(), c->s + c->fsal, tm,
                                   last_step ? c->kmap_last : c->kmap);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    return launch_lincomb(c, c->work, nullptr, tm, nt, h);
}

int esq_rk_row_id(esq_ctx *c, int logical_row, int last_step) {
    if (!c || logical_row < 0 || logical_row >= c->n_rows) return ESQ_EINVAL;
    ENTER_KEEP(c);
    return last_step ? c->kmap_last[logical_row] : c->kmap[logical_row];
}
int esq_rk_download_last_K(esq_ctx *c, int row, double *host) {
    if (!c || !host) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (row < 0 || row >= c->n_rows) return fail(c, ESQ_EINVAL, "bad row %d", row);
    return d2h(c, host, c->krow[c->kmap_last[row]], c->len * sizeof(double), c->idle);
}

int esq_rk_dense_stage(esq_ctx *c, int row, const double *a, int count, double h) {
    if (!c || !a) return ESQ_EINVAL;
    ENTER(c);
    if (row < 1 || row >= c->n_rows || count < 0 || count > row)
        return fail(c, ESQ_EINVAL, "bad row/count %d/%d", row, count);
    Terms tm;
    const int nt = build_row_terms(c, a, count, tm, c->kmap_last);
    if (nt < 0) return fail(c, ESQ_EINVAL, "too many terms");
    // after esq_rk_accept the pre-step state is in the YNEW slot
    return launch_lincomb(c, c->ystage, c->ynew, tm, nt, h);
}
int esq_rk_dense_eval(esq_ctx *c, int row, double t) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (row < 0 || row >= c->n_rows) return fail(c, ESQ_EINVAL, "bad row %d", row);
    return call_rhs(c, t, c->ystage, c->krow[c->kmap_last[row]]);
}
int esq_rk_upload_last_K(esq_ctx *c, int row, const double *host) {
    if (!c || !host) return ESQ_EINVAL;
    const bool was_idle = c->idle;
    ENTER(c);
    if (row < 0 || row >= c->n_rows) return fail(c, ESQ_EINVAL, "bad row %d", row);
    return h2d(c, c->krow[c->kmap_last[row]], host, c->len * sizeof(double), was_idle);
}

// ---- device-resident interpolant ------------------------------------------------
}  // extern "C"
struct esq_dense {
    int device = 0;
    size_t len = 0, len_pad = 0;
    int np = 0;
    unsigned grid = 0;
    double *mem = nullptr;      // (np + 2) vectors: Qh columns, base, scratch
    hipStream_t stream = nullptr;
};
template <int NT>
void launch_dense_n(esq_ctx *c, const DenseArgs &a, int np, double scale) {
    hipLaunchKernelGGL(k_dense_q<NT>, dim3(c->grid_stream), dim3(kBlock), 0,
                       c->stream, a, np, scale, c->len_pad / 2);
}
extern "C" {
int esq_dense_create(esq_ctx *c, const double *P, int rows, int p, double h,
                     int from_end, esq_dense **out) {
    if (!c || !P || !out) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (rows < 1 || rows > c->n_rows || p < 1 || p > kMaxCols)
        return fail(c, ESQ_EINVAL, "bad interpolant shape (%d, %d)", rows, p);
    esq_dense *d = new (std::nothrow) esq_dense();
    if (!d) return ESQ_ENOMEM;
    d->device = c->device;
    d->len = c->len;
    d->len_pad = c->len_pad;
    d->np = p;
    d->grid = c->grid_stream;
    hipError_t e = hipMalloc(&d->mem, (size_t)(p + 2) * d->len_pad * sizeof(double));
    if (e != hipSuccess) {
        delete d;
        return fail(c, (int)e, "hipMalloc for the interpolant failed: %s",
                    hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipFree(d->mem); delete d; return fail(c, (int)e, "stream"); }
    DenseArgs a;
    int nt = 0;
    for (int j = 0; j < rows; ++j) {
        bool any = false;
        for (int k = 0; k < p; ++k) any = any || P[(size_t)j * p + k] != 0.0;
        if (!any) continue;
        if (nt >= kMaxTerms) { esq_dense_destroy(d); return fail(c, ESQ_EINVAL, "too many rows"); }
        a.p[nt] = c->krow[c->kmap_last[j]];
        for (int k = 0; k < kMaxCols; ++k) a.w[nt][k] = k < p ? P[(size_t)j * p + k] : 0.0;
        ++nt;
    }
    for (int j = nt; j < kMaxTerms; ++j) {
        a.p[j] = nullptr;
        for (int k = 0; k < kMaxCols; ++k) a.w[j][k] = 0.0;
    }
    for (int k = 0; k < kMaxCols; ++k) a.q[k] = k < p ? d->mem + (size_t)k * d->len_pad : nullptr;
    if (nt < 1) { esq_dense_destroy(d); return fail(c, ESQ_EINVAL, "P is all zero"); }
    switch (nt) {
#define CASE(N) case N: launch_dense_n<N>(c, a, p, h); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
        CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17)
        CASE(18) CASE(19) CASE(20)
#undef CASE
    }
    // base state: after esq_rk_accept, Y is the new state, YNEW the pre-step one
    const double *base = from_end ? c->y : c->ynew;
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
    if (d->mem) (void)hipFree(d->mem);
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
int esq_rkc_stages(esq_ctx *c, int yn, int fn, int w0, int w1, int w2,
                   double hmus1, int m, const double *scalars, int *y_row_out) {
    if (!c || !y_row_out || m < 1 || (m > 1 && !scalars)) return ESQ_EINVAL;
    ENTER(c);
    // rotation instead of the reference's two full copies per stage:
    //   jm1 = first-stage result, jm2 = yn; every stage writes into a free row
    int r = esq_rkc_first_stage(c, w0, yn, fn, hmus1);
    if (r) return r;
    int jm1 = w0, jm2 = yn, free_a = w1, free_b = w2;
    int ycur = w0;
    for (int j = 2; j <= m; ++j) {
        const double *sc = scalars + 5 * (size_t)(j - 2);
        bool done = false;
        if (c->rhs_rkc) {
            // ONE sweep: derivative of yjm1 and the recursion, no fy in memory
            double *d = ROW(c, free_a), *a = ROW(c, jm1), *b = ROW(c, jm2),
                   *y0 = ROW(c, yn), *g = ROW(c, fn);
            if (!d || !a || !b || !y0 || !g) return fail(c, ESQ_EINVAL, "bad row");
            const double omn = (1.0 - sc[0]) - sc[1];
            Prof p(c, ESQ_PROF_RKC, "rhs_rkc", -1, 64.0 * (double)c->len, false,
                   40.0 * (double)c->len);
            r = c->rhs_rkc(c->rhs_user, sc[4], a, b, y0, g, sc[0], sc[1], omn,
                           sc[2], sc[3], d, c->len, (void *)c->stream,
                           (void *)p.start(), (void *)p.stop());
            if (r == 0) done = true;
            else if (r != ESQ_ENOTSUP)
                return fail(c, ESQ_ERHS, "RKC plugin entry returned %d", r);
            else p.cancel();
        }
        if (!done) {
            // fy = rhs(t_stage, yjm1) into free_a, combination overwrites free_a
            r = esq_rkc_eval_rhs(c, free_a, sc[4], jm1);
            if (r) return r;
            r = esq_rkc_stage(c, free_a, free_a, jm1, jm2, yn, fn, sc[0], sc[1],
                              sc[2], sc[3]);
            if (r) return r;
        }
        ycur = free_a;
        // shift: jm2 <- jm1, jm1 <- new; the old jm2 row becomes free
        const int old_jm2 = jm2;
        jm2 = jm1;
        jm1 = ycur;
        if (old_jm2 == yn) {      // yn is never recycled
            free_a = free_b;
        } else {
            free_a = old_jm2;
        }
    }
    *y_row_out = ycur;
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

int esq_aux_rows(esq_ctx *c, int count, int *first_id) {
    if (!c || !first_id || count < 1 || count > 32) return ESQ_EINVAL;
    ENTER(c);
    double *mem = nullptr;
    const size_t bytes = (size_t)count * c->stride * sizeof(double);
    HIPCHK(c, hipMalloc(&mem, bytes));
    HIPCHK(c, hipMemsetAsync(mem, 0, bytes, c->stream));
    c->aux_slabs.push_back(mem);
    *first_id = c->n_rows;
    for (int r = 0; r < count; ++r) {
        c->krow.push_back(mem + (size_t)r * c->stride);
        c->kmap.push_back(c->n_rows + r);
        c->kmap_last.push_back(c->n_rows + r);
    }
    c->n_rows += count;
    return 0;
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

// ---- lock-step ------------------------------------------------------------------
int esq_set_comm(esq_ctx *c, void *nccl_comm) {
    if (!c) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (nccl_comm && rccl_load() != 0) return fail(c, ESQ_ESTATE, "cannot load librccl");
    c->comm = nccl_comm;
    return 0;
}
int esq_comm_unique_id(void *id128_out) {
    if (!id128_out) return ESQ_EINVAL;
    StdoutToStderr guard;
    if (rccl_load() != 0) return ESQ_ESTATE;
    int r = g_rccl.GetUniqueId(id128_out);
    return r ? 1000 + r : 0;
}
int esq_comm_init_rank(void **comm_out, int nranks, const void *id128, int rank,
                       int device) {
    if (!comm_out || !id128) return ESQ_EINVAL;
    StdoutToStderr guard;
    if (rccl_load() != 0) return ESQ_ESTATE;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    UniqueId id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    int r = ((init_rank_fn)g_rccl.CommInitRank)(comm_out, nranks, id, rank);
    return r ? 1000 + r : 0;
}
int esq_comm_count(void *comm, int *nranks_out) {
    if (!comm || !nranks_out) return ESQ_EINVAL;
    if (rccl_load() != 0 || !g_rccl.CommCount) return ESQ_ESTATE;
    int r = g_rccl.CommCount(comm, nranks_out);
    return r ? 1000 + r : 0;
}
int esq_comm_abort(void *comm) {
    if (!comm) return 0;
    StdoutToStderr guard;
    if (rccl_load() != 0 || !g_rccl.CommAbort) return ESQ_ESTATE;
    int r = g_rccl.CommAbort(comm);
    return r ? 1000 + r : 0;
}
// all-reduce of a few host scalars over the context's communicator (identity
// without one): the lock-step mode's rank-local scalars that feed h or the stage
// count (spectral-radius estimates, debug cross-checks)
int esq_allreduce_scalars(esq_ctx *c, double *inout, int count, int op) {
    if (!c || !inout || count < 1 || count > 4) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (!c->comm) return 0;
    const int nccl_op = op == ESQ_OP_SUM ? kNcclSum : op == ESQ_OP_MIN ? kNcclMin
                      : op == ESQ_OP_MAX ? kNcclMax : -1;
    if (nccl_op < 0) return fail(c, ESQ_EINVAL, "bad reduction op %d", op);
    double *d = c->d_result + 4;
    HIPCHK(c, hipMemcpyAsync(d, inout, count * sizeof(double), hipMemcpyHostToDevice,
                             c->stream));
    int r = g_rccl.AllReduce(d, d, (size_t)count, kNcclFloat64, nccl_op, c->comm,
                             c->stream);
    if (r != 0)
        return fail(c, 1000 + r, "ncclAllReduce failed: %s",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    HIPCHK(c, hipMemcpyAsync(inout, d, count * sizeof(double), hipMemcpyDeviceToHost,
                             c->stream));
    // bounded wait, like finish_reduction
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady)
            return fail(c, (int)q, "all-reduce failed: %s", hipGetErrorString(q));
        const double el = std::chrono::duration<double>(
            std::chrono::steady_clock::now() - t0).count();
        if (el > c->comm_timeout_s) {
            if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
            c->comm = nullptr;
            return fail(c, ESQ_ETIMEOUT, "lock-step all-reduce timed out after %.0f s; "
                        "communicator aborted", c->comm_timeout_s);
        }
    }
}
int esq_comm_destroy(void *comm) {
    if (!comm) return 0;
    StdoutToStderr guard;
    if (rccl_load() != 0) return ESQ_ESTATE;
    int r = g_rccl.CommDestroy(comm);
    return r ? 1000 + r : 0;
}

// ---- measurement ----------------------------------------------------------------
int esq_profile_enable(esq_ctx *c, int class_mask) {
    if (!c) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (!class_mask) prof_drain(c);
    c->prof_mask = (unsigned)class_mask;
    return 0;
}
int esq_profile_sampling(esq_ctx *c, int every) {
    if (!c || every < 1) return ESQ_EINVAL;
    ENTER_KEEP(c);
    c->prof_every = (unsigned)every;
    return 0;
}
int esq_profile_read(esq_ctx *c, int klass, double *total_ms, long *launches,
                     double *bytes) {
    if (!c || klass < 0 || klass >= ESQ_PROF_NCLASS) return ESQ_EINVAL;
    ENTER_KEEP(c);
    prof_drain(c);
    if (total_ms) *total_ms = c->prof_ms[klass];
    if (launches) *launches = c->prof_cnt[klass];
    if (bytes) *bytes = c->prof_bytes[klass];
    return 0;
}
int esq_profile_read_moved(esq_ctx *c, int klass, double *moved_bytes) {
    if (!c || !moved_bytes || klass < 0 || klass >= ESQ_PROF_NCLASS) return ESQ_EINVAL;
    prof_drain(c);
    *moved_bytes = c->prof_moved[klass];
    return 0;
}
int esq_profile_kernels(esq_ctx *c, char *buf, size_t buflen) {
    if (!c || !buf || buflen < 2) return ESQ_EINVAL;
    ENTER_KEEP(c);
    prof_drain(c);
    size_t used = 0;
    buf[0] = 0;
    for (const auto &kv : c->prof_kernels) {
        const ProfKernel &k = kv.second;
        const int w = snprintf(buf + used, buflen - used, "%s\t%d\t%ld\t%.9g\t%.17g\t%.17g\n",
                               kv.first.c_str(), k.klass, k.launches, k.ms, k.bytes,
                               k.moved);
        if (w < 0 || (size_t)w >= buflen - used)
            return fail(c, ESQ_EINVAL, "profile table needs a larger buffer");
        used += (size_t)w;
    }
    return 0;
}
int esq_profile_reset(esq_ctx *c) {
    if (!c) return ESQ_EINVAL;
    ENTER_KEEP(c);
    prof_drain(c);
    for (int k = 0; k < ESQ_PROF_NCLASS; ++k) {
        c->prof_ms[k] = 0; c->prof_cnt[k] = 0; c->prof_bytes[k] = 0;
        c->prof_moved[k] = 0;
        c->prof_seen[k] = 0;
    }
    c->prof_kernels.clear();
    return 0;
}

}  // extern "C"
