// esq_core.hip -- context lifecycle, data movement, the reduction tail (partials
// -> pinned host slot, all-reduced over the lock-step communicator if one is
// set), RHS binding and the profiling API of libextensisq_amd.so
// (include/extensisq_amd.h).  Rotation of K rows and the y <-> y_new exchange
// are pointer swaps on the host; kernels receive row pointers and coefficients
// by value in their kernel arguments.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <mutex>

#include "esq_internal.hpp"
#include "esq_chain.hpp"

namespace esq {

// final deterministic sum of the per-block partials (one block of 1024):
// thread-strided partial sums, wave64 tree, 16 waves through LDS
__global__ __launch_bounds__(1024) void k_final_sum(
    const double *__restrict__ partials, int count, ResultSink rs) {
    __shared__ double lds[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 1024) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += lds[w];
        publish(rs, t);
    }
}
// few partials (the sweeps of small grids, where a step is bound by its launches:
// config 2's whole-step chain has 504): ONE wave, lane-strided sums and the wave64 tree --
// no LDS, no workgroup barrier, a quarter of the 1024-thread kernel's time.  The order
// of the sum is fixed (lane l takes partials l, l + 64, ...), as the big kernel's is.
__global__ __launch_bounds__(64) void k_final_sum_small(
    const double *__restrict__ partials, int count, ResultSink rs) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 64) s += partials[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) publish(rs, s);
}
constexpr int kSmallSumMax = 2048;

// after the lock-step all-reduce: device double -> host slot
__global__ void k_publish(const double *__restrict__ src, ResultSink rs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        rs.dev = nullptr;
        publish(rs, *src);
    }
}

__global__ __launch_bounds__(1024) void k_final_min(
    const double *__restrict__ partials, int count, ResultSink rs) {
    __shared__ double lds[16];
    double m = INFINITY;
    for (int i = threadIdx.x; i < count; i += 1024) m = fmin(m, partials[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) t = fmin(t, lds[w]);
        publish(rs, t);
    }
}


// lock-step scalars: pinned host doubles -> device (input of the all-reduce) ...
__global__ void k_load_scalars(const double *__restrict__ host_vals,
                               double *__restrict__ dev, int count) {
    if (blockIdx.x == 0 && (int)threadIdx.x < count)
        dev[threadIdx.x] = __hip_atomic_load(host_vals + threadIdx.x, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and back: the reduced values, then -- behind a system-scope release -- the
// sequence number the host is polling
__global__ void k_publish_scalars(const double *__restrict__ dev,
                                  double *__restrict__ host_vals, int count,
                                  unsigned long long *host_seq,
                                  unsigned long long seq) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int i = 0; i < count; ++i)
        __hip_atomic_store(host_vals + i, dev[i], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace esq

namespace esqi {

int fail(esq_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

double *slot_ptr(esq_ctx *c, int slot, int row, bool logical) {
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
Prof::Prof(esq_ctx *ctx, int klass, const char *name, int nt, double bytes,
           bool record_now, double moved)
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
    ev.floor = -1.0;
    if (nt >= 0) snprintf(ev.name, sizeof(ev.name), "%s<%d>", name, nt);
    else snprintf(ev.name, sizeof(ev.name), "%s", name);
    if (recorded) (void)hipEventRecord(ev.start, c->stream);
}
void Prof::cancel() {
    if (!on) return;
    c->prof_pool.push_back(ev.start);
    c->prof_pool.push_back(ev.stop);
    on = false;
}
Prof::~Prof() {
    if (!on) return;
    if (recorded) (void)hipEventRecord(ev.stop, c->stream);
    c->prof_live.push_back(ev);
}

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
            pk.floor += ev.floor < 0.0 ? ev.moved : ev.floor;
        }
        c->prof_pool.push_back(ev.start);
        c->prof_pool.push_back(ev.stop);
    }
    c->prof_live.clear();
}

// Wait until the reduction numbered `seq` has landed in the pinned host slot.
// The GPU writes the slot itself (no copy engine, no stream-sync wake-up: the
// 8-byte D2H copy + hipStreamSynchronize pair cost ~15 us of every step), the
// host spins on it.  The stream is queried so that a faulted kernel surfaces as an
// error instead of a hang -- but only once the wait has lasted 2 ms, then every
// millisecond: a hipStreamQuery makes the runtime put a marker behind the last
// command of the stream (here: the next step's first sweep, launched ahead), and that
// marker's system-scope release holds the following sweep back by 5.5-6 us while the
// XCDs' L2s write back (seen in the kernel trace of every step while the query came
// every 4096 spins, i.e. 20-40 us: profiles/r06_experiments.md section 16).
// `timeout_s` > 0 bounds the wait (lock-step: a peer that died never arrives at the
// all-reduce).
int wait_slot(esq_ctx *c, unsigned long long seq, double timeout_s) {
    volatile unsigned long long *flag = &c->h_slot->seq;
    const auto t0 = std::chrono::steady_clock::now();
    double next_query_s = 2e-3;
    for (unsigned long spins = 1;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spins & 0x3ff) == 0) {
            const double el = std::chrono::duration<double>(
                std::chrono::steady_clock::now() - t0).count();
            if (el < next_query_s) continue;
            next_query_s = el + 1e-3;
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                // everything on the stream has finished: the slot is written
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
                return fail(c, ESQ_ESTATE, "reduction %llu finished without a result", seq);
            }
            if (q != hipErrorNotReady)
                return fail(c, (int)q, "stream failed while waiting for a reduction: %s",
                            hipGetErrorString(q));
            if (timeout_s > 0.0 && el > timeout_s) return ESQ_ETIMEOUT;
        }
    }
}

ResultSink next_sink(esq_ctx *c, bool to_host_value) {
    ResultSink rs;
    rs.seq = ++c->red_seq;
    rs.dev = nullptr;
    rs.host_value = to_host_value ? &c->h_slot->value : nullptr;
    rs.host_seq = &c->h_slot->seq;
    return rs;
}
void launch_load_scalars(esq_ctx *c, double *dev, int count) {
    hipLaunchKernelGGL(k_load_scalars, dim3(1), dim3(64), 0, c->stream,
                       c->h_slot->vals, dev, count);
}
void launch_publish_scalars(esq_ctx *c, const double *dev, int count,
                            unsigned long long seq) {
    hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(64), 0, c->stream, dev,
                       c->h_slot->vals, count, &c->h_slot->seq, seq);
}

// partials -> one double on the host (all-reduced over the communicator if set)
int finish_reduction(esq_ctx *c, double *out, bool take_min, const double *partials,
                     int count) {
    if (c->detached) {             // host-side dry run: nothing was enqueued
        launch_ahead_if_asked(c);
        if (out) *out = 0.0;
        return 0;
    }
    if (!partials) partials = c->partials;
    if (count < 0) count = (int)c->grid_reduce;
    ResultSink rs = next_sink(c, !c->comm);
    rs.dev = c->comm ? c->d_result : nullptr;
    if (take_min)
        hipLaunchKernelGGL(k_final_min, dim3(1), dim3(1024), 0, c->stream,
                           partials, count, rs);
    else if (count <= kSmallSumMax)
        hipLaunchKernelGGL(k_final_sum_small, dim3(1), dim3(64), 0, c->stream,
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
    // the next step's first launch goes in behind the reduction, BEFORE the host
    // waits for it (only if the caller asked: esq_rk_solution_error_ahead)
    const bool went_ahead = launch_ahead_if_asked(c);
    int w = wait_slot(c, rs.seq, c->comm ? c->comm_timeout_s : 0.0);
    if (w == ESQ_ETIMEOUT) {
        // a peer never reached the collective
        abort_comm(c);
        return fail(c, ESQ_ETIMEOUT, "lock-step all-reduce did not complete within "
                    "%.0f s (a peer rank failed?); communicator aborted",
                    c->comm_timeout_s);
    }
    if (w) return w;
    // idle: the final-sum kernel was the last thing enqueued -- unless the next
    // step's first sweep went in behind it
    c->idle = !went_ahead;
    if (out) *out = c->h_slot->value;
    return 0;
}

// the early estimate of a whole-step attempt: partials -> pre[seq % kPreSlots] of the
// pinned slot, behind a sequence number; the host picks it up after the attempt's
// final reduction (esq_rk_pre_result)
int publish_pre(esq_ctx *c, const double *partials, int count) {
    const unsigned long long seq = ++c->pre.seq;
    c->pre_last_seq = seq;
    if (c->detached) return 0;
    auto &slot = c->h_slot->pre[seq % kPreSlots];
    ResultSink rs;
    rs.seq = seq;
    rs.host_seq = &slot.seq;
    rs.host_value = c->comm ? nullptr : &slot.value;
    double *dev = c->d_result + 8 + (seq % kPreSlots);
    rs.dev = c->comm ? dev : nullptr;
    if (count <= kSmallSumMax)
        hipLaunchKernelGGL(k_final_sum_small, dim3(1), dim3(64), 0, c->stream, partials, count,
                           rs);
    else
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(1024), 0, c->stream, partials, count, rs);
    HIPCHK(c, hipGetLastError());
    if (c->comm) {
        const int r = g_rccl.AllReduce(dev, dev, 1, kNcclFloat64, kNcclSum, c->comm, c->stream);
        if (r != 0)
            return fail(c, 1000 + r, "ncclAllReduce failed: %s",
                        g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        rs.host_value = &slot.value;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, dev, rs);
        HIPCHK(c, hipGetLastError());
    }
    return 0;
}

int call_rhs(esq_ctx *c, double t, const double *src, double *dst) {
    if (!c->rhs) return fail(c, ESQ_ESTATE, "no device RHS set (esq_set_rhs)");
    if (c->detached) return 0;                  // host-side dry run
    Prof p(c, ESQ_PROF_RHS, "rhs_plugin", -1, 16.0 * (double)c->len, /*record_now=*/true);
    c->self_valid = false;     // a plugin kernel does not signal its own completion
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

// ---- host-slab mode ------------------------------------------------------------
static bool in_host_slab(const esq_ctx *c, const void *dev) {
    const double *p = (const double *)dev;
    return c->host_slab && p >= c->slab && p < c->slab + c->slab_doubles;
}
static double *host_of(const esq_ctx *c, const void *dev) {
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
    const ResultSink rs = next_sink(c, true);
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
// ---- device memory is kept, not freed ------------------------------------------
// Device memory that has been FREED ONCE -- by this process, at times by one before it
// on the same device -- and is allocated again is slow to read for the DMA engines: 76 MiB to
// the host take 2.8 ms from it, 1.41 ms from memory handed out for the first time, at
// any offset, for as long as the allocation lives (tools/alloc_probe.py: three fresh
// 1.2 GB buffers 1.41 each; two freed, two new ones in their place 2.8; a fresh 8 GB
// one 1.41; everything freed, then 11 GB physically contiguous: 2.8.  Kernels see no
// difference).  A process that makes one solver after the other -- consecutive
// solve_ivp calls -- ran on such memory from its second solver on whenever the new
// slab was allocated after the old one's hipFree, i.e. nearly always.  So the slabs of
// contexts and the interpolants' blocks go back to a small cache instead -- exact
// sizes, per device, oldest out when ESQ_SLAB_CACHE_MB (default: a quarter of the
// device's memory) or eight blocks are exceeded -- and the next solver of that size
// gets the same, still-fast allocation, without a hipMalloc.  A request that hipMalloc
// cannot serve empties the cache and tries again.  What the cache cannot help: the
// FIRST slab of a process on a device other processes have used (it may be slow memory,
// and then stays that solver size's memory for the life of the process).
// Every block remembers the rate of the latest engine download that read from it
// (lane_copy times them all): a request is served by a block known to be fast before
// one nobody has read from yet, and by one known to be slow only when a new hipMalloc
// -- made while the slow block is still held, so that it is other memory -- fails; a
// process whose first slab happened to be slow memory thus gets a second chance with
// every solver it makes, until a fast block is found and kept.
struct Block { int device; char *ptr; size_t bytes; double gbs; };   // gbs 0: not read yet
std::mutex g_block_mu;
std::vector<Block> g_blocks;                    // cached; oldest first
std::vector<Block> g_live;                      // handed out
double g_best_gbs = 0.0;                        // fastest engine download of the process
constexpr size_t kCacheMinBytes = (size_t)8 << 20;
constexpr size_t kCacheMaxBlocks = 8;

// (slow memory halves the rate; a download beside a running sweep reaches 0.73 of it:
// tools/numa_probe.py)
int block_rank_locked(const Block &b) {         // 2 known fast, 1 unknown, 0 known slow
    return b.gbs == 0.0 ? 1 : b.gbs >= 0.6 * g_best_gbs ? 2 : 0;
}
// cap of the bytes cached PER DEVICE: ESQ_SLAB_CACHE_MB, else a quarter of that device's
// memory (asked once per device, with that device selected)
size_t cache_cap_bytes(int device) {
    // (the switch is read per release -- they are rare: tests change it)
    const char *e = env_get("SLAB_CACHE_MB");
    if (e && *e) return (size_t)strtoull(e, nullptr, 10) << 20;
    static size_t dflt[64] = {0};
    static bool known[64] = {false};
    if (device < 0 || device >= 64) return 0;
    if (!known[device]) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(device) == hipSuccess &&
            hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            dflt[device] = total_b / 4;
        else
            (void)hipGetLastError();
        if (cur >= 0) (void)hipSetDevice(cur);
        known[device] = true;
    }
    return dflt[device];
}
// (the calling thread's current device is what it was afterwards)
void drop_cached_locked(int device_or_all) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (size_t k = 0; k < g_blocks.size();) {
        if (device_or_all < 0 || g_blocks[k].device == device_or_all) {
            (void)hipSetDevice(g_blocks[k].device);
            (void)hipFree(g_blocks[k].ptr);
            g_blocks.erase(g_blocks.begin() + (long)k);
        } else {
            ++k;
        }
    }
    if (cur >= 0) (void)hipSetDevice(cur);
}
// the best cached block of that size (newest among equals) leaves the cache; false: none
bool take_cached_locked(int device, size_t bytes, int min_rank, Block *out) {
    long pick = -1;
    int best = -1;
    for (size_t k = g_blocks.size(); k-- > 0;) {
        const Block &b = g_blocks[k];
        if (b.device != device || b.bytes != bytes) continue;
        const int r = block_rank_locked(b);
        if (r > best) { best = r; pick = (long)k; }
    }
    if (pick < 0 || best < min_rank) return false;
    *out = g_blocks[(size_t)pick];
    g_blocks.erase(g_blocks.begin() + pick);
    return true;
}
// (the current device is `device`)
hipError_t dev_acquire(int device, void **ptr, size_t bytes) {
    Block b{device, nullptr, bytes, 0.0};
    {
        std::lock_guard<std::mutex> lk(g_block_mu);
        if (take_cached_locked(device, bytes, /*min_rank=*/1, &b)) {
            g_live.push_back(b);
            *ptr = b.ptr;
            return hipSuccess;
        }
    }
    // (a cached block of that size, if there is one, is known to be slow: it stays where
    // it is while hipMalloc looks for other memory)
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(g_block_mu);
            if (take_cached_locked(device, bytes, /*min_rank=*/0, &b)) {
                g_live.push_back(b);
                *ptr = b.ptr;
                return hipSuccess;
            }
            drop_cached_locked(device);
        }
        (void)hipSetDevice(device);
        e = hipMalloc(ptr, bytes);
    }
    if (e == hipSuccess && bytes >= kCacheMinBytes) {
        std::lock_guard<std::mutex> lk(g_block_mu);
        g_live.push_back({device, (char *)*ptr, bytes, 0.0});
    }
    return e;
}
void dev_release(int device, void *ptr, size_t bytes) {
    if (!ptr) return;
    double gbs = 0.0;
    {
        std::lock_guard<std::mutex> lk(g_block_mu);
        for (size_t k = 0; k < g_live.size(); ++k)
            if (g_live[k].ptr == (char *)ptr) {
                gbs = g_live[k].gbs;
                g_live.erase(g_live.begin() + (long)k);
                break;
            }
    }
    std::lock_guard<std::mutex> lk(g_block_mu);      // (also guards cache_cap_bytes' table)
    const size_t cap = cache_cap_bytes(device);
    if (bytes >= kCacheMinBytes && bytes <= cap) {
        g_blocks.push_back({device, (char *)ptr, bytes, gbs});
        // what THIS device holds against ITS cap; at most kCacheMaxBlocks per device
        auto held_of = [&](size_t *count) {
            size_t held = 0;
            *count = 0;
            for (const auto &b : g_blocks)
                if (b.device == device) { held += b.bytes; ++*count; }
            return held;
        };
        size_t count = 0, held = held_of(&count);
        while (count > 0 && (held > cap || count > kCacheMaxBlocks)) {
            // out: a block known to be slow before any other, the oldest among equals
            long out = -1;
            for (size_t k = 0; k < g_blocks.size(); ++k) {
                if (g_blocks[k].device != device) continue;
                if (out < 0 || block_rank_locked(g_blocks[k]) < block_rank_locked(g_blocks[(size_t)out]))
                    out = (long)k;
            }
            (void)hipFree(g_blocks[(size_t)out].ptr);       // (the current device is `device`)
            g_blocks.erase(g_blocks.begin() + out);
            held = held_of(&count);
        }
        return;
    }
    (void)hipFree(ptr);
}
// an engine download of `bytes` from `dev` ran at `gbs`: the block it read from knows
void note_block_rate(int device, const void *dev, double gbs) {
    std::lock_guard<std::mutex> lk(g_block_mu);
    if (gbs > g_best_gbs) g_best_gbs = gbs;
    for (auto &b : g_live)
        if (b.device == device && (const char *)dev >= b.ptr && (const char *)dev < b.ptr + b.bytes) {
            b.gbs = gbs;
            return;
        }
}

// ---- large device-to-host copies ------------------------------------------------
// Every download of 8 MiB and more runs on ONE stream per device and process (made
// once, never destroyed), by hipMemcpyAsync -- the device's DMA engines -- into a
// page-locked destination: 56 GB/s from memory that was allocated for the first time,
// 25-30 GB/s from memory that has been freed and allocated again (above: why the
// contexts' memory is cached, not freed).  (Round 5 carried an opt-in copy KERNEL for
// the slow case here; it faulted in two of seven suite runs, was never root-caused,
// and went in round 6 -- the block cache is what keeps downloads fast.)
constexpr int kLaneDevices = 64;
constexpr size_t kLaneMinBytes = (size_t)8 << 20;
struct CopyLane {
    hipStream_t stream = nullptr;
    bool failed = false;
    double best_gbs = 0.0;         // fastest download so far
    double last_gbs = 0.0;         // the latest one
    long copies = 0;
    int busy = 0;                  // downloads in flight on the lane right now
};
std::mutex g_lane_mu;
CopyLane g_lane[kLaneDevices];

// the process's download stream of `device` (the current device); nullptr: none
hipStream_t copy_lane(int device) {
    if (device < 0 || device >= kLaneDevices) return nullptr;
    std::lock_guard<std::mutex> lk(g_lane_mu);
    CopyLane &ln = g_lane[device];
    if (!ln.stream && !ln.failed &&
        hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ln.stream = nullptr;
        ln.failed = true;
    }
    return ln.stream;
}

// One large copy on `stream` (everything it depends on has been waited for); blocks
// until the bytes are in `host`.  `pinned`: the destination is page-locked (a staged
// pageable copy says nothing about the engines: not timed)
hipError_t lane_copy(int device, hipStream_t stream, void *host, const void *dev, size_t bytes,
                     bool pinned) {
    const bool lane_ok = device >= 0 && device < kLaneDevices && bytes >= kLaneMinBytes;
    bool alone = false;
    if (lane_ok) {
        std::lock_guard<std::mutex> lk(g_lane_mu);
        alone = g_lane[device].busy++ == 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (!lane_ok) return e;
    const double gbs = (double)bytes * 1e-9 /
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
        std::lock_guard<std::mutex> lk(g_lane_mu);
        CopyLane &ln = g_lane[device];
        // (two downloads that shared the lane halved each other's rate: neither says
        // anything about the memory it read from)
        const bool last_out = --ln.busy == 0;
        alone = alone && last_out;
        if (e == hipSuccess) {
            ln.last_gbs = gbs;
            if (alone && gbs > ln.best_gbs) ln.best_gbs = gbs;
            ++ln.copies;
        }
    }
    if (e == hipSuccess && pinned && alone) note_block_rate(device, dev, gbs);
    return e;
}

int d2h(esq_ctx *c, void *host, const void *dev, size_t bytes, bool was_idle) {
    if (in_host_slab(c, dev)) {
        const int w = host_wait(c, was_idle);
        if (w) return w;
        memcpy(host, host_of(c, dev), bytes);
        return 0;
    }
    bool pinned = false;
    hipStream_t lane = nullptr;
    if (bytes >= kLaneMinBytes) {
        pinned = hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable) ==
                 hipSuccess;
        if (!pinned) (void)hipGetLastError();
        lane = copy_lane(c->device);
    }
    hipError_t e = hipSuccess;
    if (lane) {
        // behind everything the context has enqueued, on the process's download stream
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = lane_copy(c->device, lane, host, dev, bytes, pinned);
    } else {
        e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
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

}  // namespace esqi

using namespace esqi;

extern "C" {

int esq_abi_version(void) { return ESQ_ABI_VERSION; }

int esq_option_level(const char *key) { return key ? esq::option_level(key) : 0; }

int esq_device_count(int *count_out) {
    if (!count_out) return ESQ_EINVAL;
    const hipError_t e = hipGetDeviceCount(count_out);
    return e == hipSuccess ? 0 : (int)e;
}

int esq_device_pci_bus_id(int device, char *buf, size_t buflen) {
    if (!buf || buflen < 16 || device < 0) return ESQ_EINVAL;
    buf[0] = 0;
    const hipError_t e = hipDeviceGetPCIBusId(buf, (int)buflen, device);
    return e == hipSuccess ? 0 : (int)e;
}

int esq_create(esq_ctx **out, int device, size_t n, int n_rows, int is_complex) {
    return esq_create2(out, device, n, n_rows, is_complex, 0, nullptr);
}
int esq_create2(esq_ctx **out, int device, size_t n, int n_rows, int is_complex,
                int flags, const char *options) {
    if (!out || n_rows < 1 || n_rows > 64) return ESQ_EINVAL;
    esq_ctx *c = new (std::nothrow) esq_ctx();
    if (!c) return ESQ_ENOMEM;
    *out = c;   // returned even on failure so the caller can read the message
    {
        // this context's switches; a key the caller did not give falls back to the
        // process default ESQ_<KEY> (esq_options.hpp)
        std::string bad;
        if (c->opts.parse(options, esq::kOptContext, &bad) != 0)
            return fail(c, ESQ_EINVAL, "esq_create2: '%s' is not a context option "
                        "(esq_option_level)", bad.c_str());
    }
    const esq::Options &o = c->opts;
    c->device = device;
    c->n = n;
    c->cplx = is_complex != 0;
    c->len = c->cplx ? 2 * n : n;
    c->len_pad = ((c->len + kPadDoubles - 1) / kPadDoubles) * kPadDoubles;
    if (c->len_pad == 0) c->len_pad = kPadDoubles;
    c->stride = c->len_pad;
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
        HIPCHK(c, dev_acquire(c->device, (void **)&c->slab, slab_doubles * sizeof(double)));
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
    HIPCHK(c, hipHostMalloc((void **)&c->h_slot, 128,
                            hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_slot, 0, sizeof(HostSlot));
    c->comm_timeout_s = (double)o.uint_or("COMM_TIMEOUT_S", 120);
    // K_i of the stage / block sweeps is streamed out (nobody re-reads it soon).
    // The last stage's and the end-point sweep's derivative is read again by the
    // next launches: kept cacheable while three vectors fit the Infinity Cache
    // (Pr9 at n = 5e6: plain 74 us vs streamed 80 us for the following chain),
    // streamed beyond (Pr8 at n = 1e7: end-point sweep 58 -> 48 us, step -2.5 %)
    const bool three_fit = 3 * c->len_pad * sizeof(double) <= ((size_t)160 << 20);
    c->epi_nt = o.uint_or("EPI_NT", three_fit ? 0x3 : 0xf);
    c->lazy_rows = o.uint_or("LAZY_ROWS", 1) != 0;
    c->lazy_end = o.uint_or("LAZY_END", 1) != 0;
    c->ahead_on = false;           // esq_rk_set_launch_ahead: callers that take whole steps
    c->chain_from_rows = o.uint_or("CHAIN_FROM_ROWS", 1) != 0;
    c->plan_debug = o.has("PLAN_DEBUG");
    if (c->plan_debug)
        fprintf(stderr, "esq: slab at %p (%zu doubles, row stride %zu)\n", (void *)c->slab,
                c->slab_doubles, c->stride);
    c->block_acc = o.uint_or("BLOCK_ACC", 1) != 0;
    if (const char *e = o.get("CHAIN_LDNT")) {          // "first,middle,last" bit masks
        unsigned a = 4, b = 4, d = 4;
        if (sscanf(e, "%u,%u,%u", &a, &b, &d) == 3) {
            c->chain_ld_nt[0] = a; c->chain_ld_nt[1] = b; c->chain_ld_nt[2] = d;
            c->chain_ld_nt_set = true;
        }
    }
    c->chain_depth = (int)o.uint_or("CHAIN_DEPTH", 4);     // 5 and 6 exist too
    if (c->chain_depth > ESQ_CHAIN_MAX_DEPTH) c->chain_depth = ESQ_CHAIN_MAX_DEPTH;
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
    c->src_pays = o.uint_or("SRC", fits_mall ? 1 : 0) != 0;
    const unsigned per_cu = o.uint_or("BLOCKS_PER_CU", fits_mall ? 8 : 2);
    c->stage_policy = (int)o.uint_or("STAGE_POLICY", fits_mall ? 0 : 10);
    size_t g = (size_t)cus * per_cu;
    if (g > need) g = need;
    if (g < 1) g = 1;
    c->grid_stream = (unsigned)g;
    c->grid_reduce = (unsigned)(g > (size_t)kMaxPartials ? kMaxPartials : g);
    size_t gb = (size_t)cus * 16;
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
    // (a snapshot copy of a small vector may still read the slab: before it goes back
    // to the cache -- where the next context could take and overwrite it)
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    if (c->host_slab) {
        if (c->slab_host) (void)hipHostFree(c->slab_host);
    } else if (c->slab) {
        dev_release(c->device, c->slab, c->slab_doubles * sizeof(double));
    }
    for (const auto &a : c->aux_slabs) dev_release(c->device, a.first, a.second);
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
    // a later restore must not undo the upload -- and must not see it: rows that are
    // evaluated on demand (esq_rk_lazy_rows) are functions of the state as the step
    // left it (K[0] = f(t, y) of the accepted state: the reference's `self.f` stays
    // what it was when `solver.y` is assigned, common.py:298), so they are
    // evaluated BEFORE the state changes
    if (slot == ESQ_SLOT_K || slot == ESQ_SLOT_Y || slot == ESQ_SLOT_YNEW) ENSURE_ROWS(c);
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
    if (slot == ESQ_SLOT_K) ENSURE_ROWS(c);
    double *d = slot_ptr(c, slot, row);
    if (!d) return fail(c, ESQ_EINVAL, "bad slot/row %d/%d", slot, row);
    const size_t cnt = (slot == ESQ_SLOT_ATOL) ? c->n : c->len;
    return d2h(c, host, d, cnt * sizeof(double), c->idle);
}
// ---- a vector on its way to the host while the solver goes on stepping --------
// (the per-step `solver.y` of plain solve_ivp, scipy ivp.py:665, 702: 80 MB at
// n = 1e7, 1.4 ms over PCIe against a 0.47 ms step)
struct esq_snapshot {
    int device;
    hipEvent_t ready;        // recorded on the context's stream: the vector is final
    hipStream_t stream;      // the process's download stream of the device (copy_lane)
    const double *src;
    size_t bytes;
};
int esq_snapshot_begin(esq_ctx *c, int slot, int row, void **token_out) {
    if (!c || !token_out) return ESQ_EINVAL;
    *token_out = nullptr;
    ENTER_KEEP(c);
    if (c->host_slab || c->detached) return ESQ_ENOTSUP;
    if (slot == ESQ_SLOT_K) ENSURE_ROWS(c);
    const double *d = slot_ptr(c, slot, row);
    if (!d) return fail(c, ESQ_EINVAL, "bad slot/row %d/%d", slot, row);
    const size_t bytes = ((slot == ESQ_SLOT_ATOL) ? c->n : c->len) * sizeof(double);
    hipStream_t lane = bytes >= kLaneMinBytes ? copy_lane(c->device) : nullptr;
    if (!lane) {
        // (small vectors, or no lane to be had: a stream of the context's own)
        if (!c->copy_stream)
            HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        lane = c->copy_stream;
    }
    esq_snapshot *tk = new esq_snapshot{c->device, nullptr, lane, d, bytes};
    hipError_t e = hipEventCreateWithFlags(&tk->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(tk->ready, c->stream);
    if (e != hipSuccess) {
        if (tk->ready) (void)hipEventDestroy(tk->ready);
        delete tk;
        return fail(c, (int)e, "snapshot event: %s", hipGetErrorString(e));
    }
    *token_out = tk;
    return 0;
}
int esq_release_cached_memory(size_t *bytes_out) {
    std::lock_guard<std::mutex> lk(g_block_mu);
    size_t held = 0;
    for (const auto &b : g_blocks) held += b.bytes;
    drop_cached_locked(-1);
    if (bytes_out) *bytes_out = held;
    return 0;
}
int esq_copy_lane_info(int device, double *best_gbs_out, double *last_gbs_out,
                       long *copies_out) {
    if (device < 0 || device >= kLaneDevices) return ESQ_EINVAL;
    std::lock_guard<std::mutex> lk(g_lane_mu);
    const CopyLane &ln = g_lane[device];
    if (best_gbs_out) *best_gbs_out = ln.best_gbs;
    if (last_gbs_out) *last_gbs_out = ln.last_gbs;
    if (copies_out) *copies_out = ln.copies;
    return 0;
}
int esq_host_pin(void *host, size_t bytes) {
    if (!host || bytes == 0) return ESQ_EINVAL;
    const hipError_t e = hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int esq_host_unpin(void *host) {
    if (!host) return ESQ_EINVAL;
    const hipError_t e = hipHostUnregister(host);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int esq_snapshot_copy(void *token, double *host, int host_is_pinned) {
    esq_snapshot *tk = (esq_snapshot *)token;
    if (!tk) return ESQ_EINVAL;
    int rc = 0;
    hipError_t e = hipSetDevice(tk->device);
    if (e == hipSuccess && host) {
        // large copies run into a pinned destination (as d2h's do: the staged
        // pageable path is half as fast): pinned ahead of time by the caller
        // (esq_host_pin, 0.2 ms for 80 MB of resident pages) or here; unpinned here
        bool pinned = host_is_pinned != 0;
        if (!pinned && tk->bytes >= kLaneMinBytes) {
            pinned = hipHostRegister(host, tk->bytes, hipHostRegisterMapped | hipHostRegisterPortable) ==
                     hipSuccess;
            if (!pinned) (void)hipGetLastError();
        }
        static const bool dbg = env_get("SNAPSHOT_DEBUG") != nullptr;
        const auto t_a = std::chrono::steady_clock::now();
        e = hipEventSynchronize(tk->ready);              // (the copy itself is timed)
        const auto t_b = std::chrono::steady_clock::now();
        if (e == hipSuccess)
            e = lane_copy(tk->device, tk->stream, host, tk->src, tk->bytes, pinned);
        const auto t_c = std::chrono::steady_clock::now();
        if (pinned) (void)hipHostUnregister(host);
        if (dbg)
            fprintf(stderr, "[esq_snapshot_copy] %zu B, caller-pinned %d, pinned %d: wait %.3f ms, "
                    "copy %.3f ms, unpin %.3f ms\n", tk->bytes, host_is_pinned, (int)pinned,
                    std::chrono::duration<double, std::milli>(t_b - t_a).count(),
                    std::chrono::duration<double, std::milli>(t_c - t_b).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count());
    } else if (host_is_pinned && host) {
        (void)hipHostUnregister(host);
    }
    if (e != hipSuccess) rc = (int)e;
    (void)hipEventDestroy(tk->ready);
    delete tk;
    return rc;
}

int esq_copy(esq_ctx *c, int dst_slot, int dst_row, int src_slot, int src_row) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    if (dst_slot == ESQ_SLOT_K || src_slot == ESQ_SLOT_K || dst_slot == ESQ_SLOT_Y ||
        dst_slot == ESQ_SLOT_YNEW)
        ENSURE_ROWS(c);
    double *d = slot_ptr(c, dst_slot, dst_row), *s = slot_ptr(c, src_slot, src_row);
    if (!d || !s) return fail(c, ESQ_EINVAL, "bad slot/row");
    HIPCHK(c, hipMemcpyAsync(d, s, c->len_pad * sizeof(double),
                             hipMemcpyDefault, c->stream));
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
    drop_plans(c);
    c->rhs_rkc = nullptr;
    c->rhs_rkc_chain = nullptr;
    c->rkc_depth = 1;
    c->rkc_refused = 0;
    const bool had_chain = c->rhs_chain != nullptr;
    c->rhs_chain = nullptr;
    // the block plan was made for a chaining plugin: make it again for this one
    return (had_chain && c->have_tab) ? esq_replan(c) : 0;
}
int esq_set_rhs_chain(esq_ctx *c, esq_rhs_chain_fn fn, int caps) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs_chain = fn;
    c->chain_caps = fn ? caps : 0;
    // the blocked-accumulation plan depends on how the plugin sweeps
    return c->have_tab ? esq_replan(c) : 0;
}
int esq_set_rhs_rkc(esq_ctx *c, esq_rhs_rkc_fn fn) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs_rkc = fn;
    return 0;
}
int esq_set_rhs_rkc_chain(esq_ctx *c, esq_rhs_rkc_chain_fn fn, int max_depth) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs_rkc_chain = fn;
    c->rkc_refused = 0;
    c->rkc_first = fn && (max_depth & ESQ_RKC_CHAIN_FIRST) && c->opts.uint_or("RKC_FIRST", 1) != 0;
    c->rkc_first_refused = false;
    c->rkc_last = fn && (max_depth & ESQ_RKC_CHAIN_LAST) && c->opts.uint_or("RKC_LAST", 1) != 0;
    c->rkc_last_refused = 0;
    max_depth &= 0xff;
    int d = fn ? max_depth : 1;
    const int env = (int)c->opts.uint_or("RKC_DEPTH", 0);    // 0: the plugin's own
    if (env > 0 && env < d) d = env;
    if (d > ESQ_RKC_CHAIN_MAX_DEPTH) d = ESQ_RKC_CHAIN_MAX_DEPTH;
    c->rkc_depth = d < 1 ? 1 : d;
    return 0;
}
int esq_set_rhs_fused(esq_ctx *c, esq_rhs_fused_fn fn, int fuse_mask) {
    if (!c) return ESQ_EINVAL;
    ENTER(c);
    c->rhs_fused = fn;
    c->fuse_mask = fn ? fuse_mask : 0;
    drop_plans(c);
    return 0;
}

int esq_aux_rows(esq_ctx *c, int count, int *first_id) {
    if (!c || !first_id || count < 1 || count > 32) return ESQ_EINVAL;
    ENTER(c);
    double *mem = nullptr;
    const size_t bytes = (size_t)count * c->stride * sizeof(double);
    if (c->detached) {
        // no device behind this context (esq_plan_describe): distinct fake addresses
        mem = reinterpret_cast<double *>(((uintptr_t)2 << 40) +
                                         (uintptr_t)c->n_rows * c->stride * sizeof(double));
    } else {
        HIPCHK(c, dev_acquire(c->device, (void **)&mem, bytes));
        HIPCHK(c, hipMemsetAsync(mem, 0, bytes, c->stream));
        c->aux_slabs.push_back({mem, bytes});
    }
    *first_id = c->n_rows;
    for (int r = 0; r < count; ++r) {
        c->krow.push_back(mem + (size_t)r * c->stride);
        c->kmap.push_back(c->n_rows + r);
        c->kmap_last.push_back(c->n_rows + r);
    }
    c->n_rows += count;
    return 0;
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
        const int w = snprintf(buf + used, buflen - used,
                               "%s\t%d\t%ld\t%.9g\t%.17g\t%.17g\t%.17g\n",
                               kv.first.c_str(), k.klass, k.launches, k.ms, k.bytes,
                               k.moved, k.floor);
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
