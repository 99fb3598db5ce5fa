// esq_comm.hip -- lock-step batched integration (BASELINE.json configs[4]):
// RCCL is loaded lazily so that single-GPU use never pays for it; the data path
// crosses the links with ONE fp64 all-reduce per error evaluation
// (finish_reduction, esq_core.hip) plus a few host scalars per step where a
// rank-local scalar feeds h or the stage count (esq_allreduce_scalars).
#include <dlfcn.h>
#include <unistd.h>

#include <chrono>

#include "esq_internal.hpp"

namespace esqi {

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


void abort_comm(esq_ctx *c) {
    if (c->comm && g_rccl.CommAbort) {
        StdoutToStderr guard;
        (void)g_rccl.CommAbort(c->comm);
    }
    if (c->comm) c->comm_aborted = true;
    c->comm = nullptr;
}

}  // namespace esqi

using namespace esqi;

extern "C" {

// ---- lock-step ------------------------------------------------------------------
int esq_set_comm(esq_ctx *c, void *nccl_comm) {
    if (!c) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (nccl_comm && rccl_load() != 0) return fail(c, ESQ_ESTATE, "cannot load librccl");
    c->comm = nccl_comm;
    c->comm_aborted = false;
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
// count (spectral-radius estimates, norms of the power iteration in host-reducer
// free mode, debug cross-checks).  No copy engine is involved: the inputs go
// through the pinned slot (a kernel reads them), the all-reduce runs on the
// context's stream, a second kernel writes the results back to the slot behind a
// sequence number, and the host waits for that number with the same bounded
// spin as the error-norm reduction -- a dead peer gives ESQ_ETIMEOUT, not a hang
// (a pageable hipMemcpyAsync behind the collective would block in the runtime).
int esq_allreduce_scalars(esq_ctx *c, double *inout, int count, int op) {
    if (!c || !inout || count < 1 || count > kSlotScalars) return ESQ_EINVAL;
    ENTER_KEEP(c);
    if (!c->comm) return 0;
    const int nccl_op = op == ESQ_OP_SUM ? kNcclSum : op == ESQ_OP_MIN ? kNcclMin
                      : op == ESQ_OP_MAX ? kNcclMax : -1;
    if (nccl_op < 0) return fail(c, ESQ_EINVAL, "bad reduction op %d", op);
    // the previous user of the slot has been waited for (every reduction ends
    // with wait_slot), so the host may write the inputs now
    for (int i = 0; i < count; ++i) c->h_slot->vals[i] = inout[i];
    __atomic_thread_fence(__ATOMIC_RELEASE);
    double *d = c->d_result + 4;
    launch_load_scalars(c, d, count);
    HIPCHK(c, hipGetLastError());
    int r = g_rccl.AllReduce(d, d, (size_t)count, kNcclFloat64, nccl_op, c->comm,
                             c->stream);
    if (r != 0)
        return fail(c, 1000 + r, "ncclAllReduce failed: %s",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    const unsigned long long seq = ++c->red_seq;
    launch_publish_scalars(c, d, count, seq);
    HIPCHK(c, hipGetLastError());
    const int w = wait_slot(c, seq, c->comm_timeout_s);
    if (w == ESQ_ETIMEOUT) {
        abort_comm(c);
        return fail(c, ESQ_ETIMEOUT, "lock-step all-reduce timed out after %.0f s; "
                    "communicator aborted", c->comm_timeout_s);
    }
    if (w) return w;
    for (int i = 0; i < count; ++i) inout[i] = c->h_slot->vals[i];
    return 0;
}
int esq_comm_is_aborted(const esq_ctx *c) { return c && c->comm_aborted ? 1 : 0; }
int esq_comm_destroy(void *comm) {
    if (!comm) return 0;
    StdoutToStderr guard;
    if (rccl_load() != 0) return ESQ_ESTATE;
    int r = g_rccl.CommDestroy(comm);
    return r ? 1000 + r : 0;
}

}  // extern "C"
