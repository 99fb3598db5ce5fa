// esq_internal.hpp -- what the translation units of libextensisq_amd.so share:
// the context (one device, one HIP stream, ONE HBM slab holding every vector of
// the step, all 4-KiB aligned and padded to a multiple of 512 doubles so that
// kernels run without tail code), error / profiling helpers and the launchers
// that live in another unit.
//
//   esq_core.hip     lifecycle, data movement, reductions -> pinned slot, profiling
//   esq_step.hip     tableau + blocked-accumulation plan, the explicit RK step
//   esq_lincomb.hip  k_lincomb<NT, policy> launchers        (126 instantiations)
//   esq_reduce.hip   solution/error, error-norm, pre-error and block kernels
//   esq_aux.hip      dense output, RKC, vector plumbing, starting step
//   esq_comm.hip     RCCL (loaded lazily), lock-step scalars
//   esq_rhs*.hip     built-in device RHS plugins
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/extensisq_amd.h"
#include "esq_kernels.hpp"
#include "esq_options.hpp"

namespace esqi {

using namespace esq;

constexpr size_t kPadDoubles = 512;   // 4 KiB
constexpr int kFixedSlots = 5;        // Y, YNEW, YSTAGE, ATOL, WORK
constexpr int kPartialsCap = 1 << 17; // one partial per workgroup of a sweep
constexpr int kSlotScalars = 4;       // esq_allreduce_scalars carries <= 4 doubles

struct ProfEvent {
    hipEvent_t start, stop;
    int klass;
    double bytes;      // algorithmic bytes (SURVEY.md §8d definition)
    double moved;      // bytes the launch is designed to move
    double floor;      // ... without the halo points tiles read twice: every vector
                       // read once, every output written once (< 0: same as moved)
    char name[40];     // kernel label for the per-kernel table (esq_profile_kernels)
};
struct ProfKernel {    // per-label totals since the last reset
    int klass = 0;
    long launches = 0;
    double ms = 0.0, bytes = 0.0, moved = 0.0, floor = 0.0;
};
// pinned host slot a reduction's result lands in: the value(s), then the
// sequence number of the reduction (system-scope release), polled by the host
constexpr int kPreSlots = 4;          // early estimates in flight: a ring (one step ahead)
struct HostSlot {
    double value;
    unsigned long long seq;
    double vals[kSlotScalars];     // lock-step scalars (esq_allreduce_scalars)
    // the early error estimates of whole-step attempts (esq_rk_set_pre): nobody waits
    // for them -- the attempt's final reduction is what the host waits for -- and the
    // NEXT attempt's may land before this one's has been read (launch_ahead)
    struct { double value; unsigned long long seq; } pre[kPreSlots];
};

// one non-zero entry of a coefficient row
struct Term {
    int col;
    double c;
};

// ---- the step as a program (esq_step.hip: build_plan / run_plan) --------------
// One entry = one launch (or library kernel pair) of esq_rk_stages.  A plan is
// built ONCE per (stage range, what the step starts from) by asking the plugin's
// entries side-effect-free queries (esq_chain.dry_run / esq_epilogue.dry_run) and
// replayed every step.
enum PlanOp : unsigned char {
    OP_RHS_K0,        // K[0] = f(t, y): the end-point derivative the last accept deferred
    OP_CHAIN,         // `depth` stages from stage i in one marching sweep (i == 0: from K[0])
    OP_SRC_STAGE,     // stage 1's sweep forms its input from y and K[0] (ESQ_FUSE_SRC)
    OP_ACCUM,         // argument of stage i: blocked accumulation at a boundary / k_lincomb
    OP_LINCOMB,       // argument of stage i after a block sweep left only the partial sums
    OP_STAGE_SWEEP,   // RHS sweep of stage i + argument of stage i + 1
    OP_BLOCK_SWEEP,   // RHS sweep of stage i + blocked accumulation at boundary i + 1
    OP_YNEW_SWEEP,    // FSAL: RHS sweep of the last stage + y_new
    OP_SOLERR_SWEEP,  // others: RHS sweep of the last stage + y_new + error partial sums
    OP_RHS,           // plain RHS launch of stage i
    OP_PRE_KERNEL     // the early estimate by the library's own pass (k_pre_error) where no
                      // chain sweep carries it; its sum is published, not waited for
};
struct PlanStep {
    unsigned char op;
    signed char i, depth, what;       // what: 0 next argument, 1 y_new (FSAL), 2 y_new + error,
                                      // 3 the early estimate (+ next argument where it is
                                      // y_pre), 4 FSAL: y_new, the end-point stage, error
    bool lazy, from_rows, skip_out;   // chain forms
    float reads, writes;              // designed words per element (halo re-reads not counted)
    float amp = 0.0f;                 // chain: read amplification the plugin reported (0: none)
};
struct Plan {
    std::vector<PlanStep> steps;
    bool ynew_ready = false, solerr_ready = false;   // formed by the last launch
};

// What one launch of a step's program may change besides device memory -- and,
// therefore, what launch_ahead (esq_step.hip) saves and puts back around the launch
// it runs on the NEXT step's behalf.  One POD, copied by assignment: a field added
// here is saved with the rest (tests/test_step_plans.py runs the speculation on a
// detached context and compares the sub-struct with its copy).
struct StepState {
    double *y = nullptr, *ynew = nullptr, *ystage = nullptr, *work = nullptr;
    bool ynew_ready = false;     // YNEW already formed by the last stage's sweep
    bool solerr_ready = false;   // ... and the error partial sums too
    int red_count = 0;           // partials written by the last reducing sweep
    // rows of K that only the solution/error epilogue of their own (last) chain
    // sweep reads are not written by a step; every other reader restores them
    // first (esqi::restore_rows)
    bool tail_missing = false;            // logical rows `missing_rows` (bit i: K_i)
    bool tail_accepted = false;           // ... of the step in flight / just accepted
    unsigned long long missing_rows = 0;
    double tail_t = 0.0, tail_h = 0.0;    // that step's (t, h)
    // non-FSAL pairs: f(t_new, y_new) of an accepted step is not evaluated by
    // esq_rk_accept but as stage 0 of the NEXT step's first chain sweep; whoever
    // reads logical row 0 earlier has it evaluated first (esqi::restore_rows)
    bool k0_missing = false;
    double k0_t = 0.0;
    long end_fused = 0, end_plain = 0;    // how the end-point evaluations ran
    // sequence number of the early estimate of the attempt in flight (0: none)
    unsigned long long pre_last_seq = 0;
};

}  // namespace esqi

struct esq_ctx : esqi::StepState {
    int device = 0;
    size_t n = 0;          // state dimension as the user counts it
    size_t len = 0;        // doubles per vector (n or 2n)
    size_t len_pad = 0;    // padded doubles per vector
    size_t stride = 0;     // doubles between consecutive vectors in the slab
    int n_rows = 0;
    bool cplx = false;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // esq_snapshot_* of small vectors (large ones:
                                           // the process's download stream, esq_core.hip)
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
    std::vector<std::pair<double *, size_t>> aux_slabs;  // lazily added work rows (esq_aux_rows): block, bytes
    std::vector<double *> krow;       // physical K rows
    std::vector<int> kmap;            // logical -> physical (step in flight)
    std::vector<int> kmap_last;       // mapping of the step just accepted
    double *atolv = nullptr;          // (y, ynew, ystage, work: StepState)
    double *partials = nullptr;       // kPartialsCap doubles (own kernels use
                                      // <= kMaxPartials, fused sweeps their grid)
    double *partials2 = nullptr;      // second set (min reductions)
    double *d_result = nullptr;       // 8 doubles (device)
    esqi::HostSlot *h_slot = nullptr; // pinned, device-visible host memory
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
    bool src_pays = false;                  // working set inside the Infinity Cache
    esq_rhs_rkc_fn rhs_rkc = nullptr;       // optional RHS + Chebyshev recursion entry
    esq_rhs_rkc_chain_fn rhs_rkc_chain = nullptr;   // optional multi-stage RKC sweep
    int rkc_depth = 1;                      // stages per RKC launch (1: one each)
    bool rkc_first = false;                 // the chain entry forms y_1 itself (FIRST)
    bool rkc_first_refused = false;
    bool rkc_last = false;                  // ... and ends the step itself (LAST)
    unsigned rkc_last_refused = 0;          // bit d: declined with d stages before the end
    unsigned rkc_refused = 0;               // bit D: the plugin declined depth D
    esq_rhs_chain_fn rhs_chain = nullptr;   // optional multi-stage marching sweep
    int chain_depth = 4;                    // ESQ_CHAIN_DEPTH: 1 off, up to 4 stages per sweep
    int chain_caps = 0;                     // ESQ_CHAIN_CAP_*: what the chain entry handles
    // the step programs of this context by key (stage range, ready, k0_missing,
    // lazy rows), and the launches a plugin WITHOUT the query capability has refused
    // at run time (signature of the plan step); both dropped when the tableau or a
    // plugin entry changes (esqi::drop_plans)
    std::map<unsigned, esqi::Plan> plans;
    std::set<unsigned long long> refused;
    // early error estimate of BS5 / CFMR7osc (esq_rk_set_pre): weights over K[0..rows),
    // tested after stage rows - 1; rows == 0: none
    struct Pre {
        int rows = 0;
        std::vector<double> e, b;
        bool b_is_next = false;           // b == A[rows, :rows]: y_pre IS stage `rows`' argument
        unsigned long long seq = 0;       // estimates published so far
        long fused = 0, plain = 0;        // inside a chain sweep / by k_pre_error
    } pre;
    esq::Options opts;                    // this context's switches (esq_create2)
    bool plan_debug = false;              // PLAN_DEBUG: the planner's queries on stderr
    bool block_acc = true;                // BLOCK_ACC=0: no blocked accumulation
    bool detached = false;                // no device behind the context (esq_plan_describe)
    bool chain_from_rows = true;          // CHAIN_FROM_ROWS=0: never
    unsigned chain_ld_nt[3] = {4, 4, 4};  // CHAIN_LDNT: forced load policy of the
    bool chain_ld_nt_set = false;         // first / middle / last chain (tuning)
    // first stage argument of the NEXT step, formed at accept time
    bool pre_valid = false;
    double pre_h = 0.0;
    // THE NEXT STEP'S FIRST LAUNCH, AHEAD OF TIME (esqi::launch_ahead): where a
    // whole step's program starts with a chain sweep, that sweep is enqueued behind
    // the error norm of the step in flight -- before the host has even seen the
    // norm, if the caller could name the next step size (a run at max_step), else
    // when the step is accepted -- so the GPU does not idle while the result
    // travels to the host, the controller runs and the next launch is enqueued.
    // The sweep's K rows go to SPARE physical rows (the rows of the step in flight
    // stay readable: dense output, solver.K); accepting the step with that step
    // size swaps them in (`ahead.committed`), anything else drops them.
    bool ahead_on = true;                 // ESQ_LAUNCH_AHEAD=0: never
    std::vector<int> spare_rows;          // physical rows nobody's map points at
    double *spare_vec = nullptr;          // ... and a spare state vector: a chain that
                                          // ends in y_new writes the NEXT step's there
    double ahead_ask_t = 0.0, ahead_ask_h = 0.0;   // request of esq_rk_solution_error_ahead
    struct Ahead {
        bool valid = false;               // launched for (t, h), not yet accepted
        bool committed = false;           // accepted: esq_rk_stages(1, s, t, h) skips it
        double t = 0.0, h = 0.0;
        unsigned key = 0;                 // plan it is the first entry of ...
        bool k0_next = false;             // ... built with f(t, y) still to be evaluated
        std::vector<int> kmap;            // the next step's row map (spares swapped in)
        std::vector<int> spares;          // the spare rows once it is accepted
        double *ystage = nullptr, *work = nullptr;
        bool wrote_ynew = false;          // its last target was y_new (into spare_vec)
        bool tail_missing = false;        // rows the sweep left unwritten (lazy rows)
        unsigned long long missing_rows = 0;
        bool k0_done = false;             // it evaluated f(t, y) of the new state
        int red_count = 0;                // partials of a reducing sweep (the whole step in
                                          // one launch: Ts5's chain through the error norm)
        unsigned long long pre_seq = 0;   // its early estimate (0: none)
    } ahead;
    long ahead_used = 0, ahead_dropped = 0;
    // the opening chain sweep of the next Chebyshev step, launched behind this step's final
    // sum (esq_rkc_guess_next)
    struct RkcAhead {
        bool valid = false;               // launched, not yet taken up
        int yn = 0, fn = 0, out = 0, outp = 0, d = 0, m = 0;
        double hmus1 = 0.0;
        double sc[5 * ESQ_RKC_CHAIN_MAX_DEPTH];
        // the request (cleared by the esq_rkc_stages_end that follows it)
        bool ask = false;
        bool armed = false;               // ... and the step's end chain has named the rows
        int ask_m = 0;
        double ask_hmus1 = 0.0;
        double ask_sc[5 * ESQ_RKC_CHAIN_MAX_DEPTH];
        // where the step being enqueued leaves y_{n+1}, f(y_{n+1}) and two free work rows
        int at_y = 0, at_f = 0, free_a = 0, free_b = 0;
    } rkc_ahead;
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
    // per stage: the non-zero entries of its row of A at columns >= stage_from
    // (what a stage kernel still has to add), and the row's total non-zero count
    std::vector<std::vector<esqi::Term>> stage_terms;
    std::vector<int> stage_nnz;
    // rows left unwritten by their sweep (StepState::tail_missing): ESQ_LAZY_ROWS=0
    // never; a reader that asks twice within a few steps makes them kept
    bool lazy_rows = true;
    bool keep_rows = false;               // sticky: a reader asked twice in a row
    long accepted_steps = 0, last_restore_at = -100, restores = 0;
    // the end-point derivative as stage 0 of the next step (StepState::k0_missing):
    // ESQ_LAZY_END=0: evaluated at accept time
    bool lazy_end = true;
    // launch geometry
    unsigned grid_stream = 0;         // grid for streaming kernels
    unsigned grid_reduce = 0;
    unsigned grid_block = 0;          // grid of the (write-heavy) block kernel
    int stage_policy = 0;             // cache policy of k_lincomb (tuning knob)
    // lock-step
    void *comm = nullptr;
    bool comm_aborted = false;        // the library called ncclCommAbort itself
    // profiling
    unsigned prof_mask = 0;           // bit k: time launches of class k
    unsigned prof_every = 1;          // time every prof_every-th launch of a class
    unsigned long prof_seen[ESQ_PROF_NCLASS] = {0};
    std::vector<esqi::ProfEvent> prof_live;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[ESQ_PROF_NCLASS] = {0};
    long prof_cnt[ESQ_PROF_NCLASS] = {0};
    double prof_bytes[ESQ_PROF_NCLASS] = {0};
    double prof_moved[ESQ_PROF_NCLASS] = {0};
    std::map<std::string, esqi::ProfKernel> prof_kernels;
    char err[512] = {0};
};

namespace esqi {

int fail(esq_ctx *c, int code, const char *fmt, ...);
#define HIPCHK(c, call)                                                         \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess)                                                   \
            return esqi::fail((c), (int)e_, "%s failed: %s (%s:%d)", #call,     \
                              hipGetErrorString(e_), __FILE__, __LINE__);       \
    } while (0)

double *slot_ptr(esq_ctx *c, int slot, int row, bool logical = true);

// One process per GPU is the intended use, but a process MAY hold contexts on
// several devices and drive a context from any thread: hipSetDevice is
// per-thread state and costs well under a microsecond, so every entry point
// selects the context's device unconditionally.
// The first stage argument formed ahead of time by esq_rk_accept lives in
// YSTAGE until the next esq_rk_stages: any entry point that may write a vector
// drops it (ENTER); the read-only ones keep it (ENTER_KEEP).
// (a detached context -- esq_plan_describe, the host-side dry runs -- has no device:
// selecting "device -1" would leave an error behind on the calling thread)
#define ENTER_KEEP(c)                                        \
    do {                                                     \
        if (!(c)->detached) (void)hipSetDevice((c)->device); \
    } while (0)
// before anything but the step itself reads rows of K
#define ENSURE_ROWS(c)                                   \
    do {                                                 \
        if ((c)->tail_missing || (c)->k0_missing) {      \
            const int rr_ = esqi::restore_rows(c);       \
            if (rr_) return rr_;                         \
        }                                                \
    } while (0)
#define ENTER(c)                             \
    do {                                     \
        ENTER_KEEP(c);                       \
        (c)->pre_valid = false;              \
        (c)->idle = false;                   \
        (c)->self_valid = false;             \
        (c)->ahead.valid = false;            \
        (c)->ahead.committed = false;        \
        (c)->rkc_ahead.valid = false;        \
    } while (0)

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
         bool record_now = false, double moved = -1.0);
    void cancel();             // the launch did not happen: return the events
    hipEvent_t start() const { return on && !recorded ? ev.start : nullptr; }
    hipEvent_t stop() const { return on && !recorded ? ev.stop : nullptr; }
    ~Prof();
};
void prof_drain(esq_ctx *c);

// ---- esq_core.hip ------------------------------------------------------------
// (the process environment is read in ONE place: esq::env_get, esq_options.hpp)
using esq::env_get;
int wait_slot(esq_ctx *c, unsigned long long seq, double timeout_s);
// partials -> one double on the host (all-reduced over the communicator if set)
int finish_reduction(esq_ctx *c, double *out, bool take_min = false,
                     const double *partials = nullptr, int count = -1);
int host_wait(esq_ctx *c, bool already_idle);
int d2h(esq_ctx *c, void *host, const void *dev, size_t bytes, bool was_idle = false);
// device memory of contexts and interpolants: handed back to a small per-process cache
// instead of hipFree (esq_core.hip: why), taken from it when the size matches
hipError_t dev_acquire(int device, void **ptr, size_t bytes);
void dev_release(int device, void *ptr, size_t bytes);
int h2d(esq_ctx *c, void *dev, const void *host, size_t bytes, bool was_idle);
int call_rhs(esq_ctx *c, double t, const double *src, double *dst);
int build_row_terms(esq_ctx *c, const double *coef, int count, Terms &tm,
                    const std::vector<int> &map);
int build_row_terms2(esq_ctx *c, const double *b, int nb, const double *e, int ne,
                     Terms2 &tm, const std::vector<int> &map);
// ---- esq_step.hip --------------------------------------------------------------
// re-evaluate the rows of K the last chain sweep did not write (lazy_rows)
int restore_rows(esq_ctx *c);
// forget the step programs (the tableau, a plugin entry or a tuning knob changed)
void drop_plans(esq_ctx *c);
// the next step's first launch behind the reduction just enqueued (finish_reduction
// calls it before it waits, if esq_rk_solution_error_ahead asked for it)
// -> whether a launch went into the queue
bool launch_ahead_if_asked(esq_ctx *c);
// ---- esq_aux.hip: the opening chain sweep of the next Chebyshev step (esq_rkc_guess_next),
// called by launch_ahead_if_asked -> whether a launch went into the queue
bool rkc_launch_ahead_if_asked(esq_ctx *c);
// ---- esq_core.hip ------------------------------------------------------------
// sink of the next reduction / completion signal (bumps red_seq)
ResultSink next_sink(esq_ctx *c, bool to_host_value);
// lock-step: scalars in h_slot->vals -> device, and back behind a sequence number
void launch_load_scalars(esq_ctx *c, double *dev, int count);
void launch_publish_scalars(esq_ctx *c, const double *dev, int count,
                            unsigned long long seq);

// ---- esq_lincomb.hip ---------------------------------------------------------
int launch_lincomb(esq_ctx *c, double *out, const double *base, const Terms &tm,
                   int nt, double h, const Prof *p = nullptr,
                   const double *init = nullptr);

// ---- esq_reduce.hip ----------------------------------------------------------
int launch_solerr(esq_ctx *c, const Terms2 &tm, int nt, double h, const Prof &p);
int launch_errnorm(esq_ctx *c, const Terms &tm, int nt, double h, const Prof &p);
int launch_preerr(esq_ctx *c, const Terms2 &tm, int nt, double h, const Prof &p);
// ---- esq_core.hip: the early estimate's sum -> its slot of the ring (all-reduced over
// the communicator if set); nobody waits.  -> c->pre_last_seq
int publish_pre(esq_ctx *c, const double *partials, int count);
int launch_block(esq_ctx *c, const BlockArgs &a, int nt, int no, const Prof &p);

// ---- esq_comm.hip ------------------------------------------------------------
// RCCL, loaded lazily so that single-GPU use never pays for it
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
extern Rccl g_rccl;
int rccl_load();
constexpr int kNcclFloat64 = 8;   // ncclDouble
constexpr int kNcclSum = 0;       // ncclSum
constexpr int kNcclMax = 2;       // ncclMax
constexpr int kNcclMin = 3;       // ncclMin
// a lock-step collective timed out: abort the communicator (this rank, and
// through RCCL the others, then fail instead of hanging) and remember that the
// handle is gone, so that the host does not abort / destroy it a second time
void abort_comm(esq_ctx *c);

}  // namespace esqi

// internal (not in the public header): rebuild the blocked-accumulation plan
extern "C" int esq_replan(esq_ctx *c);
