/*
 * extensisq_amd.h -- C ABI of libextensisq_amd.so (gfx950 / MI355X)
 *
 * Drop-in boundary for ONE hot path of the reference package extensisq v0.6.0
 * (pure Python/NumPy): the adaptive explicit Runge-Kutta step
 *     RungeKutta._step_impl            extensisq/common.py:222-308
 *     BS5._step_impl                   extensisq/bogacki.py:238-338
 *     SSV2stab._step_impl / _stages    extensisq/sommeijer.py:162-329
 * The reference has no FFI of its own (it is NumPy all the way down); every
 * entry point below replaces the NumPy expression(s) cited next to it.  The
 * step-size controller, the accept/reject decision and all scalar recurrences
 * stay on the host (Python, extensisq_amd/common.py); one double -- the sum of
 * squares of the weighted error -- crosses back per step attempt.
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t / ncclResult_t
 *     (+1000) on a runtime failure, a negative ESQ_E* code on misuse;
 *     esq_last_error(ctx) gives the text.
 *   - all vectors are fp64 ("double"); a complex state of n elements is n
 *     interleaved (re, im) pairs and is created with is_complex = 1.
 *   - host pointers are borrowed for the duration of the call only.
 *   - a context is bound to one device and one HIP stream; calls on one context
 *     must come from one host thread at a time (the library holds no global
 *     state, so one thread per context is fine).
 *   - functions that return a host scalar synchronise the context's stream;
 *     all others only enqueue work.
 */
#ifndef EXTENSISQ_AMD_H
#define EXTENSISQ_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESQ_ABI_VERSION 9

/* error codes (negative = misuse) */
#define ESQ_EINVAL   (-1)   /* bad argument (row/slot out of range, NULL, ...) */
#define ESQ_ESTATE   (-2)   /* call out of order (no tableau / no RHS set)     */
#define ESQ_ENOMEM   (-3)   /* host allocation failed                          */
#define ESQ_ERHS     (-4)   /* the RHS plugin returned non-zero                */
#define ESQ_ENOTSUP  (-5)   /* optional plugin entry cannot handle this case   */
#define ESQ_ETIMEOUT (-6)   /* a lock-step collective did not complete in time */

/* vector slots of a context (row is ignored unless slot == ESQ_SLOT_K) */
#define ESQ_SLOT_K       0  /* stage derivatives K[row], row < n_rows          */
#define ESQ_SLOT_Y       1  /* current state y                                 */
#define ESQ_SLOT_YNEW    2  /* tentative state y_new (== y_old after accept)   */
#define ESQ_SLOT_YSTAGE  3  /* stage argument  y + h * sum_j a_ij K_j          */
#define ESQ_SLOT_ATOL    4  /* per-component absolute tolerance (real, n)      */
#define ESQ_SLOT_WORK    5  /* scratch (error vector, dense output, ...)       */

typedef struct esq_ctx esq_ctx;

/*
 * Device RHS plugin: enqueue f_dev = fun(t, y_dev) on `hip_stream` (a
 * hipStream_t) and return 0.  n counts doubles (2n' for a complex state).
 * Replaces the Python callable `fun(t, y)` of scipy's OdeSolver contract
 * (scipy/integrate/_ivp/base.py:139-166) for states that live in HBM.
 */
typedef int (*esq_rhs_fn)(void *user, double t, const double *y_dev,
                          double *f_dev, size_t n, void *hip_stream);

/*
 * OPTIONAL fused entry of a plugin: enqueue ONE sweep that evaluates
 *     f_dev = fun(t, y_in)
 * and, while the derivative is still in registers, applies a POINTWISE epilogue
 * to it -- the Runge-Kutta arithmetic that follows every RHS evaluation
 * (common.py:343-356).  The separate streaming kernel and its re-read of f_dev
 * disappear; nothing is recomputed on stencil halos because the epilogue is
 * pointwise.  `kind` selects the epilogue (device side: csrc/esq_epilogue.hpp,
 * which a plugin author includes and instantiates in the sweep kernel):
 *
 *   ESQ_EPI_STAGE    out = base + h*(init + sum_{j<nt} c[j]*rows[j] + c_self*f)
 *                    base = y, or y_in itself when y == NULL.  The argument of
 *                    the next stage (common.py:355), y_new of an FSAL pair
 *                    (:343), or -- with y == NULL -- the end-point evaluation
 *                    K[s] = fun(t+h, y_new) chained with the NEXT step's first
 *                    stage argument y_new + h_next*a_10*K[s] (:289-291, 355).
 *   ESQ_EPI_BLOCK    blocked accumulation (see esq_rk_stage_accumulate) with f
 *                    as the block's last column:
 *                    out_o[o] = init_o[o] + sum_j w[j][o]*rows[j] + w_self[o]*f;
 *                    if y != NULL output 0 is y + h*(that sum) instead.
 *   ESQ_EPI_SOLERR   y_new = y + h*(sum c[j]*rows[j] + c_self*f)  -> out
 *                    err   = h*(sum e[j]*rows[j] + e_self*f)
 *                    partials[block] = sum |err/(atol + rtol*max(|y|,|y_new|))|^2
 *                    (`_comp_sol_err`, common.py:341-351, non-FSAL pairs)
 *   ESQ_EPI_ERRNORM  FSAL pairs: y_in is y_new, f is K[s];
 *                    err = h*(sum e[j]*rows[j] + e_self*f), scale from y and
 *                    y_in, partial sums as above      (common.py:348-351)
 *
 *   ESQ_EPI_RKCERR   end of a Runge-Kutta-Chebyshev step (sommeijer.py:214-220):
 *                    y_in is the new state y, f = fun(t+h, y) is stored in f_dev
 *                    (it is the next step's f_n); rows[0] = y_n, rows[1] = f_n;
 *                    est = 0.8*(y_n - y) + 0.4*h*(f_n + f), partial sums of
 *                    |est/(atol + rtol*max(|y|,|y_n|))|^2 as above.  One sweep
 *                    instead of RHS + error kernel (3 words per element less).
 *
 * The FMA chains run over j ascending and take f last; products with h and the
 * final add are rounded separately: bit-identical to the stand-alone kernels.
 * Reducing kinds write ONE partial per workgroup to partials[blockIdx.x] and
 * report the workgroup count in *partials_used (<= partials_cap, else return
 * ESQ_ENOTSUP).  f_store_nt != 0: f_dev is not re-read soon, stream it out.
 * start/stop are hipEvent_t or NULL (dispatch timestamps for esq_profile_*).
 * Return ESQ_ENOTSUP for any case the plugin does not fuse: the library falls
 * back to esq_rhs_fn + its own kernels.
 */
#define ESQ_EPI_STAGE    1
#define ESQ_EPI_BLOCK    2
#define ESQ_EPI_SOLERR   3
#define ESQ_EPI_ERRNORM  4
#define ESQ_EPI_RKCERR   6   /* (bit 5 of the fuse mask is ESQ_FUSE_SRC) */
#define ESQ_EPI_MAX_ROWS 20
#define ESQ_EPI_MAX_OUT  12
typedef struct esq_epilogue {
    int kind;
    int nt;                                  /* K rows read from memory        */
    const double *rows[ESQ_EPI_MAX_ROWS];
    double c[ESQ_EPI_MAX_ROWS];              /* stage / solution weights       */
    double e[ESQ_EPI_MAX_ROWS];              /* error weights                  */
    double c_self, e_self;                   /* weights of the fresh f         */
    const double *init;                      /* STAGE: leading partial or NULL */
    const double *y;                         /* base state (NULL: y_in)        */
    double h;
    double *out;                             /* STAGE: argument; SOLERR: y_new */
    int no;                                  /* BLOCK: number of outputs       */
    double w[ESQ_EPI_MAX_ROWS][ESQ_EPI_MAX_OUT];
    double w_self[ESQ_EPI_MAX_OUT];
    const double *init_o[ESQ_EPI_MAX_OUT];
    double *out_o[ESQ_EPI_MAX_OUT];
    const double *atol_vec;                  /* reductions: NULL = scalar atol */
    double atol_s, rtol;
    size_t n_valid;                          /* elements beyond are padding    */
    double *partials;
    int partials_cap;
    int *partials_used;
    int f_store_nt;
    /* ON-THE-FLY INPUT (first stage of a step; requested only from plugins
     * registered with ESQ_FUSE_SRC): if in_row != NULL the sweep's input is not
     * y_in (which is then NULL) but
     *     in_base + in_h * (in_c * in_row)        rounded like ESQ_EPI_STAGE,
     * evaluated where the sweep needs it (halos included; one term, so this is
     * one more row window of ONE vector, served by L1/L2).  The first stage
     * argument is then never written to nor read from memory. */
    const double *in_base, *in_row;
    double in_c, in_h;
    /* the state is complex: one (re, im) pair per element.  Only the reducing
     * kinds differ (|.| is the complex modulus, n_valid counts complex
     * elements); a plugin for real states returns ESQ_ENOTSUP for them. */
    int is_complex;
    /* QUERY: decide exactly as for a launch -- return 0, ESQ_ENOTSUP or an error --
     * but enqueue nothing and write nothing (*partials_used included).  The library
     * builds its launch plan of a step from such queries (esq_rk_stages) instead of
     * learning from refused launches; asked only of entries registered with
     * ESQ_FUSE_QUERY.  esq::dispatch_epilogue (csrc/esq_plugin.hpp) implements it. */
    int dry_run;
} esq_epilogue;
typedef int (*esq_rhs_fused_fn)(void *user, double t, const double *y_in,
                                double *f_dev, const esq_epilogue *epi, size_t n,
                                void *hip_stream, void *start_event,
                                void *stop_event);

/*
 * OPTIONAL chain entry of a plugin: `depth` consecutive stages in ONE marching
 * sweep.  With T_0 = y_in (the argument of the chain's first stage i):
 *     K_{i+k} = fun(t[k], T_k)                          -> f_out[k]   (k < depth)
 *     T_{e+1} = y + h*(sum_u cu[e][u]*rows[u] + sum_{k<=e} ck[e][k]*K_{i+k})
 * T_1 .. T_{depth-1} (the arguments of the chain's later stages) live in
 * registers only; the last target T_depth goes to `out`:
 *   kind_last == ESQ_EPI_STAGE   the argument of stage i+depth, or y_new of an
 *                                FSAL pair (common.py:355, 343)
 *   kind_last == ESQ_EPI_SOLERR  y_new with the solution weights, and
 *       err = h*(sum_u eu[u]*rows[u] + sum_k ek[k]*K_{i+k}), partial sums of
 *       |err/(atol + rtol*max(|y|,|y_new|))|^2            (common.py:341-351)
 *       The same kind carries the EARLY error estimate of BS5 / CFMR7osc
 *       (bogacki.py:340-346, calvo.py:255-261; esq_rk_set_pre): the weights are the
 *       estimate's, the target is y_pre -- stored (out: CFMR7osc's y_pre IS the
 *       argument of the next stage) or, out == NULL, only the partner of y in the
 *       scale (ESQ_CHAIN_CAP_PRE).
 *   kind_last == ESQ_EPI_ERRNORM (ESQ_CHAIN_CAP_ERRNORM)  FSAL pairs: the chain's LAST
 *       stage is the end-point evaluation K_s = fun(t + h, y_new).  Its argument
 *       T_{depth-1} = y_new (solution weights in cu/ck[depth-2]) is handed on in
 *       registers AND stored to `out`; "target depth" is no vector but the error sum
 *       err = h*(sum_u cu[depth-1][u]*rows[u] + sum_k ck[depth-1][k]*K_{i+k}) with the
 *       error weights E in the last row of cu / ck (K_s = the chain's last derivative
 *       included), partial sums of |err/(atol + rtol*max(|y|,|y_new|))|^2.
 *       common.py:341-351 in one sweep with the stages before it.
 * The intermediate arguments never touch memory; y and the K rows the sums share
 * are read once for the whole chain.  rows[] is the union of the K rows any
 * target reads from memory, in ascending column order; bit u of umask[e] (bit k
 * of kmask[e]) says whether rows[u] (K_{i+k}) takes part in target e+1's sum --
 * the weight of one that does not is +0.0, so a plugin may skip it or multiply
 * through (fma(0, v, s) == s for finite v; a non-finite v then poisons the target as
 * NumPy's K[:i].T @ A[i,:i] over all rows does, common.py:355; the built-in sweeps
 * multiply through since ABI 8: no test per term).  A leading partial
 * sum of the blocked accumulation is passed as the FIRST row of its target with
 * weight 1 (fma(1, p, 0) == p).  Every sum runs over ascending column index
 * (rows, then the chain's own K in order), each
 * product with h and the final add rounded separately: K rows and states are
 * bit-identical to `depth` one-stage sweeps.  y == NULL: the base is y_in itself
 * (a chain that starts with the end-point evaluation of the previous step).
 * How a stencil plugin does it (stage k marches k grid rows behind stage 0):
 * csrc/esq_chain.hpp.  Return ESQ_ENOTSUP for any case the plugin does not chain.
 */
#define ESQ_CHAIN_MAX_DEPTH 6
#define ESQ_CHAIN_MAX_ROWS 10
typedef struct esq_chain {
    int depth;                               /* 2 .. ESQ_CHAIN_MAX_DEPTH        */
    int kind_last;                           /* ESQ_EPI_STAGE, _SOLERR or _ERRNORM */
    int nu;                                  /* K rows read from memory         */
    const double *rows[ESQ_CHAIN_MAX_ROWS];
    double cu[ESQ_CHAIN_MAX_DEPTH][ESQ_CHAIN_MAX_ROWS];
    double eu[ESQ_CHAIN_MAX_ROWS];
    unsigned umask[ESQ_CHAIN_MAX_DEPTH];
    double ck[ESQ_CHAIN_MAX_DEPTH][ESQ_CHAIN_MAX_DEPTH];
    double ek[ESQ_CHAIN_MAX_DEPTH];
    unsigned kmask[ESQ_CHAIN_MAX_DEPTH];
    const double *y;
    double h;
    /* from_rows != 0: T_0 is NOT read from y_in but formed by the sweep itself,
     *     T_0 = y + h*(sum_u c0[u]*rows[u])      (bit u of umask0: rows[u] takes part)
     * -- the argument of the chain's first stage from rows the chain reads anyway
     * (same ascending FMA chain, same two roundings as the sweep that would have
     * written it; y_in is then unspecified, y is not NULL).  Asked for only where
     * it adds no row to rows[]; the library then tells the PREVIOUS chain not to
     * write that argument (its `out` is NULL, kind_last == ESQ_EPI_STAGE).
     * ESQ_ENOTSUP makes the library form the argument with its own kernel. */
    int from_rows;
    double c0[ESQ_CHAIN_MAX_ROWS];
    unsigned umask0;
    double t[ESQ_CHAIN_MAX_DEPTH];
    double *f_out[ESQ_CHAIN_MAX_DEPTH];      /* NULL: do not store K_{i+k} (the library
                                              * re-evaluates the row for whoever reads
                                              * it, esq_rk_lazy_rows)             */
    double *out;                             /* ESQ_EPI_STAGE, _SOLERR: NULL = not wanted */
    int f_store_nt;
    /* cache policy of the loads, a hint (bit 0: y_in, bit 1: y, bit 8 + u: rows[u]):
     * set = nothing in this step reads the vector again (non-temporal load),
     * clear = a later sweep of the step does (keep it in the Infinity Cache) */
    int load_nt;
    const double *atol_vec;                  /* SOLERR, as in esq_epilogue      */
    double atol_s, rtol;
    size_t n_valid;
    double *partials;
    int partials_cap;
    int *partials_used;
    /* out (may be NULL): the factor by which the plugin's tile geometry
     * multiplies the bytes READ (halo rows and columns are loaded by two tiles);
     * the library books it in the launch's designed traffic (esq_profile_*) */
    double *read_amplification;
    /* QUERY (see esq_epilogue.dry_run): asked only of entries registered with
     * ESQ_CHAIN_CAP_QUERY; esq::dispatch_chain implements it. */
    int dry_run;
} esq_chain;
typedef int (*esq_rhs_chain_fn)(void *user, const double *y_in,
                                const esq_chain *chain, size_t n, void *hip_stream,
                                void *start_event, void *stop_event);

/*
 * OPTIONAL RKC entry of a plugin: enqueue ONE sweep that evaluates
 * f = fun(t, yjm1) and finishes the Chebyshev stage without storing f,
 *   y_out = mu*yjm1 + nu*yjm2 + omn*yn + hmus*(f - ajm1*fn)
 * (sommeijer.py:311-313; every product and sum rounded, left to right).
 * y_out must not alias yjm1.  Return ESQ_ENOTSUP to fall back.
 */
typedef int (*esq_rhs_rkc_fn)(void *user, double t, const double *yjm1,
                              const double *yjm2, const double *yn,
                              const double *fn, double mu, double nu, double omn,
                              double hmus, double ajm1, double *y_out, size_t n,
                              void *hip_stream, void *start_event,
                              void *stop_event);

/*
 * OPTIONAL RKC chain entry of a plugin: `depth` consecutive Chebyshev stages in
 * ONE marching sweep (the loop body of sommeijer.py:309-329, `depth` times).
 * With Y_0 = yjm1 and Y_{-1} = yjm2, for k = 0 .. depth-1
 *     Y_{k+1} = mu[k]*Y_k + nu[k]*Y_{k-1} + omn[k]*yn
 *               + hmus[k]*(fun(t[k], Y_k) - ajm1[k]*fn)
 * every product and sum rounded, left to right, exactly as esq_rhs_rkc_fn does
 * for one stage: the results are bit-identical to `depth` one-stage launches.
 * Y_1 .. Y_{depth-2} never touch memory; Y_depth goes to `out`, Y_{depth-1} to
 * `out_prev` (the two inputs of the next chain; out_prev == NULL: not wanted,
 * the step's last chain).  out and out_prev alias none of the four inputs
 * (tiles re-read their neighbours' points).  Per element and chain the memory
 * sees 4 words read + 2 written instead of 5*depth.  How a 3-D stencil plugin
 * does it (stage k marches k planes behind stage 0, a workgroup owns a patch of
 * the plane and its waves exchange their edge rows through LDS):
 * csrc/esq_rkc3d.hpp.  Return ESQ_ENOTSUP for any case the plugin does not chain
 * (the library then runs shorter chains or one launch per stage).
 */
#define ESQ_RKC_CHAIN_MAX_DEPTH 8
typedef struct esq_rkc_chain {
    int depth;                               /* 2 .. ESQ_RKC_CHAIN_MAX_DEPTH    */
    const double *yjm1, *yjm2, *yn, *fn;
    double mu[ESQ_RKC_CHAIN_MAX_DEPTH], nu[ESQ_RKC_CHAIN_MAX_DEPTH];
    double omn[ESQ_RKC_CHAIN_MAX_DEPTH];     /* (1 - mu) - nu, formed by the library */
    double hmus[ESQ_RKC_CHAIN_MAX_DEPTH], ajm1[ESQ_RKC_CHAIN_MAX_DEPTH];
    double t[ESQ_RKC_CHAIN_MAX_DEPTH];       /* t + h*theta_{j-1} of each stage */
    double *out, *out_prev;
    /* FIRST (ABI v5; asked only of entries registered with ESQ_RKC_CHAIN_FIRST): the
     * chain opens a step -- yjm1 == NULL, and the chain's first input is the first
     * Chebyshev iterate  y_1 = yn + hmus_first * fn  (sommeijer.py:289; product and sum
     * rounded separately), formed where the sweep needs it; yjm2 == yn.  No sweep
     * writes y_1, the chain reads two vectors instead of four. */
    double hmus_first;
    /* LAST (ABI v5; asked only of entries registered with ESQ_RKC_CHAIN_LAST): the
     * chain ends a step -- fy_out != NULL (and out_prev == NULL): after the `depth`
     * stages, in the same sweep,  fy_out = fun(t_end, out)  and the partial sums of the
     * error estimate exactly as ESQ_EPI_RKCERR forms them from (out, yn, fn, fy_out, h)
     * with the tolerances below: one partial per workgroup in partials[blockIdx.x],
     * the workgroup count in *partials_used (> partials_cap: return ESQ_ENOTSUP).
     * The sweep `f(t + h, y_{n+1})` + error estimate of sommeijer.py:214-220 and its
     * launch disappear; y_{n+1} is never read back. */
    double *fy_out;
    double t_end, h;
    const double *atol_vec;
    double atol_s, rtol;
    size_t n_valid;
    double *partials;
    int partials_cap;
    int *partials_used;
    /* out (may be NULL): bytes read per byte of the four inputs (halo points are
     * loaded by several tiles); booked in the launch's designed traffic */
    double *read_amplification;
} esq_rkc_chain;
typedef int (*esq_rhs_rkc_chain_fn)(void *user, const esq_rkc_chain *chain,
                                    size_t n, void *hip_stream, void *start_event,
                                    void *stop_event);

/* ---- lifecycle ---------------------------------------------------------- */
int  esq_abi_version(void);
/* devices this process can see (a launcher may restrict each rank to one) */
int  esq_device_count(int *count_out);
/* PCI address of HIP device `device` ("0000:c1:00.0", NUL-terminated) -- for a
 * launcher that pinned its CPUs by sysfs position before it touched the GPU and
 * wants to know afterwards whether that was the device it got (bench.py) */
int  esq_device_pci_bus_id(int device, char *buf, size_t buflen);
/* n: state dimension (complex elements if is_complex); n_rows: rows of K
 * (n_stages + 1, plus extra rows for BS5's interpolants / RKC work vectors).
 * Replaces `self.K = np.empty((n_stages + 1, n))`  common.py:216 and the
 * per-step temporaries of common.py:343-356. */
int  esq_create(esq_ctx **out, int device, size_t n, int n_rows, int is_complex);
/* flags: ESQ_CREATE_HOST_SLAB -- small problems whose RHS is a host callable
 * (the reference's own tests live at n <= 400): every vector of the context is
 * pinned, device-mapped HOST memory.  The kernels are the same; the per-stage
 * download of the stage argument and upload of the derivative become plain
 * memcpy calls (completion is signalled through the pinned result slot), which
 * removes two copy-engine round trips per stage.  Meant for vectors of at most
 * a few thousand doubles: kernels then read their operands over PCIe. */
#define ESQ_CREATE_HOST_SLAB 1
/* ... and THIS context's tuning switches (since ABI 8): `options` is
 * "key=value;key=value" (keys in any case, with or without the ESQ_ prefix; NULL or ""
 * for none), e.g. "chain_depth=1;lazy_rows=0".  A key the caller does not give takes the
 * process default -- the environment variable ESQ_<KEY>, read now, on this thread -- so
 * one solver's switches never reach another, and nothing writes the environment.  An
 * unknown key, or one that steers a plugin object (esq_rhs_set_options), is refused:
 * ESQ_EINVAL, the key in esq_last_error.  esq_option_level: 1 = a context's switch,
 * 2 = a plugin object's, 0 = not a switch of the library. */
int  esq_create2(esq_ctx **out, int device, size_t n, int n_rows, int is_complex,
                 int flags, const char *options);
int  esq_option_level(const char *key);
int  esq_destroy(esq_ctx *ctx);
const char *esq_last_error(const esq_ctx *ctx);
int  esq_synchronize(esq_ctx *ctx);
/* number of doubles one vector holds on the device (n or 2n) */
size_t esq_vector_len(const esq_ctx *ctx);

/* ---- data movement ------------------------------------------------------ */
/* `K[i] = f` (common.py:240), `y0` upload, and the lazy host mirror of
 * solver.y / solver.K.  Synchronous.  K rows are addressed LOGICALLY (row 0 is
 * always the first stage of the step in flight; esq_rk_accept rotates). */
int  esq_upload(esq_ctx *ctx, int slot, int row, const double *host);
int  esq_download(esq_ctx *ctx, int slot, int row, double *host);
/* A download that runs BESIDE the steps that follow (plain solve_ivp keeps every
 * accepted `solver.y`, scipy ivp.py:665, 702: 80 MB per step at n = 1e7, three times
 * the step's own time).  esq_snapshot_begin (the thread that drives the context)
 * marks the point in the context's stream at which the vector is final and returns
 * a token; esq_snapshot_copy (ANY thread, typically a copy worker; blocks until the
 * data is in `host`; touches nothing of the context) waits for that point on the
 * process's download stream of the device (esq_copy_lane_info below) and copies.  The caller guarantees that nothing
 * overwrites the vector before the copy has returned -- a state vector of an
 * explicit pair is next written two steps after it was formed (Python side:
 * extensisq_amd/lazy.py).  host == NULL: give the token back without copying.
 * Copies of 8 MB and more run into a PINNED destination: host_is_pinned != 0 says the
 * caller has pinned `host` already (esq_host_pin: a helper thread does it while the
 * step runs), otherwise the call pins it; either way it is unpinned before the call
 * returns.  Not for host-slab contexts (ESQ_ENOTSUP). */
int  esq_snapshot_begin(esq_ctx *ctx, int slot, int row, void **token_out);
int  esq_snapshot_copy(void *token, double *host, int host_is_pinned);
/* The device memory of destroyed contexts and interpolants (blocks of 8 MiB and more) is
 * kept in a small per-process cache instead of being freed -- memory that hipMalloc hands
 * out a second time is read by the DMA engines at half the rate of memory allocated for the
 * first time (csrc/esq_core.hip), and the next solver of the same size starts without a
 * hipMalloc.  At most eight blocks and ESQ_SLAB_CACHE_MB (default: a quarter of the device's
 * memory; 0 = no cache); a request the device cannot serve empties it first.  This frees
 * everything cached now and says how many bytes that was. */
int  esq_release_cached_memory(size_t *bytes_out);
/* Downloads of 8 MiB and more (esq_download, esq_snapshot_copy) run on ONE stream per
 * device and process, through the device's DMA engines (hipMemcpyAsync) into a page-locked
 * destination.  This reports the record: the fastest and the latest download (GB/s; the
 * fastest among those that had the stream to themselves) and how many there were.
 * (The opt-in copy kernel of ABI 7 -- ESQ_D2H_MODE -- is gone: it faulted.) */
int  esq_copy_lane_info(int device, double *best_gbs_out, double *last_gbs_out,
                        long *copies_out);
/* page-lock / release a host buffer (hipHostRegister: mapped into the device's address
 * space, portable across devices) */
int  esq_host_pin(void *host, size_t bytes);
int  esq_host_unpin(void *host);
/* device-to-device copy between two (slot,row) vectors, asynchronous */
int  esq_copy(esq_ctx *ctx, int dst_slot, int dst_row, int src_slot, int src_row);

/* ---- method description ------------------------------------------------- */
/* Butcher tableau: A is s*s row-major (strictly lower triangular), B[s], C[s],
 * E[s+1]; fsal != 0 iff E[s] != 0 (common.py:217).  Zero coefficients are
 * skipped by the kernels (only 0*Inf NaN propagation differs from NumPy).
 * Mirrors the class attributes A, B, C, E of common.py:97-107. */
int  esq_rk_set_tableau(esq_ctx *ctx, int s, const double *A, const double *B,
                        const double *C, const double *E, int fsal);
/* rtol scalar, atol scalar (n_atol == 1) or per-component (n_atol == n);
 * values as returned by validate_tol  common.py:30-54. */
int  esq_set_tol(esq_ctx *ctx, double rtol, const double *atol, size_t n_atol);
int  esq_set_rhs(esq_ctx *ctx, esq_rhs_fn fn, void *user);
/* register (or clear) the optional fused entry: every RHS evaluation of a step
 * then is ONE kernel that also does the Runge-Kutta arithmetic which follows it
 * (next stage argument / blocked accumulation / solution + error norm).
 * fuse_mask selects the epilogue kinds the library may request (bit k =
 * ESQ_EPI_* value k; ESQ_FUSE_ALL for all) -- each one is bit-identical to the
 * unfused sequence and can be switched off for A/B tests. */
#define ESQ_FUSE_ALL 0x5e
#define ESQ_FUSE_SRC 0x20    /* the entry also accepts the on-the-fly input    */
#define ESQ_FUSE_QUERY 0x80  /* the entry honours esq_epilogue.dry_run         */
int  esq_set_rhs_fused(esq_ctx *ctx, esq_rhs_fused_fn fn, int fuse_mask);
/* register (or clear) the optional chain entry: esq_rk_stages then runs up to
 * ESQ_CHAIN_DEPTH (default 4) stages per launch wherever plain stage sweeps (and
 * the solution/error sweep) follow each other.  Needs the fused entry too (the
 * remaining single stages). */
/* `caps`: the optional forms of a chain the entry handles (anything else in an
 * esq_chain it may still decline with ESQ_ENOTSUP, case by case); the library asks
 * only for forms the plugin has declared:
 *   FROM_STATE  chain->y == NULL (the base of the sums is y_in: a chain whose first
 *               stage is the end-point derivative the previous accept left to it)
 *   SKIP_ROWS   chain->f_out[k] == NULL (derivatives nothing later reads)
 *   FROM_ROWS   chain->from_rows (the chain forms its own input)
 *   SKIP_OUT    chain->out == NULL with ESQ_EPI_STAGE (the next chain forms its input)
 * 0 = plain chains only: every row written, every input read from y_in. */
#define ESQ_CHAIN_CAP_FROM_STATE 1
#define ESQ_CHAIN_CAP_SKIP_ROWS  2
#define ESQ_CHAIN_CAP_FROM_ROWS  4
#define ESQ_CHAIN_CAP_SKIP_OUT   8
#define ESQ_CHAIN_CAP_ALL        15
/*   QUERY       the entry honours chain->dry_run (side-effect-free "would you take
 *               this chain?"): the library then plans a step from the answers; without
 *               it a refused launch is remembered and the step finished the plain way */
#define ESQ_CHAIN_CAP_QUERY      16
/*   PRE         ESQ_EPI_SOLERR with chain->out == NULL (an early error estimate whose
 *               y_pre nothing reads)
 *   ERRNORM     kind_last == ESQ_EPI_ERRNORM (the FSAL end-point stage and the error
 *               norm inside the chain) */
#define ESQ_CHAIN_CAP_PRE        32
#define ESQ_CHAIN_CAP_ERRNORM    64
int  esq_set_rhs_chain(esq_ctx *ctx, esq_rhs_chain_fn fn, int caps);
/* register (or clear) the optional RKC entry: esq_rkc_stages then issues ONE
 * kernel per Chebyshev stage (RHS + recursion) instead of two */
int  esq_set_rhs_rkc(esq_ctx *ctx, esq_rhs_rkc_fn fn);
/* register (or clear) the optional RKC chain entry: esq_rkc_stages then runs up
 * to max_depth (<= ESQ_RKC_CHAIN_MAX_DEPTH; ESQ_RKC_DEPTH in the environment
 * lowers it, 1 = one launch per stage) stages per launch when it is given four
 * work rows.  Needs the one-stage RKC entry too (remainders, refused chains). */
#define ESQ_RKC_CHAIN_FIRST 0x100   /* or-ed into max_depth: the entry takes the FIRST form */
#define ESQ_RKC_CHAIN_LAST  0x200   /* ... and the LAST form                              */
int  esq_set_rhs_rkc_chain(esq_ctx *ctx, esq_rhs_rkc_chain_fn fn, int max_depth);

/* ---- explicit RK launches ----------------------------------------------- */
/* YSTAGE = Y + h * sum_j A[i][j] * K[j]           common.py:355 (`dy`, `y+dy`)
 * Blocked accumulation (ESQ_BLOCK_ACC, default on): at column boundaries chosen
 * from the tableau's sparsity, one extra pass forms the leading part of the sums
 * of ALL later stages, which then resume the same FMA chain -- bit-identical
 * results, each K row of a block read once instead of once per later stage.
 * Stages must therefore be requested in ascending order within an attempt. */
int  esq_rk_stage_accumulate(esq_ctx *ctx, int i, double h);
/* the blocked-accumulation plan of the current tableau: returns the number of
 * column boundaries (>= 0; negative on misuse), writes them and the 8-byte
 * words per element and step moved by the stage kernels without / with it */
int  esq_rk_block_plan(esq_ctx *ctx, int *boundaries, int max_boundaries,
                       int *words_plain, int *words_blocked);
/* K[dst_row] = rhs(t, <src vector>)                common.py:356, 348, 291    */
int  esq_rk_eval_rhs(esq_ctx *ctx, int dst_row, double t, int src_slot,
                     int src_row);
/* for i in [i_from, i_to): stage_accumulate(i, h); K[i] = rhs(t + C[i]*h, YSTAGE)
 *                                                  common.py:241-242, 353-356
 * With a fused plugin entry (esq_set_rhs_fused) the RHS sweep of stage i also
 * forms stage i+1's argument or runs the blocked accumulation at a column
 * boundary; when i_to == s the last sweep forms YNEW (FSAL tableaux) or YNEW
 * and the error partial sums (others), which esq_rk_solution_error then does
 * not recompute.  K rows and states are bit-identical to the
 * one-kernel-per-operation sequence; the error norm agrees to rounding (its
 * partial sums are grouped by the sweep's workgroups). */
int  esq_rk_stages(esq_ctx *ctx, int i_from, int i_to, double t, double h);
/* YNEW = Y + h * sum_j B[j] K[j]                   common.py:343              */
int  esq_rk_solution(esq_ctx *ctx, double h);
/* sum over elements of |h * sum_j E[j] K[j] / scale|^2 with
 * scale = atol + rtol * max(|Y|, |YNEW|)           common.py:335-339, 57-66
 * (host finishes with sqrt(sumsq / n)).  Synchronises. */
int  esq_rk_error_norm(esq_ctx *ctx, double h, double *sumsq_out);
/* Whole `_comp_sol_err` (common.py:341-351) with a device RHS: solution, the
 * FSAL evaluation K[s] = rhs(t + h, YNEW) if the tableau is FSAL, error norm.
 * Non-FSAL tableaux use ONE fused pass (solution + scale + error).
 * Synchronises. */
int  esq_rk_solution_error(esq_ctx *ctx, double t, double h, double *sumsq_out);
/* The same, and -- h_next != 0: the step size the caller will use next IF this
 * attempt is accepted (a run at max_step knows it before it knows the error norm)
 * -- the NEXT step's first launch goes into the queue right behind the error norm,
 * before the host waits for it, where that step's program starts with a chain
 * sweep.  The sweep writes spare rows: the K rows of the attempt stay readable.
 * esq_rk_accept(t + h, ., h_next) with exactly that step size makes it real (the
 * next esq_rk_stages(1, s, t + h, h_next) skips its first launch); a rejected
 * attempt, another step size or any call that may write a vector drops it.  The
 * GPU then does not idle while the norm travels to the host, the controller runs
 * and the first launch is enqueued (~17 us per step).  Needs esq_rk_set_launch_ahead. */
int  esq_rk_solution_error_ahead(esq_ctx *ctx, double t, double h, double h_next,
                                 double *sumsq_out);
/* A whole attempt of a step in ONE call: esq_rk_stages(ctx, 1, s, t, h) followed by
 * esq_rk_solution_error_ahead(ctx, t, h, h_next, sumsq_out) (h_next == 0: no launch ahead)
 * -- and, pre_sumsq_out != NULL with an early estimate registered, esq_rk_pre_result.  On
 * small grids the host's share bounds the step; this saves two trips through the binding. */
int  esq_rk_attempt(esq_ctx *ctx, double t, double h, double h_next, double *sumsq_out,
                    double *pre_sumsq_out);
/* BS5's early estimate (bogacki.py:340-346) over K[0..rows): YSTAGE-free,
 * y_pre = Y + h*sum b_scale_pre[j] K[j] is formed in registers only.
 * Synchronises. */
int  esq_rk_pre_error(esq_ctx *ctx, double h, const double *e_pre,
                      const double *b_scale_pre, int rows, double *sumsq_out);
/* WHOLE STEPS of the pairs that test an early estimate (BS5, bogacki.py:238-338;
 * CFMR7osc, calvo.py:152-253).  esq_rk_set_pre registers the estimate -- weights
 * e_pre / b_scale_pre over K[0..rows), tested after stage rows - 1 -- and from then
 * on esq_rk_stages(1, s, t, h) runs the WHOLE attempt: the stages before the
 * estimate, the estimate (as the last target of the chain sweep that evaluates stage
 * rows - 1 where the plugin takes it: no pass of its own; its sum goes to a slot of
 * its own, nobody waits for it), and -- speculatively, as if the estimate had
 * passed -- the remaining stages.  esq_rk_solution_error then waits ONCE;
 * esq_rk_pre_result returns the early estimate's sum of that attempt (no wait of
 * its own beyond a visibility spin).  A caller that finds it > tolerance rejects the
 * attempt exactly as the piecewise sequence would have and counts the discarded
 * stages itself; K rows and states are bit-identical to esq_rk_stages(1, rows) +
 * esq_rk_pre_error + esq_rk_stages(rows, s).  rows == 0 clears.  ESQ_EINVAL unless
 * 2 <= rows <= s - 1. */
int  esq_rk_set_pre(esq_ctx *ctx, const double *e_pre, const double *b_scale_pre,
                    int rows);
int  esq_rk_pre_result(esq_ctx *ctx, double *sumsq_out);
/* Same pass with caller-supplied weights over K[0..rows): the embedded pairs of
 * the variable-order CKdisc (`_comp_sol_err_tol`, cash.py:397-401):
 *   sol = Y + h*sum b[j] K[j];  err = h*sum e[j] K[j];
 *   sum |err / (atol + rtol*max(|Y|, |sol|))|^2
 * store_ynew != 0 also writes sol into YNEW.  Synchronises. */
int  esq_rk_custom_sol_err(esq_ctx *ctx, double h, const double *b,
                           const double *e, int rows, int store_ynew,
                           double *sumsq_out);
/* Accept the attempt (common.py:289-303): non-FSAL tableaux with a device RHS
 * get K[s] = rhs(t_new, YNEW); then Y <-> YNEW are swapped and K[s] becomes the
 * new K[0] by pointer rotation (no copies).  with_end_eval = 0 skips the RHS
 * launch (host-RHS mode uploads K[s] itself before calling this).
 * h_next != 0: the step size the host controller intends to use next.  The
 * first stage argument of the next step, YSTAGE = Y + h_next*A[1][0]*K[0], is
 * then formed right away -- by the end-point sweep itself where the plugin has
 * a fused entry, else by a kernel that runs while the host is between steps.
 * The next esq_rk_stages(1, ..., h) uses it iff h == h_next exactly and no call
 * that may write a vector came in between; otherwise it is recomputed. */
int  esq_rk_accept(esq_ctx *ctx, double t_new, int with_end_eval, double h_next);
/* WORK = h * sum_j E[j] K[j]  (the vector `_estimate_error` returns,
 * common.py:333-336); K rows of the step just accepted if last_step != 0. */
int  esq_rk_error_vector(esq_ctx *ctx, double h, int last_step);
/* vector id (physical row, >= 0) of logical K row `logical_row` of the step in
 * flight, or of the step just accepted if last_step != 0; negative on misuse */
int  esq_rk_row_id(esq_ctx *ctx, int logical_row, int last_step);
/* logical->physical row of K for the step just accepted (for solver.K) */
int  esq_rk_download_last_K(esq_ctx *ctx, int row, double *host);
/* The launch plans ("step programs") esq_rk_stages would run for a method on one
 * of the built-in plugins ("bruss2d", "heat2d", "diff3d", "plain" = esq_rhs_fn
 * only) with the given entry capabilities, as text -- built exactly as on a
 * device, but on a detached context: no GPU is touched (the plugins answer the
 * library's queries on the host).  One line per starting state of a step:
 *   first / deferred / prelaunched: <launch> ... | launches=<n> words=<r>+<w> cost=<c>
 * (designed 8-byte words per element read + written, halo re-reads not counted; the
 * planner's cost of the sequence, which it minimises).  pre_rows != 0: with that early
 * estimate registered (esq_rk_set_pre) -- the whole-step program of BS5 / CFMR7osc.
 * tests/test_step_plans.py pins the plans of every tableau with it. */
int  esq_plan_describe(const char *plugin, int N, int s, const double *A,
                       const double *B, const double *C, const double *E, int fsal,
                       int chain_caps, int fuse_mask, int lazy_rows, int chain_depth,
                       int src_pays, const double *e_pre, const double *b_scale_pre,
                       int pre_rows, char *buf, size_t buflen);
/* Whole steps on a context WITHOUT a device: the host side of the step -- plans, row
 * maps, the first launch ahead of time and what it saves and restores, rows left
 * unwritten, the deferred end-point derivative -- with every launch replaced by the
 * plugin's answer to the corresponding query (arguments as esq_plan_describe).
 * `script`: one code per attempt, the calls RungeKutta._step_impl makes:
 *   0 accepted, the next step size named before the norm is known and confirmed
 *   1 accepted, the next step size named at accept time
 *   2 rejected (repeated with half the step)
 *   3 accepted, but the following attempt takes another step size
 *   4 accepted, then a reader asks for a row of K (esq_rk_row_id)
 *   5 accepted with the guess of 0, but the accept names another step size
 * One line per attempt:
 *   "<code>: state_ok=<0|1> used=<n> dropped=<n> missing=<rows> k0=<0|1> fused=<n> plain=<n>"
 * (state_ok: a launch ahead of time left the context's step state and row maps as it
 * found them, and the row maps are permutations disjoint from the spare rows; the
 * rest are the counters of esq_rk_launch_ahead_stats / esq_rk_lazy_rows).
 * tests/test_step_plans.py runs it, also against the sanitizer build of the library. */
int  esq_step_dry_run(const char *plugin, int N, int s, const double *A,
                      const double *B, const double *C, const double *E, int fsal,
                      int chain_caps, int fuse_mask, int lazy_rows, int chain_depth,
                      int src_pays, const double *e_pre, const double *b_scale_pre,
                      int pre_rows, const int *script, int n_attempts, char *buf,
                      size_t buflen);
/* Rows of K that only the solution / error sums of their own sweep read (the
 * stages of a step's last chain sweep, non-FSAL pairs: `self.K[s] = f` of
 * common.py:355 for rows nothing in `_step_impl` reads again) are NOT written
 * by esq_rk_stages.  Every entry point that reads rows of K (this block, the
 * dense output, esq_download / esq_copy on ESQ_SLOT_K, esq_rk_row_id ...)
 * first re-evaluates them, bit-identically, so callers never see the
 * difference; a context asked twice within four accepted steps writes them in
 * every step from then on.  ESQ_LAZY_ROWS=0 in the environment: always written.
 * Likewise the end-point derivative of a non-FSAL pair (`self.K[-1] = fun(t_new,
 * y_new)`, common.py:300-301 -- the next step's K[0]): esq_rk_accept leaves it to
 * the next esq_rk_stages, which evaluates it as one more stage in front of its
 * first chain sweep (the stage reads the state itself: no end-point sweep, no
 * first stage argument in memory); a reader of logical row 0 in between gets it
 * evaluated first.  ESQ_LAZY_END=0: evaluated by esq_rk_accept.
 * State for tests and tuning: rows currently not in memory, whether the context
 * now keeps its rows, re-evaluations so far, end-point derivatives evaluated
 * inside a chain sweep / by a sweep of their own (after the opt-out or a
 * reader, or where no chain sweep fits).  Any pointer may be NULL. */
int  esq_rk_lazy_rows(esq_ctx *ctx, int *missing_out, int *keeps_out,
                      long *restores_out, long *end_fused_out, long *end_plain_out);
/* on != 0: the caller takes WHOLE steps -- every accepted step is followed by
 * esq_rk_stages(1, s, ...) -- so esq_rk_accept(t_new, ., h_next != 0) may enqueue
 * the next step's first launch right away and esq_rk_solution_error_ahead may be
 * used (default off: callers that run the stages in pieces would pay for a launch
 * they then repeat).  The Python classes switch it on unless ESQ_LAUNCH_AHEAD=0. */
int  esq_rk_set_launch_ahead(esq_ctx *ctx, int on);
/* first launches made ahead of time (esq_rk_solution_error_ahead / esq_rk_accept)
 * that the next step used / that were dropped; either pointer may be NULL */
int  esq_rk_launch_ahead_stats(esq_ctx *ctx, long *used_out, long *dropped_out);

/* ---- dense output (common.py:358-368, 766-790) --------------------------- */
/* Device-resident interpolant (ref HornerDenseOutput, common.py:766-790): an
 * object of its own (it outlives the step and even the solver context) holding
 * Qh = h * K_last.T @ P (one fused pass over K) and the base state on the
 * device.  from_end != 0: the polynomial is anchored at the END of the step
 * (BS5 'best', bogacki.py:390-393): base = Y, else base = the pre-step state. */
typedef struct esq_dense esq_dense;
int  esq_dense_create(esq_ctx *ctx, const double *P, int rows, int p, double h,
                      int from_end, esq_dense **out);
/* The same object from ANY vectors of the context (ids as in esq_vec_*: a physical K
 * row >= 0 -- esq_rk_row_id -- or ESQ_VEC_Y ...):  Q_k = sum_j W[j*p + k] * vec_j
 * (ascending j, one pass), base = vec(base_vec).  The C1 cubic Hermite interpolant of
 * tableaux without P and of SSV2stab (common.py:793-821, sommeijer.py:400-406) in Horner
 * form: with d = y - y_old,
 *   y(x) = y_old + x (h f_old) + x^2 (3 d - 2 h f_old - h f) + x^3 (-2 d + h f_old + h f)
 * so that their dense_output() copies nothing to the host until it is evaluated. */
int  esq_dense_create_vecs(esq_ctx *ctx, const int *vec_ids, int nvec, const double *W,
                           int p, int base_vec, esq_dense **out);
/* host_out[n] = base + sum_c Qh[:,c] x^(c+1)  (Horner on the device, x scaled) */
int  esq_dense_eval(esq_dense *d, double x, double *host_out);
/* Qh as a (p, n) row-major host matrix (transpose of the reference's Q*h) */
int  esq_dense_download(esq_dense *d, double *Qh_host);
int  esq_dense_destroy(esq_dense *d);

/* Extra stages of BS5's 'low'/'best' interpolants (bogacki.py:356-368), on the
 * rows of the step just accepted:
 *   YSTAGE = y_old + h * sum_{j<count} a[j] * K_last[j]   (y_old: pre-step state)
 *   K_last[row] = rhs(t, YSTAGE)          -- or uploaded by the host */
int  esq_rk_dense_stage(esq_ctx *ctx, int row, const double *a, int count,
                        double h);
int  esq_rk_dense_eval(esq_ctx *ctx, int row, double t);
int  esq_rk_upload_last_K(esq_ctx *ctx, int row, const double *host);

/* vector ids of the esq_rkc_* / esq_vec_* / esq_hs_* family: id >= 0 is a
 * PHYSICAL K row (no rotation map), the negative ids name the fixed slots */
#define ESQ_VEC_NONE    (-1)
#define ESQ_VEC_Y       (-2)
#define ESQ_VEC_YNEW    (-3)
#define ESQ_VEC_YSTAGE  (-4)
#define ESQ_VEC_WORK    (-5)

/* ---- Runge-Kutta-Chebyshev (SSV2stab) launches -------------------------- */
/* Rows are vector ids chosen by the host (it rotates them instead of the
 * two full copies of sommeijer.py:318-319).
 * dst = yn + hmus * fn                              sommeijer.py:289          */
int  esq_rkc_first_stage(esq_ctx *ctx, int dst, int yn, int fn, double hmus);
/* dst = mu*yjm1 + nu*yjm2 + (1-mu-nu)*yn + hmus*(fy - ajm1*fn)
 *                                                   sommeijer.py:312-313      */
int  esq_rkc_stage(esq_ctx *ctx, int dst, int fy, int yjm1, int yjm2, int yn,
                   int fn, double mu, double nu, double hmus, double ajm1);
/* all m stages of one step incl. the RHS launches (sommeijer.py:273-329);
 * scalars[5*(j-2)..] = (mu, nu, hmus, ajm1, t_stage) for j = 2..m.  The result
 * is left in physical row *y_row_out. rows: yn, fn, three work rows and a fourth
 * one or ESQ_VEC_NONE: with four, and a plugin that has an RKC chain entry, the
 * stages run as chains (two iterates in, two out: a ping-pong of row pairs). */
int  esq_rkc_stages(esq_ctx *ctx, int yn, int fn, int w0, int w1, int w2, int w3,
                    double hmus1, int m, const double *scalars, int *y_row_out);
/* sum |(0.8*(yn - y) + 0.4*h*(fn + fy)) / (atol + rtol*max(|y|,|yn|))|^2
 *                                                   sommeijer.py:218-220      */
int  esq_rkc_error_norm(esq_ctx *ctx, int y, int yn, int fn, int fy, double h,
                        double *sumsq_out);
/* the tail of a Chebyshev step in one call (sommeijer.py:214-220):
 *   fy = rhs(t_end, y);  then the error sum of squares as esq_rkc_error_norm.
 * With a plugin whose fused entry takes ESQ_EPI_RKCERR this is ONE sweep + the
 * final sum (instead of RHS, error kernel, final sum).  Synchronises. */
/* esq_rkc_stages + esq_rkc_end_error in one call: all stages, f(t_end, y_{n+1}) and
 * the error estimate; rows of y_{n+1} and of its derivative in *y_row_out /
 * *fy_row_out (two of the work rows).  With a chain entry that takes the LAST form the
 * end of the step rides in the step's last chain sweep. */
int  esq_rkc_stages_end(esq_ctx *ctx, int yn, int fn, int w0, int w1, int w2, int w3,
                        double hmus1, int m, const double *scalars, double t_end,
                        double h, int *y_row_out, int *fy_row_out, double *sumsq_out);
/* The recursion of the NEXT step if this attempt is accepted and the controller keeps the
 * step (a run that sits at max_step): called right before esq_rkc_stages_end, which then
 * enqueues that step's opening chain sweep (FIRST form; y_{n+1} and its derivative are the
 * rows it has just produced) behind its own final sum, BEFORE the host waits for the error
 * norm -- the device does not idle while the host digests the step (35 us of a 1.17 ms
 * step at N = 159).  The next esq_rkc_stages_end skips that sweep if it is asked for
 * exactly this recursion on exactly those rows; anything else (a rejection, another step
 * size, any other call on the context) discards it: the sweep only wrote work rows
 * (counted by esq_rk_launch_ahead_stats).  scalars: the first 5 * min(m_next - 1,
 * ESQ_RKC_CHAIN_MAX_DEPTH) doubles of the table are read.  Plugins without an RKC chain
 * entry that opens a step: nothing is launched.                  (as esq_rk_set_launch_ahead:
 * common.py:222-308 has no counterpart, the results are those of the plain sequence) */
int  esq_rkc_guess_next(esq_ctx *ctx, double hmus1_next, int m_next, const double *scalars_next);
/* The launch sequence esq_rkc_stages_end runs for m stages with a chain entry of the
 * given depth and forms (ESQ_RKC_CHAIN_FIRST / _LAST or-ed into max_depth, as for
 * esq_set_rhs_rkc_chain) that takes the LAST form with up to end_slots_max stage
 * slots (the built-in 3-D sweep: 5, the 2-D one: 6), as text -- no GPU is touched:
 *   <launch> <launch> ... | launches=<n>
 * with the labels of esq_profile_kernels (rkc_chain<d>[-first|-end|-last], rhs_rkc,
 * k_rkc_first, rhs+rkcerr).  max_depth < 2: one launch per stage.
 * tests/test_step_plans.py pins the sequences with it. */
int  esq_rkc_plan_describe(int m, int max_depth, int end_slots_max, char *buf, size_t buflen);
int  esq_rkc_end_error(esq_ctx *ctx, int y, int yn, int fn, int fy, double t_end,
                       double h, double *sumsq_out);
/* generic K[dst] = rhs(t, K[src]) on physical rows   sommeijer.py:214, 311    */
int  esq_rkc_eval_rhs(esq_ctx *ctx, int dst, double t, int src);
/* sum x^2 and sum (x - y)^2 (np.linalg.norm pieces of sommeijer.py:350-374;
 * y == ESQ_VEC_NONE means "no subtraction") */
int  esq_vec_sumsq(esq_ctx *ctx, int x, int y, double *sumsq_out);
/* dst = a + alpha * (b - c)   (sommeijer.py:354, 389, 383; a and/or c may be
 * ESQ_VEC_NONE) */
int  esq_vec_axpbmc(esq_ctx *ctx, int dst, int a, double alpha, int b, int c);
/* sum |(a - b) / (atol + rtol*|w|)|^2               sommeijer.py:154-155      */
int  esq_vec_wdiff_sumsq(esq_ctx *ctx, int a, int b, int w, double *sumsq_out);

/* generic vector plumbing on vector ids (used by the starting-step estimate
 * and the spectral-radius iteration) */
/* value_im is the imaginary part for a complex context (ignored otherwise) */
int  esq_vec_fill(esq_ctx *ctx, int dst, double value, double value_im);
int  esq_vec_copy(esq_ctx *ctx, int dst, int src);
int  esq_vec_eval_rhs(esq_ctx *ctx, int dst, double t, int src);  /* dst = rhs(t, src) */
int  esq_vec_upload(esq_ctx *ctx, int dst, const double *host);
int  esq_vec_download(esq_ctx *ctx, int src, double *host);

/* add `count` zeroed work vectors to the context; their ids are
 * *first_id ... *first_id + count - 1 (used by the stiffness diagnosis, which
 * must not disturb the K rows of the step just taken) */
int  esq_aux_rows(esq_ctx *ctx, int count, int *first_id);
/* sum a.b / wt^2 with wt = max(0.5*(|y1| + |y2|), floor)   common.py:413-415,
 * 968, 1014 (RKSuite's weighted inner product of the stiffness diagnosis) */
int  esq_vec_wdot(esq_ctx *ctx, int a, int b, int y1, int y2, double floor_,
                  double *out);

/* ---- starting step size (Watts' dhstrt; ref h_start, common.py:519-763) --- */
/* sum and min over components of log10(atol + rtol*|y|)      common.py:725-727 */
int  esq_hs_log_etol(esq_ctx *ctx, int y, double *sum_out, double *min_out);
/* next perturbation direction                                 common.py:700-714
 *   dy = where(src, src, fill); spy = where(spy, spy, yp);
 *   yp = where(spy, copysign(dy, spy), dy)   (componentwise for complex) */
int  esq_hs_select(esq_ctx *ctx, int yp, int spy, int src, double fill);

/* ---- multi-GPU lock-step (BASELINE.json configs[4]; not in the reference) - */
/* comm is an ncclComm_t created by the caller (one rank per GPU); every
 * *_sumsq_out / error-norm entry point then all-reduces (sum) its double over
 * the communicator before returning, so all ranks take identical decisions.
 * n_total is informational (host divides by it). */
int  esq_set_comm(esq_ctx *ctx, void *nccl_comm);
/* helpers so the host never needs another RCCL binding */
int  esq_comm_unique_id(void *id128_out);              /* 128-byte ncclUniqueId */
int  esq_comm_init_rank(void **comm_out, int nranks, const void *id128, int rank,
                        int device);
int  esq_comm_destroy(void *comm);
/* number of ranks RCCL reports for the communicator (bench.py echoes it) */
int  esq_comm_count(void *comm, int *nranks_out);
/* ncclCommAbort: a rank that fails outside a collective calls this before it
 * exits so that its peers' pending all-reduce ends with an error instead of
 * blocking.  The library itself aborts the communicator when an all-reduce does
 * not complete within ESQ_COMM_TIMEOUT_S (default 120 s) and returns
 * ESQ_ETIMEOUT. */
int  esq_comm_abort(void *comm);
/* 1 if the library itself aborted the context's communicator (a lock-step
 * collective returned ESQ_ETIMEOUT): the handle passed to esq_set_comm is then
 * gone and must be neither aborted nor destroyed again by the host. */
int  esq_comm_is_aborted(const esq_ctx *ctx);
/* all-reduce `count` (<= 4) host doubles in place over the context's
 * communicator; a no-op without one.  Used for the rank-local scalars that feed
 * the step size or the stage count (SSV2stab's spectral radius,
 * sommeijer.py:174-204) so that all ranks stay in lock-step. */
#define ESQ_OP_SUM 0
#define ESQ_OP_MAX 1
#define ESQ_OP_MIN 2
int  esq_allreduce_scalars(esq_ctx *ctx, double *inout, int count, int op);

/* ---- built-in device RHS plugins (synthetic workloads of BASELINE.json) --- */
/* each *_create returns an opaque `user` pointer to pass with the matching
 * esq_rhs_* function to esq_set_rhs; free with esq_rhs_free. */
int  esq_rhs_diag_create(void **user_out, int device, const double *lam_host,
                         size_t n, double forcing_amp);  /* f = lam*y + amp*sin(t) */
/* complex twin: lam_host holds n_complex (re, im) pairs, the state is complex
 * (context created with is_complex = 1); f = lam*y + (amp_re + i amp_im)*sin(t).
 * Carries the reference's complex-state support (common.py:64-66, 187-190) onto
 * the device-RHS path, fused reducing epilogues included. */
int  esq_rhs_cdiag_create(void **user_out, int device, const double *lam_host,
                          size_t n_complex, double amp_re, double amp_im);
int  esq_rhs_heat2d_create(void **user_out, int N);       /* n = N*N            */
int  esq_rhs_bruss2d_create(void **user_out, int N, double alpha, double a,
                            double b);                    /* n = 2*N*N          */
int  esq_rhs_diff3d_create(void **user_out, int N);       /* n = N*N*N          */
int  esq_rhs_free(void *user);
/* the tuning switches of ONE built-in plugin object ("chain_rows=12;rkc_force=1": the
 * keys of esq_option_level == 2; the process defaults ESQ_<KEY> applied when the object
 * was made are replaced).  ESQ_EINVAL for any other key. */
int  esq_rhs_set_options(void *user, const char *options);
int  esq_rhs_diag(void *user, double t, const double *y, double *f, size_t n,
                  void *stream);
int  esq_rhs_cdiag(void *user, double t, const double *y, double *f, size_t n,
                   void *stream);
int  esq_rhs_heat2d(void *user, double t, const double *y, double *f, size_t n,
                    void *stream);
int  esq_rhs_bruss2d(void *user, double t, const double *y, double *f, size_t n,
                     void *stream);
int  esq_rhs_diff3d(void *user, double t, const double *y, double *f, size_t n,
                    void *stream);
/* RKC entries (esq_rhs_rkc_fn) */
int  esq_rhs_heat2d_rkc(void *user, double t, const double *yjm1,
                        const double *yjm2, const double *yn, const double *fn,
                        double mu, double nu, double omn, double hmus, double ajm1,
                        double *y_out, size_t n, void *stream, void *start_event,
                        void *stop_event);
int  esq_rhs_diff3d_rkc(void *user, double t, const double *yjm1,
                        const double *yjm2, const double *yn, const double *fn,
                        double mu, double nu, double omn, double hmus, double ajm1,
                        double *y_out, size_t n, void *stream, void *start_event,
                        void *stop_event);
/* RKC chain entry (esq_rhs_rkc_chain_fn) */
int  esq_rhs_heat2d_rkc_chain(void *user, const esq_rkc_chain *chain, size_t n,
                              void *hip_stream, void *start_event, void *stop_event);
int  esq_rhs_diff3d_rkc_chain(void *user, const esq_rkc_chain *chain, size_t n,
                              void *stream, void *start_event, void *stop_event);
/* fused entries (esq_rhs_fused_fn) */
int  esq_rhs_bruss2d_fused(void *user, double t, const double *y_in, double *f,
                           const esq_epilogue *epi, size_t n, void *stream,
                           void *start_event, void *stop_event);
int  esq_rhs_heat2d_fused(void *user, double t, const double *y_in, double *f,
                          const esq_epilogue *epi, size_t n, void *stream,
                          void *start_event, void *stop_event);
/* chain entries (esq_rhs_chain_fn) */
int  esq_rhs_bruss2d_chain(void *user, const double *y_in, const esq_chain *chain,
                           size_t n, void *stream, void *start_event,
                           void *stop_event);
int  esq_rhs_heat2d_chain(void *user, const double *y_in, const esq_chain *chain,
                          size_t n, void *stream, void *start_event,
                          void *stop_event);
int  esq_rhs_diff3d_chain(void *user, const double *y_in, const esq_chain *chain,
                          size_t n, void *stream, void *start_event,
                          void *stop_event);
int  esq_rhs_diff3d_fused(void *user, double t, const double *y_in, double *f,
                          const esq_epilogue *epi, size_t n, void *stream,
                          void *start_event, void *stop_event);
int  esq_rhs_diag_fused(void *user, double t, const double *y_in, double *f,
                        const esq_epilogue *epi, size_t n, void *stream,
                        void *start_event, void *stop_event);
int  esq_rhs_cdiag_fused(void *user, double t, const double *y_in, double *f,
                         const esq_epilogue *epi, size_t n, void *stream,
                         void *start_event, void *stop_event);

/* ---- measurement (bench.py `roofline`) ------------------------------------ */
/* class_mask bit k = 1: every launch of kernel class k carries a start/stop HIP
 * event pair on the context's stream (the library's own kernels take them as
 * dispatch timestamps via hipExtLaunchKernelGGL, RHS plugins are bracketed with
 * hipEventRecord); esq_profile_read synchronises and returns the summed device
 * time, launch count and algorithmic bytes since the last reset. */
#define ESQ_PROF_STAGE     0   /* stage-accumulate kernels                     */
#define ESQ_PROF_RHS       1   /* RHS plugin launches                          */
#define ESQ_PROF_SOLERR    2   /* solution / error-norm kernels (+final sum)   */
#define ESQ_PROF_RKC       3   /* RKC stage kernels                            */
#define ESQ_PROF_NCLASS    4
int  esq_profile_enable(esq_ctx *ctx, int class_mask);
/* time only every `every`-th launch of an enabled class (an event-carrying
 * dispatch costs ~6 us of queue time; a prime stride samples all stages evenly) */
int  esq_profile_sampling(esq_ctx *ctx, int every);
int  esq_profile_read(esq_ctx *ctx, int klass, double *total_ms, long *launches,
                      double *bytes);
/* bytes the timed launches of a class were DESIGNED to move (equals the
 * algorithmic bytes except where blocked accumulation reads K rows once for
 * several stages) */
int  esq_profile_read_moved(esq_ctx *ctx, int klass, double *moved_bytes);
/* per-kernel table since the last reset, one line per kernel label:
 *   name \t class \t launches \t total_ms \t algorithmic_bytes \t moved_bytes
 *        \t floor_bytes \n
 * moved_bytes: the launches' designed traffic, the halo points the marching
 * sweeps' tiles read twice included (mostly L2 hits); floor_bytes: the same with
 * every vector read once and every output written once -- what must cross the
 * memory interface (NUL-terminated text in buf; bench.py's `roofline.kernels`) */
int  esq_profile_kernels(esq_ctx *ctx, char *buf, size_t buflen);
int  esq_profile_reset(esq_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* EXTENSISQ_AMD_H */
