"""Device-side objects: the per-solver HBM context and device RHS plugins.

A `DeviceRHS` stands where the Python callable `fun(t, y)` stands in the
reference (extensisq/common.py:356): it is still callable on host ndarrays (so
scipy's own plumbing keeps working), but it also carries a C function pointer
(`esq_rhs_fn`, include/extensisq_amd.h) that enqueues the evaluation on the
solver's HIP stream, so the state never leaves HBM during a step.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (SLOT_ATOL, SLOT_K, SLOT_WORK, SLOT_Y, SLOT_YNEW,  # noqa: F401
                   SLOT_YSTAGE, DeviceError, as_ptr, check)


class WarmBuffers:
    """Pre-faulted host arrays for the per-step state downloads of large systems.

    scipy keeps every accepted `solver.y` by reference (ivp.py:665, 702), so each
    step needs a FRESH host array.  At n = 1e7 the 80 MB copy itself takes 1.4 ms
    (PCIe, pinned or not) but first-touching 80 MB of new pages takes 4-5 ms --
    the page faults, not the copy, made plain `solve_ivp` 5x slower than the
    HBM-resident step.  A couple of daemon threads therefore fault the next
    arrays in ahead of time (`memset` through ctypes: the GIL is released, the
    cores are otherwise idle) while the GPU computes the step; `take` hands out a
    warm array when one is ready and an ordinary cold one otherwise.  The arrays
    are plain NumPy arrays: whoever holds them owns them, nothing to give back.
    Engaged only for repeated downloads of >= 8 MB."""

    MIN_BYTES = 8 << 20

    def __init__(self, depth=6, workers=4):
        import threading
        self.depth, self.workers = depth, workers
        self._lock = threading.Lock()
        self._wake = threading.Condition(self._lock)
        self._ready = []            # warm uint8 arrays of size self._nbytes
        self._nbytes = 0
        self._pending = 0
        self._seen = {}             # nbytes -> downloads so far
        self._threads = []
        self._memset = None
        self._pin = False           # page-lock the warm arrays too (`take(pinned=True)`)
        self._pinned = set()        # addresses of ready arrays that are page-locked
        self._expected = 0          # arrays asked for ahead of the pattern (`expect`)
        self.enabled = os.environ.get("ESQ_WARM_BUFFERS", "1") != "0"

    def _worker(self):
        while True:
            with self._wake:
                while not (self._nbytes and
                           len(self._ready) + self._pending < self._target()):
                    self._wake.wait()
                nbytes = self._nbytes
                self._pending += 1
            buf = np.empty(nbytes, dtype=np.uint8)
            self._memset(buf.ctypes.data, 0, nbytes)      # faults the pages in
            # a caller that copies beside the step (esq_snapshot_copy) wants its
            # destination page-locked: 0.2 ms for 80 MB of resident pages, here
            # instead of on the copy worker
            pinned = self._pin and _lib.load().esq_host_pin(buf.ctypes.data, nbytes) == 0
            with self._wake:
                self._pending -= 1
                if nbytes == self._nbytes and len(self._ready) < self.depth:
                    self._ready.append(buf)
                    if pinned:
                        self._pinned.add(buf.ctypes.data)
                    pinned = False
            if pinned:                                    # not wanted any more
                _lib.load().esq_host_unpin(buf.ctypes.data)

    def _target(self):
        # arrays to keep ready: the full depth once the pattern has shown (two downloads
        # of the size), before that only what `expect` asked for
        return self.depth if self._seen.get(self._nbytes, 0) >= 2 else self._expected

    def _start(self):
        import threading
        libc = C.CDLL(None)
        self._memset = libc.memset
        self._memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self._memset.restype = C.c_void_p
        for _ in range(self.workers):
            th = threading.Thread(target=self._worker, daemon=True,
                                  name="esq-warm-buffers")
            th.start()
            self._threads.append(th)

    def _drop_ready(self):
        for buf in self._ready:
            if buf.ctypes.data in self._pinned:
                _lib.load().esq_host_unpin(buf.ctypes.data)
        self._ready = []
        self._pinned = set()

    def expect(self, nbytes, count=2):
        """downloads of `nbytes` are about to begin: start making `count` page-locked
        arrays now (the pattern has not shown yet: `take` engages after two downloads)"""
        nbytes = int(nbytes)
        if not self.enabled or nbytes < self.MIN_BYTES:
            return
        with self._wake:
            if nbytes != self._nbytes:
                self._nbytes = nbytes
                self._drop_ready()
            self._seen = {nbytes: max(self._seen.get(nbytes, 0), 1)}
            self._pin = True
            if len(self._ready) + self._pending < count:
                self._expected = count
            if not self._threads:
                self._start()
            self._wake.notify_all()

    def take(self, n, dtype, pinned=False):
        """an (n,) array of `dtype` to download into; warm if one is ready.
        pinned=True: -> (array, is_page_locked); the caller (the copy worker's
        esq_snapshot_copy) releases the lock.  Arrays handed out otherwise are
        never page-locked."""
        nbytes = int(n) * np.dtype(dtype).itemsize
        if not self.enabled or nbytes < self.MIN_BYTES:
            out = np.empty(n, dtype=dtype)
            return (out, False) if pinned else out
        locked = False
        with self._wake:
            count = self._seen.get(nbytes, 0) + 1
            self._seen = {nbytes: count}
            if count < 2:                     # a pattern, not a one-off
                out = np.empty(n, dtype=dtype)
                return (out, False) if pinned else out
            if nbytes != self._nbytes:
                self._nbytes = nbytes
                self._drop_ready()
            if pinned:
                self._pin = True
            buf = self._ready.pop() if self._ready else None
            if buf is not None and buf.ctypes.data in self._pinned:
                self._pinned.discard(buf.ctypes.data)
                locked = True
            if not self._threads:
                self._start()
            self._wake.notify_all()
        if buf is None:
            out = np.empty(n, dtype=dtype)
            return (out, False) if pinned else out
        if locked and not pinned:
            _lib.load().esq_host_unpin(buf.ctypes.data)
            locked = False
        return (buf.view(dtype), locked) if pinned else buf.view(dtype)


_warm = WarmBuffers()


# Contexts (and plugin data) whose Python owner has been garbage-collected, waiting to be
# destroyed.  A solver is always part of a reference cycle (scipy's OdeSolver keeps
# closures over itself), so it is freed by the CYCLIC collector -- inside an arbitrary
# allocation, in whichever thread happens to allocate (the copy worker, a warm-buffer
# thread, the main thread between two statements of CopyWorker.submit), possibly while
# another solver's state download is in flight.  A finalizer that destroys a context
# there synchronises a stream, frees 2 GB (which waits for the whole device) and -- if
# it first waits for the copy worker, as close() must -- can wait for a job its own
# thread has counted but not queued yet: 60 s per occurrence until round 5.  Finalizers
# therefore only PARK what they own (DeviceContext.park); it is destroyed at the next
# well-defined point: when a context is made or closed explicitly, or at exit.
_graveyard = __import__("collections").deque()     # (append / popleft: atomic, no lock --
                                                   # a finalizer may run inside ANY allocation)


# interpolants (esq_dense_*) of collected DeviceHornerDenseOutput objects: they own their
# device memory and no state download reads from them, so they need not wait for the copy
# worker -- destroyed by the next step / interpolant / context of the thread that drives
# solvers (a t_eval run drops one interpolant per step: they must not pile up until the
# next context is made)
_dense_dead = __import__("collections").deque()


def drain_dense():
    while _dense_dead:
        try:
            free, handle = _dense_dead.popleft()
        except IndexError:
            break
        try:
            free(handle)
        except Exception:                                     # noqa: BLE001
            pass


def _drain_graveyard(timeout=60.0):
    drain_dense()
    dead = []
    while True:
        try:
            dead.append(_graveyard.popleft())
        except IndexError:
            break
    if dead:
        from . import lazy
        # no state download may still read from them: a copy still in flight after the
        # timeout keeps its source alive -- the entries go back (at exit: they are left
        # to the process's teardown rather than freed under a running copy)
        if not lazy._worker.wait_idle(timeout):
            _graveyard.extend(dead)
            return
    for free, what in dead:             # (esq_destroy, context) / (esq_rhs_free, plugin data)
        try:                            # / (esq_dense_destroy, interpolant)
            free(what)
        except Exception:                                     # noqa: BLE001
            pass


__import__("atexit").register(_drain_graveyard, 2.0)


class DeviceContext:
    """Thin owner of one `esq_ctx` (one device, one stream, one HBM slab)."""

    # host-RHS problems up to this many doubles per vector keep their vectors
    # in pinned, device-mapped host memory (ESQ_HOST_SLAB=0 switches it off)
    HOST_SLAB_MAX_DOUBLES = 8192

    def __init__(self, n, n_rows, is_complex=False, device=0, host_rhs=False,
                 options=None):
        self.lib = _lib.load()
        _drain_graveyard()              # contexts of collected solvers: freed here
        # this context's switches (`esq_options=` of the solver that owns it)
        self.options = options if isinstance(options, _lib.Options) else _lib.Options(options)
        self.n = int(n)
        self.n_rows = int(n_rows)
        self.is_complex = bool(is_complex)
        self.dtype = np.complex128 if is_complex else np.float64
        self.device = int(device)
        doubles = self.n * (2 if is_complex else 1)
        self.host_slab = (bool(host_rhs) and doubles <= self.HOST_SLAB_MAX_DOUBLES
                          and self.options.get("host_slab", "1") != "0")
        handle = C.c_void_p()
        code = self.lib.esq_create2(
            C.byref(handle), self.device, self.n, self.n_rows,
            int(self.is_complex), _lib.CREATE_HOST_SLAB if self.host_slab else 0,
            self.options.context_string)
        self.handle = handle
        if code != 0:
            msg = self.lib.esq_last_error(handle) if handle else b""
            if handle:
                self.lib.esq_destroy(handle)
            self.handle = None
            raise DeviceError(
                f"esq_create(device={device}, n={n}) failed with code {code}: "
                f"{msg.decode(errors='replace') if msg else ''} -- an MI355X "
                "and the built HIP library are required; there is no CPU path")
        self._rhs_keepalive = None

    # -- lifetime
    def close(self):
        """destroy the context now (the caller's thread, the caller's moment)"""
        if getattr(self, "handle", None):
            from . import lazy
            lazy._worker.wait_idle()
            self.lib.esq_destroy(self.handle)
            self.handle = None
        _drain_graveyard()

    def park(self):
        """give the context up WITHOUT destroying it now: it is destroyed when the next
        context is made or closed, or at exit (_graveyard).  Never blocks, takes no lock:
        this is what finalizers call -- the collector runs them inside any allocation of
        any thread, e.g. between the two halves of CopyWorker.submit, where a wait for
        the copy worker would wait for the job its own caller has not queued yet."""
        try:
            handle, self.handle = getattr(self, "handle", None), None
            if handle:
                _graveyard.append((self.lib.esq_destroy, handle))
        except Exception:                                     # noqa: BLE001
            pass

    def __del__(self):
        self.park()

    def _chk(self, code, what):
        check(code, self.handle, what)

    # -- data
    def upload(self, slot, row, arr):
        dt = np.float64 if slot == SLOT_ATOL else self.dtype
        a = np.ascontiguousarray(arr, dtype=dt)
        if a.shape != (self.n,):
            raise ValueError(f"expected shape ({self.n},), got {a.shape}")
        self._chk(self.lib.esq_upload(self.handle, slot, row, as_ptr(a)),
                  "esq_upload")

    def download(self, slot, row=0):
        dt = np.float64 if slot == SLOT_ATOL else self.dtype
        out = _warm.take(self.n, dt)
        self._chk(self.lib.esq_download(self.handle, slot, row, as_ptr(out)),
                  "esq_download")
        return out

    def download_last_K(self, row):
        out = np.empty(self.n, dtype=self.dtype)
        self._chk(self.lib.esq_rk_download_last_K(self.handle, row, as_ptr(out)),
                  "esq_rk_download_last_K")
        return out

    def copy(self, dst_slot, dst_row, src_slot, src_row):
        self._chk(self.lib.esq_copy(self.handle, dst_slot, dst_row, src_slot,
                                    src_row), "esq_copy")

    def synchronize(self):
        self._chk(self.lib.esq_synchronize(self.handle), "esq_synchronize")

    # -- method description
    def set_tableau(self, A, B, Cc, E, fsal):
        s = len(B)
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        Cc = np.ascontiguousarray(Cc, dtype=np.float64)
        E = np.ascontiguousarray(E, dtype=np.float64)
        if A.shape != (s, s) or Cc.shape != (s,) or E.shape != (s + 1,):
            raise ValueError("inconsistent tableau shapes")
        self._chk(self.lib.esq_rk_set_tableau(self.handle, s, as_ptr(A),
                                              as_ptr(B), as_ptr(Cc), as_ptr(E),
                                              int(bool(fsal))),
                  "esq_rk_set_tableau")

    def set_tol(self, rtol, atol):
        at = np.atleast_1d(np.ascontiguousarray(atol, dtype=np.float64))
        self._chk(self.lib.esq_set_tol(self.handle, float(rtol), as_ptr(at),
                                       at.size), "esq_set_tol")

    def set_rhs(self, rhs):
        """rhs: a bound DeviceRHS (or None to clear)"""
        if rhs is None:
            self._chk(self.lib.esq_set_rhs(self.handle, None, None), "esq_set_rhs")
            self._rhs_keepalive = None
            return
        fn, user = rhs._bind(self)
        self._rhs_keepalive = (rhs, fn)
        self._chk(self.lib.esq_set_rhs(self.handle, C.cast(fn, C.c_void_p), user),
                  "esq_set_rhs")
        # fused entry: every RHS sweep also does the Runge-Kutta arithmetic that
        # follows it (next stage argument, blocked accumulation, solution + error
        # norm) while the derivative is in registers -- bit-identical K rows and
        # states, one kernel per RHS evaluation.  ESQ_CHAIN=0 / 1 switches the
        # entry off / forces it on; ESQ_FUSE=stage,block,solerr,errnorm,src (any
        # subset, default all) selects the epilogue kinds for A/B runs ("src":
        # the first sweep of a step forms its own input from y and K[0]).
        fused = rhs._fused_entry(self.lib)
        want = self.options.get("chain", "")
        use = (want == "1") or (want != "0" and rhs._fuse_default)
        if fused is not None and use:
            kinds = {"stage": 1 << _lib.EPI_STAGE, "block": 1 << _lib.EPI_BLOCK,
                     "solerr": 1 << _lib.EPI_SOLERR,
                     "errnorm": 1 << _lib.EPI_ERRNORM,
                     "rkcerr": 1 << _lib.EPI_RKCERR, "src": _lib.FUSE_SRC}
            sel = self.options.get("fuse", "")
            mask = _lib.FUSE_ALL if rhs._fuse_mask is None else rhs._fuse_mask
            mask |= _lib.FUSE_SRC if rhs._fuse_src else 0
            supported = mask
            if sel:
                mask = 0
                for name in sel.split(","):
                    if name.strip() not in kinds:
                        raise ValueError(f"ESQ_FUSE: unknown epilogue {name!r}")
                    mask |= kinds[name.strip()]
                mask &= supported
            # the entry answers the library's side-effect-free queries
            mask |= _lib.FUSE_QUERY if rhs._fuse_query else 0
            self._chk(self.lib.esq_set_rhs_fused(self.handle,
                                                 C.cast(fused, C.c_void_p), mask),
                      "esq_set_rhs_fused")
        # chain entry: several stages per marching sweep (needs the fused entry
        # for the remaining single stages); ESQ_CHAIN_DEPTH=1 switches it off
        chain = rhs._chain_entry(self.lib)
        if chain is not None and fused is not None and use:
            caps = int(rhs._chain_caps)
            # A/B switches of the round-6 chain kinds: the FSAL end-point stage and the
            # error norm inside the chain / an early estimate whose y_pre is not stored
            if self.options.get("chain_errnorm", "1") == "0":
                caps &= ~_lib.CHAIN_CAP_ERRNORM
            if self.options.get("chain_pre", "1") == "0":
                caps &= ~_lib.CHAIN_CAP_PRE
            self._chk(self.lib.esq_set_rhs_chain(self.handle,
                                                 C.cast(chain, C.c_void_p), caps),
                      "esq_set_rhs_chain")
        # RKC entry: derivative + Chebyshev recursion in one sweep
        rkc = rhs._rkc_entry(self.lib)
        if rkc is not None and self.options.get("rkc_chain", "1") != "0":
            self._chk(self.lib.esq_set_rhs_rkc(self.handle,
                                               C.cast(rkc, C.c_void_p)),
                      "esq_set_rhs_rkc")
            # RKC chain entry: several Chebyshev stages per marching sweep
            # (ESQ_RKC_DEPTH=1 in the environment: one launch per stage)
            # (plugin classes written against round 5 take `lib` only)
            import inspect
            takes_options = len(inspect.signature(rhs._rkc_chain_entry).parameters) >= 2
            rkc_chain = (rhs._rkc_chain_entry(self.lib, self.options) if takes_options
                         else rhs._rkc_chain_entry(self.lib))
            if rkc_chain is not None:
                self._chk(self.lib.esq_set_rhs_rkc_chain(
                    self.handle, C.cast(rkc_chain[0], C.c_void_p),
                    int(rkc_chain[1])), "esq_set_rhs_rkc_chain")

    # -- scalar-returning launches
    def _scalar(self, fn, what, *args):
        out = C.c_double()
        self._chk(fn(self.handle, *args, C.byref(out)), what)
        return out.value

    def rk_error_norm_sumsq(self, h):
        return self._scalar(self.lib.esq_rk_error_norm, "esq_rk_error_norm", h)

    def rk_solution_error_sumsq(self, t, h, h_next=0.0):
        """h_next != 0: the step size of the next step IF this attempt is accepted --
        its first launch then goes into the queue behind the error norm
        (esq_rk_solution_error_ahead)"""
        if h_next:
            return self._scalar(self.lib.esq_rk_solution_error_ahead,
                                "esq_rk_solution_error_ahead", t, h, h_next)
        return self._scalar(self.lib.esq_rk_solution_error,
                            "esq_rk_solution_error", t, h)

    def rk_set_pre(self, e_pre, b_scale_pre):
        """register the early error estimate (weights over K[0..len)): from then on
        `esq_rk_stages(1, s, ...)` runs the whole attempt, the estimate inside it"""
        e = np.ascontiguousarray(e_pre, dtype=np.float64)
        b = np.ascontiguousarray(b_scale_pre, dtype=np.float64)
        self._chk(self.lib.esq_rk_set_pre(self.handle, as_ptr(e), as_ptr(b), len(e)),
                  "esq_rk_set_pre")

    def rk_pre_result_sumsq(self):
        return self._scalar(self.lib.esq_rk_pre_result, "esq_rk_pre_result")

    def rk_pre_error_sumsq(self, h, e_pre, b_scale_pre):
        e = np.ascontiguousarray(e_pre, dtype=np.float64)
        b = np.ascontiguousarray(b_scale_pre, dtype=np.float64)
        return self._scalar(self.lib.esq_rk_pre_error, "esq_rk_pre_error", h,
                            as_ptr(e), as_ptr(b), len(e))

    # -- profiling (bench.py)
    def profile_enable(self, classes=(0, 1, 2, 3), every=1):
        """classes: iterable of PROF_* ids to time, or a false value to stop;
        every: time only every `every`-th launch of a class"""
        self._chk(self.lib.esq_profile_sampling(self.handle, int(every)),
                  "esq_profile_sampling")
        mask = 0
        for k in (classes or ()):
            mask |= 1 << int(k)
        self._chk(self.lib.esq_profile_enable(self.handle, mask),
                  "esq_profile_enable")

    def profile_reset(self):
        self._chk(self.lib.esq_profile_reset(self.handle), "esq_profile_reset")

    def profile_read(self, klass):
        ms, cnt, by = C.c_double(), C.c_long(), C.c_double()
        self._chk(self.lib.esq_profile_read(self.handle, klass, C.byref(ms),
                                            C.byref(cnt), C.byref(by)),
                  "esq_profile_read")
        return ms.value, cnt.value, by.value

    def profile_kernels(self):
        """per-kernel rows since the last reset:
        (name, class, launches, total_ms, algorithmic_bytes, moved_bytes,
        floor_bytes)"""
        buf = C.create_string_buffer(1 << 16)
        self._chk(self.lib.esq_profile_kernels(self.handle, buf, len(buf)),
                  "esq_profile_kernels")
        rows = []
        for line in buf.value.decode().splitlines():
            name, klass, launches, ms, alg, moved, floor = line.split("\t")
            rows.append((name, int(klass), int(launches), float(ms), float(alg),
                         float(moved), float(floor)))
        return rows

    def profile_read_moved(self, klass):
        mv = C.c_double()
        self._chk(self.lib.esq_profile_read_moved(self.handle, klass,
                                                  C.byref(mv)),
                  "esq_profile_read_moved")
        return mv.value


# ----------------------------------------------------------------------------
# device RHS plugins
# ----------------------------------------------------------------------------
class DeviceRHS:
    """Base class of right-hand sides that run on the GPU.

    Subclasses implement `_create(lib, device) -> (fn_ptr, user_ptr)` and give
    `n` (number of doubles of the state).  Instances are callable on host
    arrays: `rhs(t, y)` uploads, evaluates on the device and downloads -- used
    only off the hot path (first-step estimate, user curiosity)."""

    n = None
    is_complex = False
    _fuse_default = False      # use the fused entry unless ESQ_CHAIN says otherwise
    _fuse_src = False          # the fused entry accepts the on-the-fly first-stage input
    _fuse_mask = None          # epilogue kinds the fused entry implements (None: all)
    _fuse_query = False        # the fused entry honours esq_epilogue.dry_run
    _owns_user = True          # the user pointers of _create are freed with esq_rhs_free

    def __init__(self):
        self._bound = {}       # (device, plugin switches) -> (fn, user)
        self._host_ctx = {}    # device -> DeviceContext of __call__
        self._last_device = 0

    def _create(self, lib, device):
        raise NotImplementedError

    def _fused_entry(self, lib):
        """optional `esq_rhs_fused_fn` of this plugin"""
        return None

    def _rkc_entry(self, lib):
        """optional `esq_rhs_rkc_fn` of this plugin"""
        return None

    def _chain_entry(self, lib):
        """optional `esq_rhs_chain_fn` of this plugin"""
        return None

    def _rkc_chain_entry(self, lib, options=None):
        """optional `esq_rhs_rkc_chain_fn` of this plugin and the deepest chain
        it runs: (entry, max_depth) or None"""
        return None

    # ESQ_CHAIN_CAP_* bits of the chain entry (include/extensisq_amd.h): which
    # optional forms of a chain it handles; 0 = plain chains only
    _chain_caps = 0

    def _bind(self, ctx):
        if ctx.n != self.n:
            raise ValueError(f"RHS is for n={self.n}, solver state has n={ctx.n}")
        self._last_device = ctx.device
        # one plugin object per device AND set of plugin switches: two solvers with
        # different `esq_options` sharing this RHS do not share its tile geometry
        opts = ctx.options.plugin_string
        key = (ctx.device, opts)
        if key not in self._bound:
            fn, user = self._create(ctx.lib, ctx.device)
            if opts:
                if not (self._owns_user and user):
                    raise ValueError("esq_options: plugin switches "
                                     f"({opts.decode()}) need a built-in plugin")
                check(ctx.lib.esq_rhs_set_options(user, opts), None, "esq_rhs_set_options")
            self._bound[key] = (fn, user)
        return self._bound[key]

    def __call__(self, t, y):
        y = np.asarray(y)
        if y.ndim != 1:
            raise ValueError("device RHS plugins take one state vector")
        # evaluate on the device this RHS was last bound to (a rank with
        # LOCAL_RANK != 0 must not allocate on GPU 0)
        device = self._last_device
        ctx = self._host_ctx.get(device)
        if ctx is None:
            ctx = DeviceContext(self.n, 2, self.is_complex, device)
            ctx.set_rhs(self)
            self._host_ctx[device] = ctx
        ctx.upload(SLOT_Y, 0, y)
        ctx._chk(ctx.lib.esq_rk_eval_rhs(ctx.handle, 0, float(t), SLOT_Y, 0),
                 "esq_rk_eval_rhs")
        return ctx.download(SLOT_K, 0)

    def close(self):
        lib = _lib.load()
        for ctx in self._host_ctx.values():
            ctx.close()
        self._host_ctx = {}
        for fn, user in self._bound.values():
            if user:
                lib.esq_rhs_free(user)
        self._bound = {}

    def __del__(self):
        # (a finalizer: nothing that blocks or frees device memory here -- park())
        try:
            lib = _lib.load()
            for ctx in self._host_ctx.values():
                ctx.park()
            self._host_ctx = {}
            for fn, user in self._bound.values():
                if user and self._owns_user:
                    _graveyard.append((lib.esq_rhs_free, user))
            self._bound = {}
        except Exception:                                     # noqa: BLE001
            pass


class _Builtin(DeviceRHS):
    _fuse_src = True
    _symbol = None
    _symbol_fused = None
    _symbol_rkc = None
    _symbol_chain = None
    _symbol_rkc_chain = None
    _rkc_chain_depth = 4              # Chebyshev stages per chain sweep (ESQ_RKC_MAXDEPTH)
    _rkc_chain_forms = _lib.RKC_CHAIN_FIRST | _lib.RKC_CHAIN_LAST
    # the built-in 2-D sweeps handle every form (incl. round 6's: an unstored early
    # estimate, the FSAL end-point stage inside the chain) and answer the planner's
    # queries
    _chain_caps = 31 | _lib.CHAIN_CAP_PRE | _lib.CHAIN_CAP_ERRNORM
    _fuse_query = True

    def _rkc_chain_entry(self, lib, options=None):
        if not self._symbol_rkc_chain:
            return None
        # (the forms the built-in chain sweep takes: opening / ending a step)
        options = options or _lib.Options()
        depth = int(options.get("rkc_maxdepth", self._rkc_chain_depth))
        return getattr(lib, self._symbol_rkc_chain), depth | self._rkc_chain_forms

    def _rkc_entry(self, lib):
        return getattr(lib, self._symbol_rkc) if self._symbol_rkc else None

    def _chain_entry(self, lib):
        return getattr(lib, self._symbol_chain) if self._symbol_chain else None


    def _fused_entry(self, lib):
        return getattr(lib, self._symbol_fused) if self._symbol_fused else None

    def _make_user(self, lib, device):
        raise NotImplementedError

    def _create(self, lib, device):
        user = self._make_user(lib, device)
        fn = getattr(lib, self._symbol)
        return fn, user


class DiagonalLinear(_Builtin):
    """f = lam * y + amp * sin(t)   (lam: vector of n reals, or of n complex
    numbers -- the state is then complex128, as the reference's `RungeKutta`
    accepts it, common.py:187-190)"""
    _fuse_default = True

    def __init__(self, lam, forcing_amp=0.0):
        super().__init__()
        self.is_complex = bool(np.iscomplexobj(lam) or np.iscomplexobj(forcing_amp))
        dt = np.complex128 if self.is_complex else np.float64
        self.lam = np.ascontiguousarray(lam, dtype=dt)
        self.amp = complex(forcing_amp) if self.is_complex else float(forcing_amp)
        self.n = self.lam.size
        kind = "cdiag" if self.is_complex else "diag"
        self._symbol = f"esq_rhs_{kind}"
        self._symbol_fused = f"esq_rhs_{kind}_fused"

    def _make_user(self, lib, device):
        user = C.c_void_p()
        if self.is_complex:
            check(lib.esq_rhs_cdiag_create(C.byref(user), device, as_ptr(self.lam),
                                           self.n, self.amp.real, self.amp.imag),
                  None, "esq_rhs_cdiag_create")
        else:
            check(lib.esq_rhs_diag_create(C.byref(user), device, as_ptr(self.lam),
                                          self.n, self.amp), None,
                  "esq_rhs_diag_create")
        return user


class Heat2D(_Builtin):
    """5-point heat equation on an N x N interior grid, Dirichlet 0
    (BASELINE.json configs[1], configs[4]); twin of oracle/problems.py."""
    _symbol = "esq_rhs_heat2d"
    _symbol_fused = "esq_rhs_heat2d_fused"
    _symbol_rkc = "esq_rhs_heat2d_rkc"
    _symbol_rkc_chain = "esq_rhs_heat2d_rkc_chain"
    _rkc_chain_forms = _lib.RKC_CHAIN_FIRST | _lib.RKC_CHAIN_LAST
    # (in 2-D the recomputed halo is cheap; tools/rkc2d_bench.py, ms/step by depth
    # 1 .. 6 at N = 2236: 0.716 0.520 0.393 0.322 0.288 0.268; at N = 1000 depth 5
    # 0.118, 6 0.124)
    _rkc_chain_depth = 5
    _symbol_chain = "esq_rhs_heat2d_chain"
    _fuse_default = True

    def __init__(self, N):
        super().__init__()
        self.N = int(N)
        self.n = self.N * self.N
        if self.N >= 1500:
            self._rkc_chain_depth = 6

    def _make_user(self, lib, device):
        user = C.c_void_p()
        check(lib.esq_rhs_heat2d_create(C.byref(user), self.N), None,
              "esq_rhs_heat2d_create")
        return user

    def spectral_radius(self):
        return 8.0 * (self.N + 1) ** 2


class Brusselator2D(_Builtin):
    """2-D Brusselator reaction-diffusion, periodic, y = [u.ravel(), v.ravel()]
    (BASELINE.json configs[2], the north-star workload)."""
    _symbol = "esq_rhs_bruss2d"
    _symbol_fused = "esq_rhs_bruss2d_fused"
    _symbol_chain = "esq_rhs_bruss2d_chain"
    _fuse_default = True

    def __init__(self, N, alpha=0.1, a=1.0, b=3.4):
        super().__init__()
        self.N = int(N)
        self.alpha, self.a, self.b = float(alpha), float(a), float(b)
        self.n = 2 * self.N * self.N

    def _make_user(self, lib, device):
        user = C.c_void_p()
        check(lib.esq_rhs_bruss2d_create(C.byref(user), self.N, self.alpha,
                                         self.a, self.b), None,
              "esq_rhs_bruss2d_create")
        return user

    def spectral_radius(self):
        return 8.0 * self.alpha * self.N * self.N


class Diffusion3D(_Builtin):
    """7-point diffusion on an N^3 interior grid, Dirichlet 0
    (BASELINE.json configs[3])."""
    _symbol = "esq_rhs_diff3d"
    _symbol_fused = "esq_rhs_diff3d_fused"
    _symbol_rkc = "esq_rhs_diff3d_rkc"
    _symbol_rkc_chain = "esq_rhs_diff3d_rkc_chain"
    _symbol_chain = "esq_rhs_diff3d_chain"
    # the chain sweeps start a step from the state and leave rows unwritten; they
    # do not form their own input from memory rows (no FROM_ROWS / SKIP_OUT)
    _chain_caps = _lib.CHAIN_CAP_QUERY | 1 | 2
    _fuse_default = True
    _fuse_src = False                 # no on-the-fly first-stage input in 3-D

    def __init__(self, N):
        super().__init__()
        self.N = int(N)
        self.n = self.N ** 3
        # Chebyshev stages per chain sweep: four while the sweep's vectors live in the
        # Infinity Cache (more stages = more recomputed halo points, fewer bytes: a tie
        # there), five beyond it (SSV2stab at N = 400: 19.6 instead of 20.0 ms/step)
        if 6 * 8 * self.n > (256 << 20):
            self._rkc_chain_depth = 5

    def _make_user(self, lib, device):
        user = C.c_void_p()
        check(lib.esq_rhs_diff3d_create(C.byref(user), self.N), None,
              "esq_rhs_diff3d_create")
        return user

    def spectral_radius(self):
        return 12.0 * (self.N + 1) ** 2


class CFunctionRHS(DeviceRHS):
    """A user-compiled plugin: `fn_ptr` is the address of an `esq_rhs_fn`
    (see include/extensisq_amd.h), `user_ptr` its opaque argument."""

    def __init__(self, fn_ptr, user_ptr, n, is_complex=False):
        super().__init__()
        self._fn = _lib.RHS_FN(fn_ptr) if isinstance(fn_ptr, int) else fn_ptr
        self._user = C.c_void_p(user_ptr) if isinstance(user_ptr, int) else user_ptr
        self.n = int(n)
        self.is_complex = bool(is_complex)

    _owns_user = False          # (the caller's pointer: never freed here)

    def _create(self, lib, device):
        return self._fn, None if self._user is None else self._user

    def close(self):
        for ctx in self._host_ctx.values():
            ctx.close()
        self._host_ctx = {}
        self._bound = {}
