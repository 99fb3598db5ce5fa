"""Adaptive explicit Runge-Kutta solvers whose state lives in MI355X HBM.

Host side of the hot path: `RungeKutta` keeps the scipy `OdeSolver` plugin
surface and the Butcher-tableau class attributes of the reference
(extensisq/common.py:69-368) -- so `solve_ivp(fun, t_span, y0, method=Pr8)` and
user-defined `class Heun(RungeKutta)` tableaux work unchanged -- but every
vector operation of a step is a HIP kernel launched through the C ABI
(include/extensisq_amd.h).  The step-size controller, the accept/reject
decision and the end-of-interval logic are scalar arithmetic and stay here on
the host; one double (the weighted error sum of squares) comes back per
attempt.

There is no CPU implementation of the step in this package: without the built
library and a GPU the constructor raises `DeviceError`.
"""
import ctypes
import logging
import os
from math import copysign, sqrt
from warnings import warn

import numpy as np
from scipy.integrate._ivp.base import DenseOutput, OdeSolver
from scipy.integrate._ivp.common import (validate_first_step,
                                         validate_max_step, warn_extraneous)

from ._lib import (SLOT_K, SLOT_WORK, SLOT_Y, SLOT_YNEW, SLOT_YSTAGE, VEC_NONE,
                   VEC_Y, VEC_YNEW, DeviceError, Options, as_ptr)
from .device import DeviceContext, DeviceRHS, _dense_dead, drain_dense
from .lazy import LazyState

LAZY_MIN_BYTES = 8 << 20       # states below this are downloaded at once


def _with_esq_options(init):
    """constructor decorator: the `esq_options=` keyword -- this solver's tuning switches
    (`_lib.Options`: the ESQ_* switches of DESIGN.md §3.4, lower case without the
    prefix).  They travel as ARGUMENTS -- to `esq_create2`, to `esq_rhs_set_options`, to
    this package's own decisions -- and nothing writes the process environment (until
    round 5 the keyword did, for the duration of the constructor: switches read later
    had no effect, two threads constructing solvers raced).  Unknown keys: ValueError."""
    import functools

    @functools.wraps(init)
    def wrapper(self, *args, esq_options=None, **kwargs):
        self._esq_options = Options(esq_options)
        return init(self, *args, **kwargs)
    return wrapper


def _read_by_solve_ivp(depth=2):
    """is `solver.y` being read by the loop of scipy's `solve_ivp` (ivp.py:665)?  That
    loop reads the state after EVERY step whether it uses it or not; only it gets the
    deferred mirror, so that direct users of a solver see plain ndarrays.  `depth`:
    frames between this function and the reader of the property"""
    import sys
    try:
        frame = sys._getframe(depth)
    except ValueError:
        return False
    code = frame.f_code
    return code.co_name == "solve_ivp" and code.co_filename.replace("\\", "/").endswith(
        "scipy/integrate/_ivp/ivp.py")

def _inside_solve_ivp(max_depth=16, want_events=False):
    """is scipy's `solve_ivp` among the callers (a solver constructed by it: ivp.py:590)?
    want_events: ... and was it given event functions?  (Asked ONCE, by the constructor:
    `frame.f_locals` is a snapshot the frame keeps -- read inside the loop it would hold
    the previous step's `y` alive, and a mirror somebody still holds is downloaded)"""
    import sys
    try:
        frame = sys._getframe(1)
    except ValueError:
        return False
    for _ in range(max_depth):
        if frame is None:
            return False
        code = frame.f_code
        if code.co_name == "solve_ivp" and code.co_filename.replace("\\", "/").endswith(
                "scipy/integrate/_ivp/ivp.py"):
            if not want_events:
                return True
            try:
                return frame.f_locals.get("events") is not None
            except Exception:                                 # noqa: BLE001
                return False
        frame = frame.f_back
    return False


# failed-step counter shared with the RKC module (reference: common.py:14)
NFS = np.array(0)
NFI = np.array(0)     # kept for import compatibility (implicit methods: unused)
NLS = np.array(0)

MIN_FACTOR = 0.2      # reference common.py:18-20
MAX_FACTOR = 4.0
MAX_FACTOR0 = 10

_SC_PRESETS = {"G": (0.7, -0.4, 0, 0.9),       # Gustafsson
               "S": (0.6, -0.2, 0, 0.9),       # Soederlind
               "standard": (1, 0, 0, 0.9)}


def validate_tol(rtol, atol, y):
    """RKSuite-style tolerance bounds, applied silently (ref common.py:30-54):
    atol >= sqrt(tiny), 10*epsneg <= rtol <= 0.1."""
    atol = np.asarray(atol)
    if atol.ndim > 0 and atol.shape != (y.size,):
        raise ValueError("`atol` has wrong shape.")
    if np.any(atol < 0):
        raise ValueError("`atol` must be positive.")
    if not isinstance(rtol, float):
        raise ValueError("`rtol` must be a float.")
    if rtol < 0:
        raise ValueError("`rtol` must be positive.")
    fi = np.finfo(y.dtype)
    return (np.minimum(np.maximum(rtol, 10 * fi.epsneg), 0.1),
            np.maximum(atol, sqrt(fi.tiny)))


def calculate_scale(atol, rtol, y, y_new, _mean=False):
    """host helper with the reference's signature (common.py:57-61); the hot
    path computes the same weights inside the error kernels."""
    if _mean:
        return atol + rtol * 0.5 * (np.abs(y) + np.abs(y_new))
    return atol + rtol * np.maximum(np.abs(y), np.abs(y_new))


def norm(x):
    """RMS norm of a host vector (ref common.py:64-66)."""
    return (np.real(x @ x.conjugate()) / x.size) ** 0.5


class HornerDenseOutput(DenseOutput):
    """Polynomial interpolant y_old + sum_c Q[:, c] x^(c+1), evaluated with
    Horner's rule on the host (ref common.py:766-790).  `scaled=True`: Q
    already carries the factor h (it was formed on the device)."""

    def __init__(self, t_old, t, y_old, Q, scaled=False):
        super().__init__(t_old, t)
        self.h = t - t_old
        self.Q = Q if scaled else Q * self.h
        self.y_old = y_old

    def _call_impl(self, t):
        x = (t - self.t_old) / self.h
        cols = self.Q.T
        acc = cols[-1, :, np.newaxis] * x
        for q in cols[-2::-1]:
            acc += q[:, np.newaxis]
            acc *= x
        acc += self.y_old[:, np.newaxis]
        return acc if t.shape else acc[:, 0]


class DeviceHornerDenseOutput(DenseOutput):
    """The same interpolant with Qh = h * K.T @ P and the base state kept in
    HBM (`esq_dense_*`): building it is one fused pass over the K rows, every
    evaluation is one Horner kernel plus an n-vector download.  It owns its
    device memory, so it survives the step and the solver."""

    def __init__(self, t_old, t, lib, handle, n, dtype):
        super().__init__(t_old, t)
        self.h = t - t_old
        self._lib, self._handle = lib, handle
        self._n, self._dtype = n, dtype

    def _eval(self, x):
        out = np.empty(self._n, dtype=self._dtype)
        code = self._lib.esq_dense_eval(self._handle, float(x), as_ptr(out))
        if code != 0:
            from ._lib import DeviceError
            raise DeviceError(f"esq_dense_eval failed with code {code}")
        return out

    def _call_impl(self, t):
        x = (t - self.t_old) / self.h
        if not t.shape:
            return self._eval(x)
        return np.stack([self._eval(xi) for xi in x], axis=1)

    def __del__(self):
        # (a finalizer: esq_dense_destroy synchronises a stream and may free device
        # memory -- parked like contexts, destroyed at the next well-defined point)
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            from .device import _dense_dead
            _dense_dead.append((self._lib.esq_dense_destroy, handle))


class CubicDenseOutput(DenseOutput):
    """C1 cubic Hermite interpolant (ref common.py:793-821)."""

    def __init__(self, t_old, t, y_old, y, f_old, f):
        super().__init__(t_old, t)
        self.h = t - t_old
        self.y_old, self.y, self.f_old, self.f = y_old, y, f_old, f

    def _call_impl(self, t):
        x = (t - self.t_old) / self.h
        omx2 = (1.0 - x) ** 2
        out = ((1.0 + 2.0 * x) * omx2 * self.y_old[:, np.newaxis]
               + x * omx2 * self.h * self.f_old[:, np.newaxis]
               + x ** 2 * (3.0 - 2.0 * x) * self.y[:, np.newaxis]
               + x ** 2 * (x - 1.0) * self.h * self.f[:, np.newaxis])
        return out if t.shape else out[:, 0]


def _cubic_interpolant(solver, t_old, t, y_old, y, f_old, f):
    """C1 cubic Hermite interpolant through (y_old, f_old), (y, f) -- vector ids of the
    solver's context -- as a device-resident Horner form (esq_dense_create_vecs):
    with d = y - y_old and x = (t - t_old) / h,
        y(x) = y_old + x (h f_old) + x^2 (3 d - 2 h f_old - h f) + x^3 (-2 d + h f_old + h f)
    (the expansion of ref common.py:812-821).  d is formed first -- 3 y - 3 y_old in one
    rounding -- then the small h f terms are added."""
    import ctypes
    h = t - t_old
    ids = (ctypes.c_int * 4)(y, y_old, f_old, f)
    W = np.array([[0.0, 3.0, -2.0],
                  [0.0, -3.0, 2.0],
                  [h, -2.0 * h, h],
                  [0.0, -h, h]])
    handle = ctypes.c_void_p()
    solver._chk(solver._lib.esq_dense_create_vecs(solver._ctx, ids, 4, as_ptr(W), 3, y_old,
                                                  ctypes.byref(handle)),
                "esq_dense_create_vecs")
    return DeviceHornerDenseOutput(t_old, t, solver._lib, handle, solver.n,
                                   solver._dev.dtype)


class LockstepGroup:
    """Membership of this solver in a lock-step batch (one rank per GPU):
    `comm` is an ncclComm_t handle (see extensisq_amd.lockstep), `n_total` the
    summed state dimension of all ranks.  The C library all-reduces the error
    sum of squares, so every rank takes the same accept/reject decision.

    Rank-local host scalars that feed the step size or the stage count (a
    y-dependent spectral-radius estimate, say) go through `allreduce` so that
    the ranks cannot leave lock-step.  `reduce_scalars(values, op)` replaces
    the RCCL path (CPU tests pass a gloo reducer)."""

    def __init__(self, comm, n_total, reduce_scalars=None, offset=None):
        self.comm = comm
        self.n_total = int(n_total)
        self._reduce = reduce_scalars
        # index of this shard's first element in the concatenated state (None:
        # unknown); only the spectral-radius iteration's zero-derivative branch
        # needs it (sommeijer.py:386-388 flips ONE element of the whole state)
        self.offset = None if offset is None else int(offset)
        self.debug = os.environ.get("ESQ_LOCKSTEP_DEBUG", "0") not in ("", "0")
        self._contexts = []       # DeviceContexts the communicator is set on

    # -- the communicator handle may be invalidated by the library itself: a
    #    collective that times out aborts it (esq_comm_is_aborted).  Aborting or
    #    destroying the freed handle again would be a use-after-free in RCCL.
    def attach(self, dev):
        self._contexts.append(dev)

    def sync_aborted(self):
        """forget the communicator if the library has aborted it; returns True
        if the handle is gone"""
        live = [d for d in self._contexts if getattr(d, "handle", None)]
        if self.comm and any(d.lib.esq_comm_is_aborted(d.handle) for d in live):
            # the library freed the handle when it aborted it: every other context
            # attached to the same communicator must drop its pointer too, or its
            # next reduction calls ncclAllReduce on a dead communicator
            for d in live:
                if not d.lib.esq_comm_is_aborted(d.handle):
                    d.lib.esq_set_comm(d.handle, None)
            self.comm = None
        return not self.comm

    def host_reduce(self, dev, values, op="sum"):
        """reduce scalars the LIBRARY computed shard-locally.  With an RCCL
        communicator every reducing entry point has all-reduced its result
        already (finish_reduction): identity.  With a host reducer standing in
        for RCCL (several solvers of one process, CPU tests) reduce here."""
        if self._reduce is not None and not self.comm:
            return self.allreduce(dev, values, op)
        return [float(v) for v in values]

    def allreduce(self, dev, values, op="max"):
        """all-reduce a few floats over the group; `dev`: the solver's
        DeviceContext (its stream and communicator carry the RCCL call)"""
        values = [float(v) for v in values]
        if self._reduce is not None:
            return [float(v) for v in self._reduce(values, op)]
        if not self.comm:
            return values
        import ctypes
        from ._lib import OP_MAX, OP_MIN, OP_SUM
        buf = (ctypes.c_double * len(values))(*values)
        try:
            dev._chk(dev.lib.esq_allreduce_scalars(
                dev.handle, buf, len(values),
                {"sum": OP_SUM, "max": OP_MAX, "min": OP_MIN}[op]),
                "esq_allreduce_scalars")
        except Exception:
            self.sync_aborted()
            raise
        return list(buf)

    def check_identical(self, dev, what, values):
        """debug mode (ESQ_LOCKSTEP_DEBUG=1): every rank must hold the same
        scalars -- max and min over the ranks coincide"""
        if not self.debug:
            return
        hi = self.allreduce(dev, values, "max")
        lo = self.allreduce(dev, values, "min")
        if hi != lo or any(v != h for v, h in zip(values, hi)):
            raise RuntimeError(
                f"ranks left lock-step: {what} = {list(values)} here, "
                f"min over ranks {lo}, max {hi}")



class _LazyStateMixin:
    """Deferred mirrors of the solver's state (lazy.py) for the solver classes: the
    state of generation g (g accepted steps) is on the device while the solver is at
    generation g (`_lazy_where(0)`) and g + 1 (`_lazy_where(1)`: the "previous state"
    the dense output starts from); `step()` retires older mirrors first."""

    def _lazy_init(self, nbytes, host_slab):
        # large device-resident states: `solver.y` READ BY scipy's solve_ivp loop is a
        # deferred mirror; every other caller gets the ndarray it always got.
        # ESQ_LAZY_Y=0: never; ESQ_LAZY_Y=always: for every caller
        mode = self._esq_options.get("lazy_y", "1")
        self._lazy_on = (self._device_rhs is not None and not host_slab
                         and nbytes >= LAZY_MIN_BYTES and mode != "0")
        self._lazy_always = mode == "always"
        # with event functions scipy hands the `y` it reads after every step to USER code
        # (`event(t, y)`, ivp.py:680): that code gets the ndarray the reference gives it
        # -- numba / Cython / torch.from_numpy callbacks need the buffer,
        # `isinstance(y, np.ndarray)` holds -- so no deferred mirrors then
        if self._lazy_on and not self._lazy_always and _inside_solve_ivp(want_events=True):
            self._lazy_on = False
        self._state_gen = 0          # accepted steps: which state the device holds
        self._lazy_live = []         # [weakref(mirror), generation, copy-done event]
        self._lazy_eager = False     # the caller stores its states: copy at once
        if self._lazy_on and not self._lazy_always and _inside_solve_ivp():
            # scipy will read the state after every step and, unless t_eval / dense_output
            # say otherwise, keep it: two page-locked arrays are made ready while the
            # constructor runs (the first two kept states are downloaded synchronously:
            # 12 + 22 ms into cold arrays, 1.5 ms each into these)
            from .device import _warm
            _warm.expect(nbytes, count=2)

    def _lazy_where(self, age):
        """(slot, row) of the state `age` accepted steps ago (age 0 or 1)"""
        raise NotImplementedError

    def _lazy_y(self):
        """the `y` property's value: the cached host copy, a mirror, or a download"""
        if self._y_host is None:
            lazy = self._lazy_on and (self._lazy_always or _read_by_solve_ivp(3))
            self._y_host = (self._new_lazy_state() if lazy
                            else self._dev.download(*self._lazy_where(0)))
        if isinstance(self._y_host, LazyState) and (
                self._y_host.materialized
                or not (self._lazy_always or _read_by_solve_ivp(3))):
            # (used already -- or a caller other than solve_ivp's loop asks: the array)
            self._y_host = self._y_host.materialize()
        return self._y_host

    # -- deferred mirrors of the state (lazy.py)
    def _begin_snapshot(self, slot, row=0):
        """-> (copy_fn, out): the download of (slot, row) as it is NOW, to be run by the
        copy worker beside the steps that follow (esq_snapshot_begin / _copy)"""
        import ctypes
        from .device import _warm
        out, locked = _warm.take(self.n, self._dev.dtype, pinned=True)
        token = ctypes.c_void_p()
        lib, ptr = self._lib, out.ctypes.data_as(ctypes.c_void_p)
        try:
            self._chk(lib.esq_snapshot_begin(self._ctx, slot, row, ctypes.byref(token)),
                      "esq_snapshot_begin")
        except Exception:
            if locked:
                lib.esq_host_unpin(ptr)
            raise

        def copy_fn():
            code = lib.esq_snapshot_copy(token, ptr, int(locked))
            if code != 0:
                raise DeviceError(f"esq_snapshot_copy failed with code {code}")
            return out
        return copy_fn, out

    def _new_lazy_state(self):
        import weakref
        gen = self._state_gen
        solver = self                 # (an unread mirror keeps its solver alive)

        def fetch():
            age = solver._state_gen - gen
            if age in (0, 1):         # (1: after one more accept, the "previous state")
                return solver._dev.download(*solver._lazy_where(age))
            raise DeviceError("this state is no longer on the device")   # (retired before)
        mirror = LazyState(fetch, self.n, self._dev.dtype)
        entry = [weakref.ref(mirror), gen, None]
        if self._lazy_eager:
            mirror.start_copy(lambda: self._begin_snapshot(*self._lazy_where(0)))
            entry[2] = mirror._pending[0] if mirror._pending else None
        self._lazy_live.append(entry)
        return mirror

    def _peek_lazy_state(self):
        """a mirror of the current state for a callee that may well ignore it (a
        user's `rho_jac(t, y)`): never copied ahead of its first use"""
        eager, self._lazy_eager = self._lazy_eager, False
        try:
            return self._new_lazy_state()
        finally:
            self._lazy_eager = eager

    def _retire_lazy_states(self, everything=False):
        """Before a step starts: the device is about to overwrite the buffer of the
        state before the current one.  Mirrors of it that somebody still holds are
        downloaded now -- and if that took a synchronous copy, the caller evidently
        stores its states: from now on every new mirror starts its copy at once,
        beside the following steps.  Copies under way are waited for either way."""
        if not self._lazy_live:
            return
        keep = []
        for entry in self._lazy_live:
            ref, gen, done = entry
            mirror = ref()
            if not everything and gen >= self._state_gen:
                if mirror is not None and not mirror.materialized:
                    keep.append(entry)
                continue
            if mirror is not None and not mirror.materialized:
                if done is None:
                    self._lazy_eager = True
                mirror.materialize()
            elif done is not None:
                if not done.is_set():
                    # nobody waits for it, but it reads the buffer
                    from .lazy import COPY_TIMEOUT_S
                    if not done.wait(COPY_TIMEOUT_S):
                        raise DeviceError("a state download has not finished after "
                                          f"{COPY_TIMEOUT_S:.0f} s")
                if mirror is None:
                    self._lazy_eager = False      # copied for nobody: stop that
        self._lazy_live = keep



class RungeKutta(_LazyStateMixin, OdeSolver):
    """Base class of the device-resident explicit Runge-Kutta methods.

    Subclasses provide the tableau as class attributes exactly like the
    reference (common.py:88-121): `n_stages, order, order_secondary, A, B, C,
    E` and optionally `P, stbrad, tanang, sc_params`.

    `fun` may be a `DeviceRHS` (state stays in HBM for the whole step) or any
    Python callable (host-RHS mode: the stage argument is downloaded, `fun` is
    called, the derivative is uploaded -- all RK arithmetic still runs on the
    GPU).  Extra keyword arguments beyond the reference's: `device` (GPU
    ordinal) and `lockstep` (a `LockstepGroup`).
    """

    n_stages: int = NotImplemented
    order: int = NotImplemented
    order_secondary: int = NotImplemented
    A: np.ndarray = NotImplemented
    B: np.ndarray = NotImplemented
    C: np.ndarray = NotImplemented
    E: np.ndarray = NotImplemented
    P: np.ndarray = NotImplemented
    stbrad: float = NotImplemented
    tanang: float = NotImplemented
    sc_params = "standard"
    max_factor = MAX_FACTOR0
    min_factor = MIN_FACTOR

    _extra_rows = 0           # BS5 asks for more K rows

    # ------------------------------------------------------------------ ctor
    @_with_esq_options
    def __init__(self, fun, t0, y0, t_bound, max_step=np.inf, rtol=1e-3,
                 atol=1e-6, vectorized=False, first_step=None,
                 nfev_stiff_detect=5000, sc_params=None, support_complex=True,
                 device=0, lockstep=None, **extraneous):
        warn_extraneous(extraneous)
        self._dev = None
        self._y_host = None
        self._device_rhs = fun if isinstance(fun, DeviceRHS) else None
        super().__init__(fun, t0, y0, t_bound, vectorized,
                         support_complex=support_complex)
        self.max_step = validate_max_step(max_step)
        self.rtol, self.atol = validate_tol(rtol, atol, self._y_host)
        self.error_exponent = -1 / (min(self.order_secondary, self.order) + 1)
        self._init_stiffness_detection(nfev_stiff_detect)
        self.h_min_a, self.h_min_b = self._init_min_step_parameters()
        self.tiny_err = self.h_min_b
        self._init_sc_control(sc_params)
        self.FSAL = 1 if self.E[self.n_stages] else 0

        # ---- device state
        y_host = self._y_host
        is_cplx = np.iscomplexobj(y_host)
        if self._device_rhs is not None and is_cplx != self._device_rhs.is_complex:
            raise TypeError('dtypes of solution and derivative do not match')
        # at least 5 K rows: rows 1..4 are the work vectors of the device
        # starting-step estimate (they are free until the first step)
        opts = self._esq_options
        self._dev = DeviceContext(
            self.n, max(self.n_stages + 1 + self._extra_rows, 5), is_cplx, device,
            host_rhs=self._device_rhs is None and lockstep is None, options=opts)
        self._prelaunch = opts.get("prelaunch", "1") != "0"
        self._lib = self._dev.lib
        self._ctx = self._dev.handle
        self._dev.set_tableau(self.A, self.B, self.C, self.E, self.FSAL)
        self._dev.set_tol(self.rtol, self.atol)
        self._dev.upload(SLOT_Y, 0, y_host)
        # classes that take WHOLE steps (the generic `_step_impl`: every accepted
        # step is followed by stages 1 .. s) let the library enqueue the next step's
        # first launch ahead of time; BS5 / CFMR7osc / CKdisc run their stages in
        # pieces and would only repeat it
        # the pairs that test an early error estimate (BS5, CFMR7osc) take whole
        # steps too where the state lives on the device: the estimate rides on the
        # chain sweep that completes its last stage, the rest of the attempt is
        # enqueued behind it as if it had passed, ONE wait per attempt
        # (esq_rk_set_pre; pre_whole=0: the piecewise sequence of round 5)
        pre = self._early_estimate()
        self._pre_whole = (pre is not None and self._device_rhs is not None
                           and not is_cplx and not self._dev.host_slab
                           and opts.get("pre_whole", "1") != "0")
        if self._pre_whole:
            self._dev.rk_set_pre(*pre)
        self.pre_discards = 0        # attempts whose speculative tail was thrown away
        self._launch_ahead = (self._device_rhs is not None
                              and (type(self)._step_impl is RungeKutta._step_impl
                                   or self._pre_whole)
                              and opts.get("launch_ahead", "1") != "0")
        self._dev._chk(self._lib.esq_rk_set_launch_ahead(self._ctx,
                                                         int(self._launch_ahead)),
                       "esq_rk_set_launch_ahead")
        self._lockstep = lockstep
        self._n_norm = self.n
        if lockstep is not None:
            self._dev._chk(self._lib.esq_set_comm(self._ctx, lockstep.comm),
                           "esq_set_comm")
            lockstep.attach(self._dev)
            self._n_norm = lockstep.n_total
        self._f_host = None
        self._K_host = None
        self._y_old_host = None
        self._lazy_init(y_host.nbytes, self._dev.host_slab)
        if self._device_rhs is not None:
            self._dev.set_rhs(self._device_rhs)
            self._chk(self._lib.esq_rk_eval_rhs(self._ctx, 0, float(self.t),
                                                SLOT_Y, 0), "esq_rk_eval_rhs")
            self.nfev += 1
        else:
            f0 = self.fun(self.t, y_host)
            if f0.dtype != y_host.dtype:
                raise TypeError('dtypes of solution and derivative do not match')
            self._dev.upload(SLOT_K, 0, f0)
            self._f_host = f0

        # ---- first step
        if first_step is None:
            b = self.t + self.direction * min(abs(self.t_bound - self.t),
                                              self.max_step)
            self.h_abs = abs(self._device_h_start(b))
        else:
            self.h_abs = validate_first_step(first_step, t0, t_bound)
        self.h_previous = None
        self.error_norm_old = None
        NFS[()] = 0

    def _chk(self, code, what):
        try:
            self._dev._chk(code, what)
        except Exception:
            if getattr(self, "_lockstep", None) is not None:
                self._lockstep.sync_aborted()
            raise

    def _group_reduce(self, values, op="sum"):
        """shard-local scalars from the library -> scalars of the whole batch
        (identity outside a lock-step group and on the RCCL path, where the
        library has reduced them already)"""
        grp = getattr(self, "_lockstep", None)
        if grp is None:
            return list(values)
        return grp.host_reduce(self._dev, values, op)

    # ------------------------------------------------------ starting step
    def _device_h_start(self, b):
        """Watts' starting step (ref `h_start`, common.py:519-763) with every
        vector in HBM: the perturbation vectors live in K rows 1..4, norms come
        back as one double each, the scalar decisions below are the host part.
        tests/test_gpu_parity.py::test_device_h_start pins it to the oracle's
        `first_step_size` (itself pinned to the reference's golden first steps)."""
        import ctypes
        if self.n == 0:
            return np.inf
        lib, ctx = self._lib, self._ctx
        Y, F0, SF, YP, PV, SPY = VEC_Y, 0, 1, 2, 3, 4
        neq = self._n_norm
        dtype = self._dev.dtype

        def sumsq(x, y=VEC_NONE):
            out = ctypes.c_double()
            self._chk(lib.esq_vec_sumsq(ctx, x, y, ctypes.byref(out)),
                      "esq_vec_sumsq")
            return self._group_reduce([out.value], "sum")[0]

        def rms(x, y=VEC_NONE):
            return (sumsq(x, y) / neq) ** 0.5

        def axpbmc(dst, a, alpha, bb, c=VEC_NONE):
            self._chk(lib.esq_vec_axpbmc(ctx, dst, a, float(alpha), bb, c),
                      "esq_vec_axpbmc")

        def rhs(dst, t, src):
            if self._device_rhs is not None:
                self._chk(lib.esq_vec_eval_rhs(ctx, dst, float(t), src),
                          "esq_vec_eval_rhs")
                self.nfev += 1
                return
            arg = np.empty(self.n, dtype=dtype)
            self._chk(lib.esq_vec_download(ctx, src, as_ptr(arg)),
                      "esq_vec_download")
            val = np.ascontiguousarray(self.fun(t, arg), dtype=dtype)
            self._chk(lib.esq_vec_upload(ctx, dst, as_ptr(val)),
                      "esq_vec_upload")

        fi = np.finfo(np.float64)
        big = sqrt(fi.max)
        small = np.nextafter(fi.epsneg, 1.0)
        relper = small ** 0.375
        a = self.t
        dx = b - a
        absdx = abs(dx)

        # (1) t-derivative bound and |f| bound
        da = copysign(max(min(relper * abs(a), absdx), 100. * small * abs(a)), dx)
        if da == 0.0:
            da = relper * dx
        rhs(SF, a + da, Y)
        delf = rms(SF, F0)
        dfdxb = delf / abs(da) if delf < big * abs(da) else big
        fbnd = rms(SF)

        # (2) Lipschitz estimate
        dely = relper * rms(Y)
        if dely == 0.0:
            dely = relper
        dely = copysign(dely, dx)
        delf = rms(F0)
        fbnd = max(fbnd, delf)
        if delf:
            self._chk(lib.esq_vec_copy(ctx, SPY, F0), "esq_vec_copy")
            self._chk(lib.esq_vec_copy(ctx, YP, F0), "esq_vec_copy")
        else:
            self._chk(lib.esq_vec_fill(ctx, SPY, 0.0, 0.0), "esq_vec_fill")
            self._chk(lib.esq_vec_fill(ctx, YP, 1.0, 0.0), "esq_vec_fill")
            delf = rms(YP)
        dfdub = 0.0
        n_iter = min(neq + 1, 3)
        for k in range(1, n_iter + 1):
            axpbmc(PV, Y, dely / delf, YP)
            if k == 2:
                rhs(YP, a + da, PV)
                axpbmc(PV, VEC_NONE, 1.0, YP, SF)
            else:
                rhs(YP, a, PV)
                axpbmc(PV, VEC_NONE, 1.0, YP, F0)
            fbnd = max(fbnd, rms(YP))
            delf = rms(PV)
            if delf >= big * abs(dely):
                dfdub = big
                break
            dfdub = max(dfdub, delf / abs(dely))
            if k == n_iter:
                break
            if delf == 0.0:
                delf = 1.0
            src, fill = (Y, dely / relper) if k == 2 else (PV, delf)
            self._chk(lib.esq_hs_select(ctx, YP, SPY, src, float(fill)),
                      "esq_hs_select")
            delf = rms(YP)

        # (3) step from the bounds and the tolerances
        ydpb = dfdxb + dfdub * fbnd
        tolsum, tolmin = ctypes.c_double(), ctypes.c_double()
        self._chk(lib.esq_hs_log_etol(ctx, Y, ctypes.byref(tolsum),
                                      ctypes.byref(tolmin)), "esq_hs_log_etol")
        tol_sum = self._group_reduce([tolsum.value], "sum")[0]
        tol_min = self._group_reduce([tolmin.value], "min")[0]
        tolp = 10.0 ** (0.5 * (tol_sum / neq + min(tol_min, big))
                        / (self.order_secondary + 1))
        h = absdx
        if ydpb == 0.0 and fbnd == 0.0:
            if tolp < 1.0:
                h = absdx * tolp
        elif ydpb == 0.0:
            if tolp < fbnd * absdx:
                h = tolp / fbnd
        else:
            srydpb = sqrt(0.5 * ydpb)
            if tolp < srydpb * absdx:
                h = tolp / srydpb
        if dfdub:
            h = min(h, 1.0 / dfdub)
        h = max(h, 100.0 * small * abs(a))
        if h == 0.0:
            h = small * abs(b)
        return copysign(h, dx)

    # ------------------------------------------------- lazy host mirrors
    @property
    def y(self):
        """current state; downloaded from HBM on first access after a step
        (a fresh array each step, as scipy stores it by reference).  A large
        device-resident state comes back as a `LazyState` (lazy.py): an array-like
        that downloads when it is really used -- plain `solve_ivp` reads `solver.y`
        after every step (ivp.py:665) whether it needs it or not"""
        return self._lazy_y()

    @y.setter
    def y(self, value):
        if self._dev is not None and value is not None:
            # mirrors of the state that is about to be replaced: download first
            self._retire_lazy_states(everything=True)
            if isinstance(value, LazyState):
                value = value.materialize()
        self._y_host = value
        if self._dev is not None and value is not None:
            self._dev.upload(SLOT_Y, 0, value)

    def _lazy_where(self, age):
        return (SLOT_Y, 0) if age == 0 else (SLOT_YNEW, 0)

    def step(self):
        self._retire_lazy_states()
        if _dense_dead:
            drain_dense()
        return super().step()

    @property
    def f(self):
        """derivative at (t, y): logical K row 0"""
        if self._f_host is None:
            self._f_host = self._dev.download(SLOT_K, 0)
        return self._f_host

    @f.setter
    def f(self, value):
        self._f_host = value

    @property
    def K(self):
        """stage derivatives of the last accepted step, shape (s+1, n)"""
        if self._K_host is None:
            rows = self.n_stages + 1
            K = np.empty((rows, self.n), dtype=self._dev.dtype)
            for r in range(rows):
                K[r] = self._dev.download_last_K(r)
            self._K_host = K
        return self._K_host

    @property
    def y_old(self):
        if self.t_old is None:
            return None
        if self._y_old_host is None:
            # after the accept swap the previous state sits in the YNEW slot
            self._y_old_host = self._dev.download(SLOT_YNEW)
        return self._y_old_host

    @y_old.setter
    def y_old(self, value):
        self._y_old_host = value

    @property
    def f_old(self):
        return self._dev.download_last_K(0)

    def _invalidate_mirrors(self):
        self._y_host = None
        self._f_host = None
        self._K_host = None
        self._y_old_host = None

    # ------------------------------------------------------- initialisation
    def _init_min_step_parameters(self):
        """min_step = max(h_min_a*(|t|+h), h_min_b), RKSuite's rule with the
        smallest gap between distinct abscissae (ref common.py:123-148)."""
        c = np.unique(np.asarray(self.C, dtype=float))
        gaps = np.diff(c)
        cdiff = min(1.0, gaps.min()) if gaps.size else 1.0
        if cdiff < 1e-3:
            cdiff = 1e-3
            logging.warning(
                'Some C-values of this Runge Kutta method are nearly the same '
                'but not identical. This limits the minimum stepsize. You may '
                'want to check the implementation of this method.')
        fi = np.finfo(self._y_host.dtype)
        return 10 * fi.epsneg / cdiff, sqrt(fi.tiny)

    def _init_stiffness_detection(self, nfev_stiff_detect):
        if not (isinstance(nfev_stiff_detect, int) and nfev_stiff_detect >= 0):
            raise ValueError(
                "`nfev_stiff_detect` must be a non-negative integer.")
        self.nfev_stiff_detect = nfev_stiff_detect
        if NotImplemented in (self.stbrad, self.tanang):
            if nfev_stiff_detect not in (5000, 0):
                warn("This method does not implement stiffness detection. "
                     "Changing the value of nfev_stiff_detect does nothing.")
            self.nfev_stiff_detect = 0
        self.jflstp = 0
        self.okstp = 0
        self.havg = 0.0

    def _init_sc_control(self, sc_params):
        """(k*b1, k*b2, a2, g) of the PI-like controller, ref common.py:166-185"""
        spec = sc_params or self.sc_params
        if isinstance(spec, str) and spec in _SC_PRESETS:
            kb1, kb2, a, g = _SC_PRESETS[spec]
        elif isinstance(spec, tuple) and len(spec) == 4:
            kb1, kb2, a, g = spec
        else:
            raise ValueError('sc_params should be a tuple of length 4 or one '
                             'of the strings "G", "S", "W" or "standard"')
        self.minbeta1 = kb1 * self.error_exponent
        self.minbeta2 = kb2 * self.error_exponent
        self.minalpha = -a
        self.safety = g
        self.safety_sc = g ** (kb1 + kb2)
        self.standard_sc = True

    # ------------------------------------------------------- host controller
    def _limit_step(self, t, h_abs):
        """clip to [min_step, max_step]; look ahead over the last two steps
        (ref common.py:310-331).  Pure: returns (h_abs, min_step, reset_sc)."""
        reset = False
        min_step = max(self.h_min_a * (abs(t) + h_abs), self.h_min_b)
        if not (min_step <= h_abs <= self.max_step):
            h_abs = min(self.max_step, max(min_step, h_abs))
            reset = True
        remaining = abs(self.t_bound - t)
        if remaining < 2 * h_abs:
            if remaining > h_abs:
                h_abs = max(0.5 * remaining, min_step)
                reset = True
            else:
                h_abs = remaining
        return h_abs, min_step, reset

    def _reassess_stepsize(self, t, y=None):
        h_abs, min_step, reset = self._limit_step(t, self.h_abs)
        if reset:
            self.standard_sc = True
        return h_abs, min_step

    def _accept_factor(self, error_norm, h, rejected_before):
        """step-size factor after an accepted attempt (ref common.py:252-276)"""
        if error_norm < self.tiny_err:
            factor = self.max_factor
            self.standard_sc = True
        elif self.standard_sc:
            factor = self.safety * error_norm ** self.error_exponent
            self.standard_sc = False
        else:
            factor = self.safety_sc * (
                error_norm ** self.minbeta1
                * self.error_norm_old ** self.minbeta2
                * (h / self.h_previous) ** self.minalpha)
            factor = min(self.max_factor, max(self.min_factor, factor))
        if rejected_before:
            factor = min(1, factor)
        if factor < MAX_FACTOR:
            self.max_factor = MAX_FACTOR
        return factor

    def _reject_factor(self, error_norm):
        return max(self.min_factor,
                   self.safety * error_norm ** self.error_exponent)

    def _rms_from_sumsq(self, sumsq):
        # with an RCCL communicator the library has summed over the ranks
        # already; a host reducer (several solvers of one process driven in
        # lock-step, tests) sums here
        sumsq = self._group_reduce([sumsq], "sum")[0]
        return (sumsq / self._n_norm) ** 0.5 if self._n_norm else np.nan

    # ------------------------------------------------------- device launches
    def _run_stages(self, i_from, i_to, t, h):
        """stages i_from .. i_to-1 (ref common.py:241-242, 353-356)"""
        if self._device_rhs is not None:
            self._chk(self._lib.esq_rk_stages(self._ctx, i_from, i_to, t, h),
                      "esq_rk_stages")
            self.nfev += i_to - i_from
            return
        for i in range(i_from, i_to):
            self._chk(self._lib.esq_rk_stage_accumulate(self._ctx, i, h),
                      "esq_rk_stage_accumulate")
            y_stage = self._dev.download(SLOT_YSTAGE)
            self._dev.upload(SLOT_K, i, self.fun(t + self.C[i] * h, y_stage))

    def _attempt(self, t, h, h_next=0.0, want_pre=False):
        """stages 1 .. s - 1, the solution and the error estimate(s) of one attempt with
        a device RHS, ONE call into the library: -> (sumsq, pre_sumsq or None); counts the
        evaluations of a full attempt"""
        out, pre = ctypes.c_double(), ctypes.c_double()
        self._chk(self._lib.esq_rk_attempt(self._ctx, t, h, h_next, ctypes.byref(out),
                                           ctypes.byref(pre) if want_pre else None),
                  "esq_rk_attempt")
        self.nfev += self.n_stages - 1 + self.FSAL
        return out.value, (pre.value if want_pre else None)

    def _solution_and_error(self, t, h, h_next=0.0):
        """`_comp_sol_err` (ref common.py:341-351): returns the error norm,
        leaves y_new in the YNEW slot.  `h_next`: the next step size if this
        attempt is accepted, where the caller knows it already (`_step_impl`)"""
        if self._device_rhs is not None:
            sumsq = self._dev.rk_solution_error_sumsq(t, h, h_next)
            self.nfev += self.FSAL
        elif self.FSAL:
            self._chk(self._lib.esq_rk_solution(self._ctx, h), "esq_rk_solution")
            y_new = self._dev.download(SLOT_YNEW)
            self._dev.upload(SLOT_K, self.n_stages, self.fun(t + h, y_new))
            sumsq = self._dev.rk_error_norm_sumsq(h)
        else:
            sumsq = self._dev.rk_solution_error_sumsq(t, h)
        return self._rms_from_sumsq(sumsq)

    def _finish_step(self, t_new, h, h_abs_next=None):
        """end-point derivative of non-FSAL pairs, then pointer rotation on the
        device (ref common.py:289-303).  `h_abs_next`: the step size the
        controller has just chosen; the library then forms the NEXT step's
        first stage argument right away (in the end-point sweep itself, or in a
        kernel that runs while this host code is between steps).  The next
        `_run_stages` uses it only if it asks for exactly that step."""
        end_eval = 0
        h_next = 0.0
        if self._device_rhs is not None:
            if not self.FSAL:
                end_eval = 1
                self.nfev += 1
            if h_abs_next is not None and self._prelaunch:
                h_lim, min_step, _ = self._limit_step(t_new, h_abs_next)
                if h_lim >= min_step:
                    h_next = h_lim * self.direction
        elif not self.FSAL:
            y_new = self._dev.download(SLOT_YNEW)
            self._dev.upload(SLOT_K, self.n_stages, self.fun(t_new, y_new))
        self._chk(self._lib.esq_rk_accept(self._ctx, t_new, end_eval, h_next),
                  "esq_rk_accept")
        self._state_gen += 1
        self._invalidate_mirrors()

    def _guess_next_step(self, t_new, h_abs):
        """The next step size, known BEFORE the error norm of this attempt: a run
        that sits at `max_step` keeps its step when the attempt is accepted with a
        factor >= 1 (the usual case there) -- the library then enqueues the next
        step's first launch behind the error norm instead of after the host has
        digested it (esq_rk_solution_error_ahead).  A wrong guess (a rejection, a
        factor < 1) costs one discarded launch; 0.0: no guess."""
        if not self._launch_ahead or h_abs < self.max_step:
            return 0.0
        h_lim, min_step, _ = self._limit_step(t_new, h_abs)
        return h_lim * self.direction if h_lim >= min_step else 0.0

    # ------------------------------------- pairs with an early error estimate
    def _early_estimate(self):
        """(e_pre, b_scale_pre) -- weights over K[:p] of an error estimate that is
        tested after stage p - 1, before the step's last stage(s) -- or None"""
        return None

    def _step_impl_early(self, nan_check_first):
        """`_step_impl` of BS5 (ref bogacki.py:238-338) and CFMR7osc (calvo.py:152-253):
        the generic step with an early rejection test after stage s - 2.  Device-resident
        states take the attempt WHOLE (`_pre_whole`): stages, early estimate, last stage
        and final estimate are enqueued in one go, the host waits once and then looks
        at the early estimate first -- exactly the decisions of the piecewise sequence
        (a rejected attempt's speculative tail is counted in `pre_discards`, not in
        `nfev`).  `nan_check_first`: BS5 tests for NaN before the rejection bookkeeping,
        CFMR7osc after it."""
        t = self.t
        s = self.n_stages
        h_abs, min_step = self._reassess_stepsize(t)
        rejected = False
        while True:
            if h_abs < min_step:
                return False, self.TOO_SMALL_STEP
            h = h_abs * self.direction
            t_new = t + h
            if self._pre_whole:
                if self._lockstep is not None:
                    self._lockstep.check_identical(self._dev, "(t, h)", (t, h))
                sumsq, pre_sq = self._attempt(t, h, self._guess_next_step(t_new, h_abs),
                                              want_pre=True)
                pre = self._rms_from_sumsq(pre_sq)
                if not pre > 1:
                    error_norm = self._rms_from_sumsq(sumsq)
                else:
                    # (the speculative tail: thrown away, not counted)
                    self.nfev -= 1 + self.FSAL
                    self.pre_discards += 1
            else:
                self._run_stages(1, s - 1, t, h)
                pre = self._estimate_error_norm_pre(None, h)
            if pre > 1:
                # early rejection: the last evaluation(s) are saved
                rejected = True
                h_abs *= self._reject_factor(pre)
                NFS[()] += 1
                if self.nfev_stiff_detect:
                    self.jflstp += 1
                continue
            if not self._pre_whole:
                self._run_stages(s - 1, s, t, h)
                error_norm = self._solution_and_error(t, h)
            if error_norm < 1:
                h_abs *= self._accept_factor(error_norm, h, rejected)
                break
            bad = np.isnan(error_norm) or np.isinf(error_norm)
            if bad and nan_check_first:
                return False, "Overflow or underflow encountered."
            rejected = True
            h_abs *= self._reject_factor(error_norm)
            NFS[()] += 1
            self.jflstp += 1
            if bad:
                return False, "Overflow or underflow encountered."
        self._finish_step(t_new, h, h_abs)
        self.h_previous = h
        self.h_abs = h_abs
        self.error_norm_old = error_norm
        self.t = t_new
        self._diagnose_stiffness()
        return True, None

    # ----------------------------------------------------------------- step
    def _step_impl(self):
        t = self.t
        h_abs, min_step = self._reassess_stepsize(t)
        rejected = False
        while True:
            if h_abs < min_step:
                return False, self.TOO_SMALL_STEP
            h = h_abs * self.direction
            t_new = t + h
            if self._lockstep is not None:
                self._lockstep.check_identical(self._dev, "(t, h)", (t, h))
            if self._device_rhs is not None:
                # the whole attempt in one call (esq_rk_attempt)
                error_norm = self._rms_from_sumsq(
                    self._attempt(t, h, self._guess_next_step(t_new, h_abs))[0])
            else:
                self._run_stages(1, self.n_stages, t, h)
                error_norm = self._solution_and_error(t, h)
            if error_norm < 1:
                h_abs *= self._accept_factor(error_norm, h, rejected)
                break
            rejected = True
            h_abs *= self._reject_factor(error_norm)
            NFS[()] += 1
            self.jflstp += 1
            if np.isnan(error_norm) or np.isinf(error_norm):
                return False, "Overflow or underflow encountered."
        self._finish_step(t_new, h, h_abs)
        self.h_previous = h
        self.h_abs = h_abs
        self.error_norm_old = error_norm
        self.t = t_new
        self._diagnose_stiffness()
        return True, None

    # ------------------------------------------------- public-ish helpers
    def _estimate_error(self, K, h):
        """h * (K.T @ E) as a host vector (ref common.py:333-336), evaluated on
        the device.  `K` is normally `self.K`; any other (s+1, n) array is
        pushed through a scratch context."""
        if K is self._K_host and K is not None:
            self._chk(self._lib.esq_rk_error_vector(self._ctx, float(h), 1),
                      "esq_rk_error_vector")
            return self._dev.download(SLOT_WORK)
        K = np.asarray(K)
        tmp = DeviceContext(self.n, self.n_stages + 1,
                            np.iscomplexobj(K), self._dev.device)
        try:
            tmp.set_tableau(self.A, self.B, self.C, self.E, self.FSAL)
            for r in range(self.n_stages + self.FSAL):
                tmp.upload(SLOT_K, r, K[r])
            tmp._chk(tmp.lib.esq_rk_error_vector(tmp.handle, float(h), 0),
                     "esq_rk_error_vector")
            return tmp.download(SLOT_WORK)
        finally:
            tmp.close()

    def _estimate_error_norm(self, K, h, scale):
        return norm(self._estimate_error(K, h) / scale)

    # ---------------------------------------------------------- dense output
    # below this size a host interpolant (one download of Qh) answers the many
    # tiny evaluations of event root-finding faster than kernel launches do
    _DEVICE_DENSE_MIN_N = 4096

    def _horner_interpolant(self, P, t_a, t_b, from_end=False):
        """interpolant over [t_a, t_b] from Q = K_last.T @ P (ref common.py:
        361-364), built on the device in one pass; `from_end`: anchored at the
        current state (BS5 'best'), else at the pre-step state"""
        import ctypes
        P = np.ascontiguousarray(P, dtype=np.float64)
        drain_dense()
        handle = ctypes.c_void_p()
        self._chk(self._lib.esq_dense_create(
            self._ctx, as_ptr(P), P.shape[0], P.shape[1], float(t_b - t_a),
            int(from_end), ctypes.byref(handle)), "esq_dense_create")
        if self.n >= self._DEVICE_DENSE_MIN_N:
            return DeviceHornerDenseOutput(t_a, t_b, self._lib, handle, self.n,
                                           self._dev.dtype)
        try:
            Qh = np.empty((P.shape[1], self.n), dtype=self._dev.dtype)
            self._chk(self._lib.esq_dense_download(handle, as_ptr(Qh)),
                      "esq_dense_download")
        finally:
            self._lib.esq_dense_destroy(handle)
        base = self.y if from_end else self.y_old
        return HornerDenseOutput(t_a, t_b, base, Qh.T, scaled=True)

    def _dense_output_impl(self):
        if isinstance(self.P, np.ndarray):
            return self._horner_interpolant(self.P, self.t_old, self.t)
        if self.n >= self._DEVICE_DENSE_MIN_N:
            # tableaux without P: the cubic Hermite interpolant (ref common.py:366-368,
            # 793-821) as a device-resident Horner form -- nothing is copied to the
            # host until it is evaluated
            rid = self._lib.esq_rk_row_id
            f_old, f = rid(self._ctx, 0, 1), rid(self._ctx, 0, 0)
            if f_old < 0 or f < 0:
                raise DeviceError(f"esq_rk_row_id failed with code {min(f_old, f)}")
            # (after the accept Y is the new state, YNEW the pre-step one)
            return _cubic_interpolant(self, self.t_old, self.t, VEC_YNEW, VEC_Y, f_old, f)
        return CubicDenseOutput(self.t_old, self.t, self.y_old, self.y,
                                self.f_old, self.f)

    # ---------------------------------------------------- stiffness detection
    def _diagnose_stiffness(self):
        """RKSuite's stiffness check (ref common.py:370-516): test after every
        `nfev_stiff_detect` evaluations' worth of steps or after >= 10 failed
        steps within 40; the diagnosis itself runs on the device
        (extensisq_amd/stiffness.py)."""
        if self.nfev_stiff_detect == 0:
            return
        self.okstp += 1
        h = self.h_previous
        self.havg = 0.9 * self.havg + 0.1 * h
        if self.okstp == 20:
            self.havg = h
            self.jflstp = 0
        lotsfl = False
        if self.okstp % 40 == 39:
            lotsfl = self.jflstp >= 10
            self.jflstp = 0
        many_steps = self.nfev_stiff_detect // self.n_stages
        toomch = self.okstp % many_steps == many_steps - 1
        if toomch or lotsfl:
            from .stiffness import diagnose
            diagnose(self, lotsfl)

    def __del__(self):
        # (never a blocking call in a finalizer: DeviceContext.park)
        dev = getattr(self, "_dev", None)
        if dev is not None:
            dev.park()
