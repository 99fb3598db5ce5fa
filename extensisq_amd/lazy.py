"""Deferred host mirrors of large device-resident states.

scipy's `solve_ivp` reads `solver.y` after EVERY step (ivp.py:665) and, when
`t_eval is None`, keeps what it got (ivp.py:702); with `t_eval` / `dense_output`
and no events it never looks at it.  At n = 1e7 a state is 80 MB -- a 1.4 ms copy
over PCIe against a 0.47 ms step -- so `solver.y` of a large device-RHS solver
returns a `LazyState`: an array-like that downloads the state when somebody
really uses it (`np.asarray`, indexing, arithmetic, any NumPy function), and not at
all otherwise.

The device keeps a state for exactly one more accepted step (after the accept it is
the "previous state" the dense output starts from; the step after that overwrites
its buffer), so the solver retires live mirrors at the start of `step()`:
a mirror that is still referenced by then -- a caller that stores the states, like
plain `solve_ivp` -- is downloaded before its source goes, and from then on every
new mirror starts its copy at once on the process's download stream (`esq_snapshot_*`,
a copy worker thread; csrc/esq_core.hip `lane_copy`): the 1.4 ms copy of state k runs beside steps
k + 1 and k + 2 instead of in front of them.  Each mirror owns a fresh host array (scipy stores
them by reference: they must be distinct arrays, as in the reference, common.py:343).
"""
import queue
import threading

import numpy as np
from numpy.lib.mixins import NDArrayOperatorsMixin


COPY_TIMEOUT_S = 300.0          # a download beside the step takes milliseconds


class CopyWorker:
    """one daemon thread that runs `esq_snapshot_copy` calls (ctypes releases the
    GIL: the solver's thread goes on stepping)"""

    def __init__(self):
        self._q = queue.Queue()
        self._thread = None
        self._lock = threading.Lock()
        self._idle = threading.Condition(self._lock)
        self._inflight = 0

    def _run(self):
        while True:
            job = self._q.get()
            fn, done, box = job
            try:
                box.append(fn())
            except BaseException as exc:          # noqa: BLE001 - handed to the waiter
                box.append(exc)
            done.set()
            with self._idle:
                self._inflight -= 1
                self._idle.notify_all()

    def wait_idle(self, timeout=60.0):
        """every copy handed in so far has finished (a context must not be destroyed
        under a copy that reads from it: device.py)"""
        with self._idle:
            return self._idle.wait_for(lambda: self._inflight == 0, timeout)

    def submit(self, fn):
        """-> (event, box): box[0] is fn()'s result (or the exception it raised)
        once the event is set"""
        done, box = threading.Event(), []
        with self._lock:
            if self._thread is None:
                self._thread = threading.Thread(target=self._run, daemon=True,
                                                name="esq-copy-worker")
                self._thread.start()
            # counted and queued in one piece (the queue is unbounded: no wait here)
            self._q.put((fn, done, box))
            self._inflight += 1
        return done, box


_worker = CopyWorker()


def _plain(x):
    if isinstance(x, LazyState):
        return x.materialize()
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    return x


class LazyState(NDArrayOperatorsMixin):
    """Array-like stand-in for a state vector that still lives on the device.
    `shape`, `dtype`, `size`, `ndim`, `len()` cost nothing; everything else
    downloads the vector (once) and behaves like the ndarray it then holds."""

    __array_priority__ = 1000

    def __init__(self, fetch, n, dtype):
        self._fetch = fetch            # () -> ndarray: the synchronous download
        self._arr = None
        self._pending = None           # (event, box, out) of a copy under way
        self.shape = (int(n),)
        self.dtype = np.dtype(dtype)
        self.ndim = 1
        self.size = int(n)

    # -- state
    @property
    def materialized(self):
        return self._arr is not None

    def start_copy(self, begin):
        """`begin() -> (copy_fn, out)`: put the download on the copy worker now"""
        if self._arr is not None or self._pending is not None:
            return
        copy_fn, out = begin()
        done, box = _worker.submit(copy_fn)
        self._pending = (done, box, out)

    def materialize(self):
        if self._arr is None:
            if self._pending is not None:
                done, box, out = self._pending
                if not done.wait(COPY_TIMEOUT_S):
                    raise RuntimeError(
                        f"the download of a state has not finished after {COPY_TIMEOUT_S:.0f} s "
                        "(copy worker stuck on the device?)")
                self._pending = None
                if box and isinstance(box[0], BaseException):
                    raise box[0]
                self._arr = out
            else:
                self._arr = self._fetch()
            self._fetch = None
        return self._arr

    # -- the array protocols
    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if "out" in kwargs:
            kwargs["out"] = _plain(kwargs["out"])
        return getattr(ufunc, method)(*_plain(inputs), **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        return func(*_plain(args), **_plain(kwargs))

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, key):
        return self.materialize()[key]

    def __setitem__(self, key, value):
        self.materialize()[key] = value

    def __getattr__(self, name):
        # (only reached for what is not defined above: T, real, copy, sum, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __repr__(self):
        if self._arr is None:
            return f"LazyState(shape={self.shape}, dtype={self.dtype}, on device)"
        return repr(self._arr)

    def __float__(self):
        return float(self.materialize())

    def __bool__(self):
        return bool(self.materialize())
