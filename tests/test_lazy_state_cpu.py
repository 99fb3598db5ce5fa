"""extensisq_amd/lazy.py without a GPU: the array-like that `solver.y` returns for a
large device-resident state downloads on first real use, once, and then behaves like
the ndarray it holds (what scipy's solve_ivp and user event functions do with it)."""
import threading

import numpy as np
import pytest
from numpy.testing import assert_array_equal

from extensisq_amd.lazy import CopyWorker, LazyState


def make(n=7):
    calls = []
    data = np.linspace(-1.0, 2.0, n)

    def fetch():
        calls.append(1)
        return data.copy()
    return LazyState(fetch, n, np.float64), data, calls


def test_metadata_costs_nothing():
    m, data, calls = make()
    assert m.shape == (7,) and m.size == 7 and m.ndim == 1 and len(m) == 7
    assert m.dtype == np.float64 and not m.materialized
    assert "on device" in repr(m)
    assert calls == []


def test_first_real_use_downloads_once():
    m, data, calls = make()
    assert_array_equal(np.asarray(m), data)
    assert_array_equal(np.asarray(m), data)
    assert calls == [1] and m.materialized
    assert np.asarray(m) is m.materialize()              # no second copy
    assert np.array(m, copy=True) is not m.materialize()


@pytest.mark.parametrize("use", [
    lambda m: m[2], lambda m: m[1:4], lambda m: m + 1.0, lambda m: 2.0 * m, lambda m: m @ m,
    lambda m: -m, lambda m: m < 0.5, lambda m: np.sin(m), lambda m: np.linalg.norm(m),
    lambda m: np.vstack([m, m]).T, lambda m: np.concatenate([m, m]), lambda m: m.copy(),
    lambda m: m.sum(), lambda m: m.T, lambda m: list(m), lambda m: np.maximum(m, 0.0),
    lambda m: np.add(m, m, out=np.empty(7)), lambda m: float(m[0]), lambda m: m.astype(np.float32),
    lambda m: np.asarray(m, dtype=complex), lambda m: m[:, np.newaxis] * np.ones(3)])
def test_behaves_like_the_array(use):
    m, data, calls = make()
    got, want = use(m), use(data)
    assert calls == [1]
    assert type(got) is type(want) or np.isscalar(want)
    assert_array_equal(np.asarray(got), np.asarray(want))


def test_what_solve_ivp_does_with_stored_states():
    """ivp.py:702, 734: `ys.append(y)` every step, `np.vstack(ys).T` at the end"""
    ms = [make(5) for _ in range(4)]
    ys = [m for m, _d, _c in ms]
    out = np.vstack(ys).T
    assert out.shape == (5, 4)
    for k, (_m, data, calls) in enumerate(ms):
        assert_array_equal(out[:, k], data)
        assert calls == [1]


def test_setitem_writes_the_host_copy():
    m, data, calls = make()
    m[0] = 5.0
    assert m[0] == 5.0 and calls == [1]


def test_copy_under_way_is_waited_for():
    gate = threading.Event()
    out = np.zeros(4)

    def begin():
        def copy_fn():
            gate.wait(5.0)
            out[:] = [1.0, 2.0, 3.0, 4.0]
            return out
        return copy_fn, out
    m = LazyState(lambda: (_ for _ in ()).throw(AssertionError("not the sync path")), 4,
                  np.float64)
    m.start_copy(begin)
    assert not m.materialized
    threading.Timer(0.05, gate.set).start()
    assert_array_equal(np.asarray(m), [1.0, 2.0, 3.0, 4.0])
    assert np.asarray(m) is out


def test_a_failed_copy_raises_in_the_reader():
    def begin():
        def copy_fn():
            raise RuntimeError("device lost")
        return copy_fn, np.zeros(2)
    m = LazyState(lambda: np.zeros(2), 2, np.float64)
    m.start_copy(begin)
    with pytest.raises(RuntimeError, match="device lost"):
        np.asarray(m)


def test_copy_worker_runs_jobs_in_order():
    w = CopyWorker()
    seen = []
    jobs = [w.submit(lambda k=k: seen.append(k) or k) for k in range(5)]
    for done, _box in jobs:
        assert done.wait(5.0)
    assert seen == list(range(5)) and [b[0] for _d, b in jobs] == list(range(5))
    assert w.wait_idle(5.0)
    gate = threading.Event()
    w.submit(lambda: gate.wait(5.0))
    assert not w.wait_idle(0.05)                 # a copy under way: a context must wait
    gate.set()
    assert w.wait_idle(5.0)


# ---------------------------------------------------------------------------
# finalizers: the collector runs them inside ANY allocation of ANY thread -- e.g.
# between the two halves of CopyWorker.submit (the job counted, not yet queued), where
# a wait for the copy worker waits for a job its own caller has not handed in yet
# (60 s per occurrence on the GPU box before this was fixed)
# ---------------------------------------------------------------------------
class _FakeLib:
    def __init__(self):
        self.destroyed, self.freed = [], []

    def esq_destroy(self, handle):
        self.destroyed.append(handle)

    def esq_rhs_free(self, user):
        self.freed.append(user)


def _fake_context(lib, handle):
    from extensisq_amd.device import DeviceContext
    ctx = object.__new__(DeviceContext)
    ctx.lib, ctx.handle = lib, handle
    return ctx


def test_finalizers_park_and_never_wait(monkeypatch):
    import gc
    from extensisq_amd import device, lazy
    from extensisq_amd.common import RungeKutta
    from extensisq_amd.sommeijer import SSV2stab
    waits = []
    monkeypatch.setattr(lazy._worker, "wait_idle", lambda timeout=60.0: waits.append(timeout) or True)
    device._drain_graveyard()
    waits.clear()
    lib = _FakeLib()
    for k, cls in enumerate((RungeKutta, SSV2stab)):
        solver = object.__new__(cls)
        solver._dev = _fake_context(lib, 100 + k)
        solver.__del__()                                   # what the collector calls
        assert solver._dev.handle is None
    ctx = _fake_context(lib, 200)
    del ctx
    gc.collect()
    # nothing destroyed, nobody waited for: three handles parked
    assert lib.destroyed == [] and waits == []
    assert sorted(h for _, h in device._graveyard) == [100, 101, 200]
    # ... until the next well-defined point (a context is made / closed, exit)
    device._drain_graveyard()
    assert sorted(lib.destroyed) == [100, 101, 200] and len(waits) == 1
    assert len(device._graveyard) == 0
    device._drain_graveyard()                              # (nothing parked: no wait)
    assert len(waits) == 1


def test_plugin_finalizer_frees_only_what_it_owns(monkeypatch):
    from extensisq_amd import _lib, device, lazy
    monkeypatch.setattr(lazy._worker, "wait_idle", lambda timeout=60.0: True)
    lib = _FakeLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    device._drain_graveyard()
    owned = object.__new__(device.Heat2D)
    owned._bound, owned._host_ctx = {0: ("fn", 11)}, {0: _fake_context(lib, 300)}
    users = object.__new__(device.CFunctionRHS)
    users._bound, users._host_ctx = {0: ("fn", 12)}, {}
    owned.__del__()
    users.__del__()
    assert lib.freed == [] and lib.destroyed == []         # parked, not freed in the finalizer
    device._drain_graveyard()
    assert lib.freed == [11] and lib.destroyed == [300]    # the caller's pointer (12) is the caller's


def test_a_collection_inside_submit_does_not_stall(monkeypatch):
    """the scenario itself: a solver is collected while CopyWorker.submit has counted its
    job but not queued it"""
    import time
    from extensisq_amd import device, lazy
    from extensisq_amd.common import RungeKutta
    worker = CopyWorker()
    monkeypatch.setattr(lazy, "_worker", worker)
    lib = _FakeLib()
    real_event = threading.Event

    def event_with_a_collection():
        solver = object.__new__(RungeKutta)
        solver._dev = _fake_context(lib, 400)
        solver.__del__()                                   # (the collector, right here)
        return real_event()
    monkeypatch.setattr(lazy.threading, "Event", event_with_a_collection)
    t0 = time.perf_counter()
    done, box = worker.submit(lambda: 7)
    assert done.wait(5.0) and box == [7]
    assert time.perf_counter() - t0 < 2.0
    assert worker.wait_idle(5.0)
    monkeypatch.setattr(lazy.threading, "Event", real_event)
    device._drain_graveyard()
    assert lib.destroyed and set(lib.destroyed) == {400}   # (Thread() makes an Event too)


def test_warm_buffers_can_be_asked_for_ahead_of_the_pattern():
    """`expect`: a solver made by solve_ivp knows that per-step downloads are coming; two
    arrays are faulted in while its constructor runs, the full depth only once the
    downloads have shown"""
    import time
    from extensisq_amd.device import WarmBuffers
    w = WarmBuffers(depth=4, workers=2)
    nbytes = 16 << 20
    assert w.take(nbytes // 8, np.float64).nbytes == nbytes      # a one-off: cold, no threads
    assert not w._threads
    w2 = WarmBuffers(depth=4, workers=2)
    w2.expect(nbytes, count=2)
    t0 = time.time()
    while len(w2._ready) < 2 and time.time() - t0 < 20:
        time.sleep(0.01)
    time.sleep(0.2)
    assert len(w2._ready) == 2                                   # two, not the full depth
    ready = {b.ctypes.data for b in w2._ready}
    out = w2.take(nbytes // 8, np.float64)                       # the first download: warm
    assert out.ctypes.data in ready and out.shape == (nbytes // 8,)
    t0 = time.time()
    while len(w2._ready) < 4 and time.time() - t0 < 20:          # the pattern has shown
        time.sleep(0.01)
    assert len(w2._ready) == 4
    w2.expect(8 << 20)                                           # another size: start over
    assert w2._nbytes == 8 << 20 and all(b.nbytes == 8 << 20 for b in w2._ready)
