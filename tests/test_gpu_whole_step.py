"""Round 6: whole-step device plans of the pairs that test an EARLY error estimate
(BS5, reference bogacki.py:238-346; CFMR7osc, calvo.py:152-261) and chain sweeps that
run THROUGH the end of an FSAL step (y_new, the end-point stage K_s = f(t + h, y_new)
and the error norm inside the sweep; common.py:341-351).

* the whole-step attempt (`esq_rk_set_pre`: stages, early estimate as the last target
  of a chain sweep, the rest of the attempt enqueued behind it, ONE host wait) takes
  bit for bit the steps of the round-5 sequence of pieces (`ESQ_PRE_WHOLE=0`): K, y,
  nfev, NFS, accepted and rejected attempts -- with early rejections on the way;
* the chain through the end of the step equals the chain + end-point sweep pair
  (`ESQ_CHAIN_ERRNORM=0`) bit for bit in K and y;
* BS5 / CFMR7osc / Pr7 at the BASELINE grid against the ORACLE;
* the reference's golden traces (Duffing nfev 212, README t-grid) run in host-RHS
  mode and are covered by tests/test_gpu_parity.py::test_trajectory_golden.
"""
import ctypes as C

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal

from scipy.integrate import solve_ivp

import extensisq_amd as esq
from oracle import problems as pb
from oracle import rk_oracle

pytestmark = pytest.mark.gpu


def _plugin(plugin, N):
    if plugin == "bruss":
        return (lambda: esq.Brusselator2D(N)), pb.bruss2d_y0(N), pb.bruss2d_rho(N)
    return (lambda: esq.Heat2D(N)), pb.heat2d_y0(N), pb.heat2d_rho(N)


def _labels(solver):
    return sorted(row[0] for row in solver._dev.profile_kernels())


def _pre_stats(solver):
    """(accepted launches ahead used, dropped)"""
    used, dropped = C.c_long(), C.c_long()
    solver._chk(solver._lib.esq_rk_launch_ahead_stats(solver._ctx, C.byref(used),
                                                      C.byref(dropped)),
                "esq_rk_launch_ahead_stats")
    return used.value, dropped.value


@pytest.mark.parametrize("name,plugin,N,rows", [
    ("BS5", "bruss", 48, 12), ("BS5", "bruss", 130, 9), ("BS5", "heat", 130, 30),
    ("BS5", "heat", 258, 7), ("BS5", "bruss", 512, 0), ("BS5", "heat", 700, 0),
    ("CFMR7osc", "bruss", 36, 8), ("CFMR7osc", "bruss", 124, 30),
    ("CFMR7osc", "heat", 250, 11), ("CFMR7osc", "bruss", 512, 0),
    ("CFMR7osc", "heat", 700, 0)])
def test_whole_step_attempts_equal_the_pieces_bit_for_bit(monkeypatch, name, plugin, N,
                                                          rows):
    """fixed step (every attempt accepted): whole-step attempts against the sequence of
    pieces; rows > 0 forces the tile height (and lifts the small-grid rule), rows == 0
    is the library's own choice on a grid the chains take by themselves"""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7, nfev_stiff_detect=0)
    cls = getattr(esq, name)
    if rows:
        monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    whole = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_PRE_WHOLE", "0")
    pieces = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_PRE_WHOLE")
    assert whole._pre_whole and not pieces._pre_whole
    whole._dev.profile_enable([0, 1, 2])
    for _ in range(4):
        assert whole.step() is None and pieces.step() is None
        assert whole.t == pieces.t
        assert_allclose(whole.error_norm_old, pieces.error_norm_old, rtol=1e-11)
        assert_equal(whole.y, pieces.y)
        assert_equal(whole.K, pieces.K)
    assert whole.nfev == pieces.nfev and whole.pre_discards == 0
    labels = _labels(whole)
    # the estimate rode on a chain sweep: no pass of its own
    assert any("+pre" in lab for lab in labels), labels
    assert not any(lab.startswith("k_pre_error") for lab in labels), labels
    if name == "BS5":       # ... and the FSAL end of the step is one more chain sweep
        assert any("+errnorm" in lab for lab in labels), labels
        assert not any(lab.startswith("rhs+errnorm") for lab in labels), labels


@pytest.mark.parametrize("name,plugin,N", [
    ("BS5", "bruss", 130), ("BS5", "heat", 258), ("CFMR7osc", "bruss", 124),
    ("CFMR7osc", "heat", 250), ("BS5", "bruss", 512)])
def test_whole_step_controller_with_early_rejections_equals_the_pieces(monkeypatch, name,
                                                                      plugin, N):
    """the controller left alone from a first step far beyond the stability limit: the
    early estimate rejects attempts (their speculative tails are thrown away and
    counted), the final estimate rejects others; both runs take the same decisions"""
    mk, y0, rho = _plugin(plugin, N)
    kw = dict(first_step=60.0 / rho, rtol=1e-5, atol=1e-8, nfev_stiff_detect=0)
    cls = getattr(esq, name)
    if N < 500:
        monkeypatch.setenv("ESQ_CHAIN_ROWS", "10")
    whole = cls(mk(), 0.0, y0, 400.0 / rho, **kw)
    nfs_whole = []
    for _ in range(12):
        if whole.status != "running":
            break
        assert whole.step() is None
        nfs_whole.append(int(esq.NFS[()]))
    monkeypatch.setenv("ESQ_PRE_WHOLE", "0")
    pieces = cls(mk(), 0.0, y0, 400.0 / rho, **kw)
    monkeypatch.delenv("ESQ_PRE_WHOLE")
    nfs_pieces = []
    for _ in range(len(nfs_whole)):
        assert pieces.step() is None
        nfs_pieces.append(int(esq.NFS[()]))
    assert nfs_whole == nfs_pieces and nfs_whole[-1] >= 1
    assert whole.pre_discards >= 1              # an EARLY rejection was among them
    assert whole.nfev == pieces.nfev
    # (the error norms of the two runs differ in their last digits -- other partial
    # sums -- and the step sizes follow them)
    assert_allclose(whole.t, pieces.t, rtol=1e-9)
    assert_allclose(whole.h_abs, pieces.h_abs, rtol=1e-7)
    assert_allclose(whole.y, pieces.y, rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("name", ["BS5", "CFMR7osc"])
def test_whole_step_controller_matches_the_oracle(monkeypatch, name):
    """... and the ORACLE's decisions: accepted steps, rejected attempts (early and
    final), RHS evaluations, times"""
    N = 130
    mk, y0, rho = _plugin("bruss", N)
    kw = dict(first_step=60.0 / rho, rtol=1e-5, atol=1e-8, nfev_stiff_detect=0)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", "10")
    d = getattr(esq, name)(mk(), 0.0, y0, 400.0 / rho, **kw)
    o = rk_oracle.METHODS[name](pb.bruss2d_rhs(N), 0.0, y0, 400.0 / rho, **kw)
    steps = 0
    while o.status == "running" and steps < 10:
        assert o.step() is None
        steps += 1
    nfs_ref = int(rk_oracle.NFS[()])
    for _ in range(steps):
        assert d.step() is None
    assert d._pre_whole and int(esq.NFS[()]) == nfs_ref and nfs_ref >= 1
    assert d.nfev == o.nfev
    # (the COUNTS are the assertion: attempts far beyond the stability limit have
    # error norms of 1e3 .. 1e6 that follow the last digits of a cancelling sum, and
    # the step sizes follow them -- as test_device_rhs_long_trajectory finds)
    assert_allclose(d.t, o.t, rtol=1e-4)
    assert_allclose(d.h_abs, o.h_abs, rtol=1e-2)
    assert_allclose(d.y, o.y, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name,plugin,N,rows", [
    ("Ts5", "heat", 258, 7), ("Ts5", "heat", 130, 30), ("Ts5", "bruss", 48, 12),
    ("Ts5", "bruss", 130, 9), ("Ts5", "heat", 1000, 0), ("Ts5", "bruss", 512, 0),
    ("BS5", "heat", 130, 8), ("BS5", "bruss", 512, 0)])
def test_chain_through_the_end_of_an_fsal_step_is_bit_identical(monkeypatch, name, plugin,
                                                               N, rows):
    """ESQ_CHAIN_ERRNORM=0 (the chain ends in y_new, the end-point sweep carries the
    error norm: round 5) against the chain that runs through the end of the step"""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7, nfev_stiff_detect=0)
    cls = getattr(esq, name)
    if rows:
        monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    through = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_CHAIN_ERRNORM", "0")
    pair = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN_ERRNORM")
    through._dev.profile_enable([0, 1, 2])
    pair._dev.profile_enable([0, 1, 2])
    for _ in range(4):
        assert through.step() is None and pair.step() is None
        assert through.t == pair.t
        assert_allclose(through.error_norm_old, pair.error_norm_old, rtol=1e-11)
        assert_equal(through.y, pair.y)
        assert_equal(through.f, pair.f)            # K_s, the next step's K_0
    assert_equal(through.K, pair.K)                # (rows left unwritten: restored)
    assert through.nfev == pair.nfev
    assert any("+errnorm" in lab for lab in _labels(through))
    assert any(lab.startswith("rhs+errnorm") for lab in _labels(pair))
    assert not any(lab.startswith("chain") and "+errnorm" in lab for lab in _labels(pair))


def test_ts5_whole_step_is_one_launch_and_runs_ahead():
    """config 2's step (Ts5, heat, N = 1000): ONE chain sweep from K[0] through the
    error norm + the final sum; at max_step it is enqueued behind the previous step's
    error norm (launch ahead) and taken over by the next step"""
    N = 1000
    mk, y0, rho = _plugin("heat", N)
    h = 1.0 / rho
    # (tolerances at which the controller keeps h at max_step: both runs then take
    # bitwise the same step sizes)
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    d = esq.Ts5(mk(), 0.0, y0, 1.0, **kw)
    o = rk_oracle.Ts5(pb.heat2d_rhs(N), 0.0, y0, 1.0, **kw)
    d._dev.profile_enable([0, 1, 2])
    for _ in range(5):
        assert d.step() is None and o.step() is None
        assert d.t == o.t
        assert_allclose(d.error_norm_old, o.error_norm_old, rtol=1e-5)
    labels = _labels(d)
    assert labels == ["chain6+errnorm<1>"], labels
    used, dropped = _pre_stats(d)
    assert used >= 3 and dropped == 0, (used, dropped)
    kmax = np.abs(o.K).max()
    assert_allclose(d.y, o.y, rtol=1e-11, atol=1e-13)
    assert_allclose(d.K, o.K, rtol=0, atol=1e-10 * kmax)


@pytest.mark.parametrize("name,plugin,N,plan", [
    ("BS5", "bruss", 2236, ["chain2+errnorm<6>", "chain5+pre<1>"]),
    ("BS5", "heat", 2236, None),
    ("CFMR7osc", "bruss", 2236, None),
    ("Pr7", "bruss", 2236, None)])
def test_full_size_three_steps_match_oracle_early_estimate_pairs(name, plugin, N, plan):
    """the BASELINE grid (n = 9 999 392 / 4 999 696), THREE steps against the oracle
    with nothing read in between: the launch sequence tools/method_sweep.py times for
    the pairs with an early estimate (and Pr7), the plan by name"""
    mk, y0, rho = _plugin(plugin, N)
    cpu = pb.bruss2d_rhs(N) if plugin == "bruss" else pb.heat2d_rhs(N)
    h = 1.0 / rho
    rtol, atol = (1e-6, 1e-9) if plugin == "bruss" else (1e-3, 1e-6)
    kw = dict(first_step=h, max_step=h, rtol=rtol, atol=atol, nfev_stiff_detect=0)
    d = getattr(esq, name)(mk(), 0.0, y0, 1.0, **kw)
    o = rk_oracle.METHODS[name](cpu, 0.0, y0, 1.0, **kw)
    errs = []
    for k in range(3):
        if k == 2:
            d._dev.profile_reset()
            d._dev.profile_enable([0, 1, 2])
        assert d.step() is None and o.step() is None
        assert d.t == o.t
        errs.append((d.error_norm_old, o.error_norm_old))
    d._dev.profile_enable(None)
    labels = _labels(d)
    assert d.nfev == o.nfev and int(esq.NFS[()]) == 0
    kmax, ymax = np.abs(o.K).max(), np.abs(o.y).max()
    k_atol = 10 * (2e-13 * kmax + 8 * np.finfo(float).eps * rho * ymax)
    assert_allclose(d.y, o.y, rtol=1e-11, atol=h * k_atol)
    assert_allclose(d.K, o.K, rtol=0, atol=k_atol)
    for got, ref in errs:
        assert_allclose(got, ref, rtol=1e-5)
    if plan is not None:
        assert labels == plan, labels


def test_eight_threads_construct_solvers_with_their_own_switches():
    """`esq_options=` travels as arguments (esq_create2 / esq_rhs_set_options), not through
    the process environment: eight threads construct and step solvers with DIFFERENT
    chain depths and tile heights at the same moment, each gets the plan of its own
    switches -- and all eight states are the same bits (ADVICE r05
    medium: until round 5 the keyword wrote os.environ for the duration of the
    constructor, two constructors with different options raced)"""
    import os
    import threading
    N = 96
    rho = esq.Brusselator2D(N).spectral_radius()
    y0 = pb.bruss2d_y0(N)
    h = 0.25 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    cases = [(1, 12), (2, 12), (3, 9), (4, 12), (1, 9), (2, 7), (3, 12), (4, 8)]
    env_before = dict(os.environ)
    gate = threading.Barrier(len(cases))
    out = [None] * len(cases)

    def run(k):
        depth, rows = cases[k]
        try:
            gate.wait(30)
            s = esq.Pr8(esq.Brusselator2D(N), 0.0, y0, 1.0,
                        esq_options={"chain_depth": depth, "chain_rows": rows}, **kw)
            s._dev.profile_enable([0, 1, 2])
            for _ in range(3):
                assert s.step() is None
            out[k] = (np.array(s.y), _labels(s))
        except BaseException as exc:                          # noqa: BLE001
            out[k] = exc

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(cases))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert dict(os.environ) == env_before
    for k, (depth, rows) in enumerate(cases):
        assert not isinstance(out[k], BaseException), (cases[k], out[k])
        y, labels = out[k]
        assert_equal(y, out[0][0])
        depths = [int(lab[5]) for lab in labels if lab.startswith("chain")]
        if depth == 1:
            assert not depths, (cases[k], labels)
        else:
            # (one stage more where the end-point derivative rides in front)
            assert depths and max(depths) in (depth, depth + 1), (cases[k], labels)


# ------------------------------------------ device-resident cubic Hermite interpolant
def _heun():
    class Heun(esq.RungeKutta):       # a user tableau WITHOUT P (ref docs/Demo_own_RK.ipynb)
        n_stages, order, order_secondary = 2, 2, 1
        A = np.array([[0.0, 0.0], [1.0, 0.0]])
        B = np.array([0.5, 0.5])
        C = np.array([0.0, 1.0])
        E = np.array([0.5, -0.5, 0.0])
    return Heun


@pytest.mark.parametrize("which", ["ssv2stab", "heun", "ssv2stab_small"])
def test_cubic_interpolant_lives_on_the_device(which):
    """dense_output() of SSV2stab (sommeijer.py:400-406) and of a tableau without P
    (common.py:366-368, 793-821): large states get the cubic Hermite interpolant as a
    device-resident Horner form (esq_dense_create_vecs) -- nothing is downloaded when it
    is made; it equals the reference's formula on host copies of the four vectors to
    rounding, and outlives the step and the solver"""
    from extensisq_amd.common import CubicDenseOutput, DeviceHornerDenseOutput
    N = 24 if which == "ssv2stab_small" else 96
    rhs = esq.Heat2D(N)
    y0 = pb.heat2d_y0(N, seed=3)
    rho = rhs.spectral_radius()
    if which == "heun":
        s = _heun()(rhs, 0.0, y0, 1.0, first_step=0.2 / rho, max_step=0.2 / rho,
                    rtol=1e-3, atol=1e-6)
    else:
        s = esq.SSV2stab(rhs, 0.0, y0, 1.0, rtol=1e-4, atol=1e-6,
                         rho_jac=lambda t, y: rho, const_jac=True)
    for _ in range(3):
        assert s.step() is None
    before = _lib_copies()
    sol = s.dense_output()
    if which == "ssv2stab_small":               # below the threshold: the host interpolant
        assert isinstance(sol, CubicDenseOutput)
        return
    assert isinstance(sol, DeviceHornerDenseOutput)
    assert _lib_copies() == before              # ... and nothing came to the host for it
    # the reference's formula on host copies of the four vectors
    if which == "heun":
        y_old, y, f_old, f = s.y_old, s.y, s.f_old, s.f
    else:
        from extensisq_amd._lib import SLOT_K
        r = s._r
        y_old, y, f_old, f = (s._dev.download(SLOT_K, r[k]) for k in ("yold", "yn", "fold", "fn"))
    ref = CubicDenseOutput(s.t_old, s.t, y_old, np.asarray(y), f_old, f)
    tc = s.t_old + (s.t - s.t_old) * np.array([0.0, 0.125, 0.5, 0.9, 1.0])
    scale = np.abs(y0).max()
    assert_allclose(sol(tc), ref(tc), rtol=0, atol=4e-15 * scale)
    assert_allclose(sol(s.t_old), y_old, rtol=0, atol=0)     # x = 0: the base itself
    assert_allclose(sol(s.t), np.asarray(y), rtol=0, atol=4e-15 * scale)
    # it owns its memory: the solver steps on, is closed, the interpolant still answers
    want = sol(tc[2])
    assert s.step() is None
    s._dev.close()
    assert_equal(sol(tc[2]), want)


def _lib_copies():
    """downloads so far (a spy on DeviceContext.download, below)"""
    return _DOWNLOADS[0]


_DOWNLOADS = [0]


@pytest.fixture(autouse=True)
def _count_downloads(monkeypatch):
    from extensisq_amd.device import DeviceContext
    real = DeviceContext.download

    def counting(self, *a, **k):
        _DOWNLOADS[0] += 1
        return real(self, *a, **k)
    monkeypatch.setattr(DeviceContext, "download", counting)
    yield


def test_solve_ivp_dense_output_of_ssv2stab_matches_the_oracle_on_a_large_state():
    """the drop-in call with dense_output=True on a device-resident state: every
    step's interpolant is the device-resident cubic; against the oracle's"""
    from extensisq_amd.common import DeviceHornerDenseOutput
    from oracle import rkc_oracle
    N = 80
    y0 = pb.heat2d_y0(N, seed=2)
    kw = dict(rtol=1e-4, atol=1e-6, dense_output=True)
    res = solve_ivp(esq.Heat2D(N), (0, 0.004), y0, method=esq.SSV2stab, **kw)
    ref = solve_ivp(pb.heat2d_rhs(N), (0, 0.004), y0, method=rkc_oracle.SSV2stab, **kw)
    assert res.success and len(res.t) == len(ref.t)
    assert all(isinstance(i, DeviceHornerDenseOutput) for i in res.sol.interpolants)
    tc = np.linspace(0, 0.004, 9)
    assert_allclose(res.sol(tc), ref.sol(tc), rtol=1e-7, atol=1e-10)


def test_state_dependent_spectral_radius_bound_downloads_only_if_it_looks(monkeypatch):
    """SSV2stab's `rho_jac(t, y)` (sommeijer.py:174-176) on a large device-resident state:
    the function gets a deferred mirror -- a bound that ignores y costs no download, one
    that reads y gets the state's values (and the run is the same run)"""
    import extensisq_amd.common as cm
    monkeypatch.setattr(cm, "LAZY_MIN_BYTES", 1 << 16)       # (a 96 x 96 state counts as large)
    N = 96
    rhs = esq.Heat2D(N)
    rho = rhs.spectral_radius()
    y0 = pb.heat2d_y0(N, seed=4)
    seen = []

    def ignores(t, y):
        seen.append(type(y).__name__)
        return rho

    def looks(t, y):
        return rho * (1.0 + 0.0 * float(np.abs(y).max()))

    kw = dict(rtol=1e-4, atol=1e-6)
    a = esq.SSV2stab(rhs, 0.0, y0, 1.0, rho_jac=ignores, **kw)
    start = _lib_copies()
    for _ in range(4):
        assert a.step() is None
    # (the constructor's check and the first step see the host copy of y0 that exists
    # anyway; from then on the state is on the device only)
    assert _lib_copies() == start and seen[-2:] == ["LazyState", "LazyState"], seen
    b = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1.0, rho_jac=looks, **kw)
    start = _lib_copies()
    for _ in range(4):
        assert b.step() is None
    assert _lib_copies() > start
    assert a.t == b.t and a.nfev == b.nfev
    assert_equal(np.asarray(a.y), np.asarray(b.y))
