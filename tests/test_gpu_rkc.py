"""GPU parity tests of the device-resident SSV2stab (RKC) against golden
vectors from the real reference and against the CPU oracle.  The reference's
own tests/ hold no SSV2stab test; the pins are the stage vectors and the
published integer table of docs/Demo_SSV2stab.ipynb:350-356."""
import ctypes as C
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy.integrate import solve_ivp

import extensisq_amd as esq
from extensisq_amd import sommeijer as dev_rkc
from extensisq_amd._lib import SLOT_K
from oracle import problems as pb
from oracle import rkc_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [2, 3, 10, 100, 132])
@pytest.mark.parametrize("mode", ["host_rhs", "device_rhs"])
def test_stages_golden(golden_dir, m, mode):
    g = np.load(os.path.join(golden_dir, "rkc_stages.npz"))
    lam, yn = g[f"m{m}/lam"], g[f"m{m}/yn"]
    h = float(g[f"m{m}/h"])
    cpu = lambda t, y: lam * y + np.sin(t)  # noqa: E731
    fun = cpu if mode == "host_rhs" else esq.DiagonalLinear(lam, 1.0)
    s = esq.SSV2stab(fun, 0.0, yn, 1.0, first_step=1e-3,
                     rho_jac=lambda t, y: 50.0)
    yrow = s._stages(0.0, h, m)
    y = s._dev.download(SLOT_K, yrow)
    gold = g[f"m{m}/y"]
    assert_allclose(y, gold, rtol=1e-13, atol=1e-13 * np.abs(gold).max())


@pytest.mark.parametrize("tol,expect", [
    (1e-1, (6, 1, 402, 132)),      # docs/Demo_SSV2stab.ipynb:350-356
    (1e-2, (15, 4, 729, 85)),
    (1e-3, (27, 2, 786, 40)),
    (1e-4, (57, 0, 1087, 26)),
    (1e-5, (129, 1, 1682, 20)),
    (1e-6, (262, 0, 2445, 12)),
])
def test_published_table(golden_dir, tol, expect):
    """3-D tanh heat problem (n = 59 319), host-RHS mode"""
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)[f"tanh3d_tol{tol:.0e}"]
    fun, y0, rho = pb.tanh3d_problem(39)
    res = solve_ivp(fun, (0, 0.7), y0, method=esq.SSV2stab, rtol=tol,
                    atol=tol, const_jac=True, rho_jac=rho)
    nfs = int(dev_rkc.nrejct[()])
    got = (int(res.t.size - 1 + nfs), nfs, int(res.nfev), int(dev_rkc.maxm[()]))
    assert got == expect
    assert_allclose(res.t, gold["t"], rtol=1e-9)
    assert_allclose(res.y[::5000, -1], gold["y_probe"], rtol=1e-7)


@pytest.mark.parametrize("tol,expect", [
    (1e-4, (51, 1, 525, 21, 36)),  # docs/Demo_SSV2stab.ipynb:207-211:
    (1e-5, (124, 0, 781, 27, 29)),  # steps (failed) / f-evals / f-sigma / s-max
    (1e-6, (270, 0, 1270, 39, 20)),
    (1e-7, (581, 0, 2147, 65, 14)),
])
def test_published_combustion_table(golden_dir, tol, expect):
    """3-D combustion problem (n = 128 000), host-RHS mode, spectral radius by
    the device-resident power iteration: the published integer table and the
    reference's accepted times"""
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)[f"combustion_tol{tol:.0e}"]
    fun, y0 = pb.combustion3d_problem(40)
    res = solve_ivp(fun, (0, 0.3), y0, method=esq.SSV2stab, rtol=tol, atol=tol)
    nfs = int(dev_rkc.nrejct[()])
    got = (int(res.t.size - 1 + nfs), nfs, int(res.nfev),
           int(dev_rkc.nfesig[()]), int(dev_rkc.maxm[()]))
    assert got == expect
    assert_allclose(res.t, gold["t"], rtol=1e-8)
    assert_allclose(res.y[::4001, -1], gold["y_probe"], rtol=1e-6)


@pytest.mark.parametrize("mode", ["host_rhs", "device_rhs"])
def test_power_iteration_golden(golden_dir, mode):
    """rho_jac=None: the nonlinear power iteration runs on the device"""
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)["heat2d_rho_power"]
    N = 24
    fun = pb.heat2d_rhs(N) if mode == "host_rhs" else esq.Heat2D(N)
    res = solve_ivp(fun, (0, 0.01), pb.heat2d_y0(N, seed=1234),
                    method=esq.SSV2stab, rtol=1e-4, atol=1e-6)
    nfs = int(dev_rkc.nrejct[()])
    assert (int(res.t.size - 1 + nfs), nfs, int(dev_rkc.maxm[()]),
            int(dev_rkc.nfesig[()])) == (gold["steps"], gold["nfs"],
                                         gold["maxm"], gold["nfesig"])
    if mode == "host_rhs":
        assert int(res.nfev) == gold["nfev"]
    assert_allclose(res.t, gold["t"], rtol=1e-9)
    assert_allclose(res.y[:, -1], gold["y_end"], rtol=1e-8, atol=1e-12)


def test_diffusion3d_vs_oracle():
    """BASELINE.json configs[3] at small N with the device RHS, m ~ 30"""
    N = 21
    rhs = esq.Diffusion3D(N)
    rho = rhs.spectral_radius()
    h0 = 600.0 / rho
    kw = dict(rtol=1e-3, atol=1e-3, const_jac=True, first_step=h0,
              rho_jac=lambda t, y: rho)
    d = esq.SSV2stab(rhs, 0.0, pb.diff3d_y0(N), 1.0, **kw)
    o = rkc_oracle.SSV2stab(pb.diff3d_rhs(N), 0.0, pb.diff3d_y0(N), 1.0, **kw)
    for _ in range(4):
        assert d.step() is None and o.step() is None
        assert_allclose(d.t, o.t, rtol=1e-12)
        assert_allclose(d.y, o.y, rtol=1e-11, atol=1e-14)
        assert_allclose(d.errold, o.errold, rtol=1e-6)
    assert d.nfev == o.nfev


def test_dense_output_and_ctor_errors():
    N = 16
    res = solve_ivp(esq.Heat2D(N), (0, 0.005), pb.heat2d_y0(N), rtol=1e-4,
                    atol=1e-6, method=esq.SSV2stab, dense_output=True)
    ref = solve_ivp(pb.heat2d_rhs(N), (0, 0.005), pb.heat2d_y0(N), rtol=1e-4,
                    atol=1e-6, method=rkc_oracle.SSV2stab, dense_output=True)
    tc = np.linspace(0, 0.005, 7)
    assert_allclose(res.sol(tc), ref.sol(tc), rtol=1e-7, atol=1e-10)
    with pytest.raises(TypeError):
        esq.SSV2stab(pb.heat2d_rhs(N), 0, pb.heat2d_y0(N), 1, const_jac=1)
    with pytest.raises(TypeError):
        esq.SSV2stab(pb.heat2d_rhs(N), 0, pb.heat2d_y0(N), 1, rho_jac=3.0)
    with pytest.raises(ValueError):
        esq.SSV2stab(pb.heat2d_rhs(N), 0, pb.heat2d_y0(N), 1,
                     rho_jac=lambda t, y: -1.0)


def test_full_size_rkc_diffusion_step_matches_oracle():
    """configs[3]: SSV2stab, 3-D diffusion N = 159 (n = 4 019 679), m = 100
    stages in the first step, against the oracle's step"""
    N = 159
    rhs = esq.Diffusion3D(N)
    rho = rhs.spectral_radius()
    h0 = ((100 - 1) ** 2 - 1 + 0.5 * (2 * 100 - 1)) / (1.54 * rho)
    kw = dict(rtol=1e-3, atol=1e-3, const_jac=True, first_step=h0, max_step=h0,
              rho_jac=lambda t, y: rho)
    y0 = pb.diff3d_y0(N)
    d = esq.SSV2stab(rhs, 0.0, y0, 1.0, **kw)
    o = rkc_oracle.SSV2stab(pb.diff3d_rhs(N), 0.0, y0, 1.0, **kw)
    assert d.step() is None and o.step() is None
    assert int(dev_rkc.maxm[()]) == int(rkc_oracle.maxm[()]) == 100
    # the m = 100 attempt is rejected in both and retried with a step derived
    # from its error norm: t agrees to rounding, not bitwise
    assert int(dev_rkc.nrejct[()]) == int(rkc_oracle.nrejct[()])
    assert_allclose(d.t, o.t, rtol=1e-12)
    assert d.nfev == o.nfev
    # 100 stages of a recursion whose RHS has Lipschitz constant rho ~ 3e5:
    # rounding differences are amplified along the stages
    assert_allclose(d.y, o.y, rtol=1e-9, atol=1e-12)
    assert_allclose(d.errold, o.errold, rtol=1e-5)


def test_rkc_depth5_default_path_is_bit_identical(monkeypatch):
    """N = 180: the plugin picks FIVE stages per chain sweep by itself (device.py:
    6 vectors x 8 n bytes beyond 256 MiB) -- the launch sequence of every "beyond the
    Infinity Cache" SSV2stab figure -- against one launch per stage, bit for bit, for
    stage counts that end in chains of every length"""
    N = 180
    assert esq.Diffusion3D(N)._rkc_chain_depth == 5
    for m in (7, 23, 100):
        monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
        ref, s1 = _stage_run(N, m)
        monkeypatch.delenv("ESQ_RKC_DEPTH")
        got, s2 = _stage_run(N, m)
        assert np.isfinite(ref).all()
        np.testing.assert_array_equal(got, ref, err_msg=f"m = {m}")
        assert s1.nfev == s2.nfev
        names = [k[0] for k in _profiled_kernels(s2, m)]
        # (m = 7: six stages = 4 + 2, never a single stage at the end)
        want = "rkc_chain4" if m == 7 else "rkc_chain5"
        assert any(k.startswith(want) for k in names), names
        assert "k_rkc_first" not in names and any("-first" in k for k in names), names


def test_rkc_depth5_step_matches_oracle():
    """one SSV2stab step at N = 200 (n = 8e6: depth-5 chain sweeps by default, the
    end of the step inside the last one), m = 100 stages, against the oracle's step
    (sommeijer.py:162-329) -- as test_full_size_rkc_diffusion_step_matches_oracle does
    at N = 159 for the depth-4 sweeps"""
    N = 200
    rhs = esq.Diffusion3D(N)
    assert rhs._rkc_chain_depth == 5
    rho = rhs.spectral_radius()
    h0 = ((100 - 1) ** 2 - 1 + 0.5 * (2 * 100 - 1)) / (1.54 * rho)
    kw = dict(rtol=1e-3, atol=1e-3, const_jac=True, first_step=h0, max_step=h0,
              rho_jac=lambda t, y: rho)
    y0 = pb.diff3d_y0(N)
    d = esq.SSV2stab(rhs, 0.0, y0, 1.0, **kw)
    o = rkc_oracle.SSV2stab(pb.diff3d_rhs(N), 0.0, y0, 1.0, **kw)
    assert d.step() is None and o.step() is None
    assert int(dev_rkc.maxm[()]) == int(rkc_oracle.maxm[()]) == 100
    assert int(dev_rkc.nrejct[()]) == int(rkc_oracle.nrejct[()])
    assert_allclose(d.t, o.t, rtol=1e-12)
    assert d.nfev == o.nfev
    assert_allclose(d.y, o.y, rtol=1e-9, atol=1e-12)
    assert_allclose(d.errold, o.errold, rtol=1e-5)
    names = [k[0] for k in _profiled_kernels(d, int(dev_rkc.maxm[()]))]
    assert any(k.startswith("rkc_chain5") for k in names), names


@pytest.mark.parametrize("plugin,N", [("heat", 6), ("heat", 130), ("diff3d", 5),
                                      ("diff3d", 24), ("diff3d", 41)])
def test_rkc_chained_stage_is_bit_identical(monkeypatch, plugin, N):
    """ESQ_RKC_CHAIN (default on): one sweep evaluates f(y_{j-1}) and finishes
    the Chebyshev recursion without storing f -- must equal the two-kernel path
    bit for bit"""
    if plugin == "heat":
        mk, y0 = (lambda: esq.Heat2D(N)), pb.heat2d_y0(N)
    else:
        mk, y0 = (lambda: esq.Diffusion3D(N)), pb.diff3d_y0(N)
    rho = mk().spectral_radius()
    kw = dict(rtol=1e-4, atol=1e-6, const_jac=True, rho_jac=lambda t, y: rho,
              first_step=150.0 / rho)
    a = esq.SSV2stab(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_RKC_CHAIN", "0")
    b = esq.SSV2stab(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_CHAIN")
    for _ in range(4):
        assert a.step() is None and b.step() is None
        assert a.t == b.t and a.errold == b.errold
        np.testing.assert_array_equal(a.y, b.y)
    assert a.nfev == b.nfev and a.nfev > 20


def _stage_run(N, m, h_factor=1.0):
    """all m stages of one Chebyshev step on the 3-D diffusion plugin from a
    non-smooth state (every point different); returns the final iterate and the
    second-to-last one is not needed"""
    rhs = esq.Diffusion3D(N)
    rho = rhs.spectral_radius()
    rng = np.random.default_rng(7 + N)
    y0 = pb.diff3d_y0(N) + 0.1 * rng.standard_normal(N ** 3)
    s = esq.SSV2stab(rhs, 0.0, y0, 1.0, rtol=1e-3, atol=1e-3, const_jac=True,
                     first_step=1e-6, rho_jac=lambda t, y: rho)
    h = h_factor * (m * m - 1) / (1.54 * rho)
    yrow = s._stages(0.0, h, m)
    return s._dev.download(SLOT_K, yrow), s


@pytest.mark.parametrize("N,planes", [(5, 0), (13, 3), (24, 0), (41, 7), (57, 0),
                                      (64, 5), (70, 16)])
@pytest.mark.parametrize("depth", [2, 3, 4, 5, 6])
def test_rkc_chain_sweeps_are_bit_identical(monkeypatch, N, planes, depth):
    """ESQ_RKC_DEPTH stages per marching sweep (esq_rhs_rkc_chain_fn, the 3-D
    plugin's patch sweeps: esq_rkc3d.hpp) against one launch per stage: the final
    iterate of m stages bit for bit, for stage counts that end in chains of every
    length and in a single stage, grids of one and of several patches per plane,
    forced tile depths (run-in planes, plane ranges that do not divide N).  Depth 5
    is what the plugin picks by itself from N = 178 (device.py: the sweep's six
    vectors beyond the Infinity Cache), depth 6 is instantiated for tuning."""
    monkeypatch.setenv("ESQ_RKC_FORCE", "1")
    monkeypatch.setenv("ESQ_RKC_MAXDEPTH", "6")
    for m in (2, 3, 4, 5, 6, 7, 8, 10, 23):
        monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
        ref, s1 = _stage_run(N, m)
        monkeypatch.setenv("ESQ_RKC_DEPTH", str(depth))
        monkeypatch.setenv("ESQ_RKC_PLANES", str(planes))
        got, s2 = _stage_run(N, m)
        monkeypatch.delenv("ESQ_RKC_PLANES")
        assert np.isfinite(ref).all()
        np.testing.assert_array_equal(got, ref, err_msg=f"m = {m}")
        assert s1.nfev == s2.nfev
        names = [k[0] for k in _profiled_kernels(s2, m)]
        if m == 23:     # (22 stages: chains of the full depth, whatever ends the step)
            assert any(k.startswith("rkc_chain%d" % depth) for k in names), names
        if m - 1 >= 2:
            assert any(k.startswith("rkc_chain") for k in names), names
            # the chain that opens the step forms the first iterate itself ...
            assert names.count("k_rkc_first") == 0 and any("-first" in k for k in names), names
            # ... or reads it from the sweep that wrote it: the same bits
            monkeypatch.setenv("ESQ_RKC_FIRST", "0")
            got0, s3 = _stage_run(N, m)
            monkeypatch.delenv("ESQ_RKC_FIRST")
            np.testing.assert_array_equal(got0, ref, err_msg=f"m = {m}, ESQ_RKC_FIRST=0")
            assert "k_rkc_first" in [k[0] for k in _profiled_kernels(s3, m)]


def _stage_run_heat(N, m):
    """all m stages of one Chebyshev step on the 2-D heat plugin from a non-smooth state"""
    rhs = esq.Heat2D(N)
    rho = rhs.spectral_radius()
    rng = np.random.default_rng(3 + N)
    y0 = pb.heat2d_y0(N) + 0.1 * rng.standard_normal(N * N)
    s = esq.SSV2stab(rhs, 0.0, y0, 1.0, rtol=1e-3, atol=1e-3, const_jac=True,
                     first_step=1e-6, rho_jac=lambda t, y: rho)
    h = (m * m - 1) / (1.54 * rho)
    yrow = s._stages(0.0, h, m)
    return s._dev.download(SLOT_K, yrow), s


@pytest.mark.parametrize("N,rows", [(16, 3), (40, 5), (130, 12), (250, 9), (512, 0)])
@pytest.mark.parametrize("depth", [2, 3, 4, 5, 6])
def test_rkc_chain_sweeps_2d_are_bit_identical(monkeypatch, N, rows, depth):
    """the 2-D sibling (csrc/esq_rkc2d.hpp, the heat plugin's esq_rhs_rkc_chain_fn):
    ESQ_RKC_DEPTH stages per marching sweep against one launch per stage, the final
    iterate of m stages bit for bit -- tiles of forced heights (several tiles per
    grid row and column, ragged last tiles), chains of every length at the end of a
    step, the chain that opens a step forming the first iterate or reading it"""
    if rows:
        monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))       # (also lifts the small-grid rule)
    monkeypatch.setenv("ESQ_RKC_MAXDEPTH", "6")
    for m in (2, 3, 4, 5, 6, 7, 8, 10, 23):
        monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
        ref, s1 = _stage_run_heat(N, m)
        monkeypatch.setenv("ESQ_RKC_DEPTH", str(depth))
        got, s2 = _stage_run_heat(N, m)
        assert np.isfinite(ref).all()
        np.testing.assert_array_equal(got, ref, err_msg=f"m = {m}")
        assert s1.nfev == s2.nfev
        names = [k[0] for k in _profiled_kernels(s2, m)]
        if m - 1 >= 2:
            assert any(k.startswith("rkc_chain") for k in names), names
            assert names.count("k_rkc_first") == 0 and any("-first" in k for k in names), names
            monkeypatch.setenv("ESQ_RKC_FIRST", "0")
            got0, s3 = _stage_run_heat(N, m)
            monkeypatch.delenv("ESQ_RKC_FIRST")
            np.testing.assert_array_equal(got0, ref, err_msg=f"m = {m}, ESQ_RKC_FIRST=0")
            assert "k_rkc_first" in [k[0] for k in _profiled_kernels(s3, m)]


def test_rkc_chain_2d_whole_steps(monkeypatch):
    """whole adaptive SSV2stab steps on the 2-D heat plugin at a size where the chain
    sweeps run by themselves (N = 512: first iterate inside the first chain, chains of
    five, the end of the step its own sweep) against one launch per stage: identical
    t, y, error norms and counters"""
    N = 512
    rho = esq.Heat2D(N).spectral_radius()
    h0 = (40 ** 2 - 1) / (1.54 * rho) * 0.999
    kw = dict(rtol=1e-2, atol=1e-2, const_jac=True, first_step=h0, rho_jac=lambda t, y: rho)
    y0 = pb.heat2d_y0(N)
    a = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
    b = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_DEPTH")
    for _ in range(4):
        assert a.step() is None and b.step() is None
        # (the end of the step inside the last chain sweep: the error estimate summed
        # over other workgroups -- equal to rounding, and the step sizes with it)
        assert abs(a.errold - b.errold) <= 1e-13 * b.errold
        assert abs(a.t - b.t) <= 1e-13 * b.t and abs(a.absh - b.absh) <= 1e-13 * b.absh
        np.testing.assert_allclose(a.y, b.y, rtol=1e-11, atol=1e-13)
    assert a.nfev == b.nfev
    # ... and bit for bit with that form switched off
    monkeypatch.setenv("ESQ_RKC_LAST", "0")
    a0 = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_LAST")
    monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
    b0 = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_DEPTH")
    for _ in range(4):
        assert a0.step() is None and b0.step() is None
        assert a0.t == b0.t and a0.errold == b0.errold and a0.absh == b0.absh
        np.testing.assert_array_equal(a0.y, b0.y)
    m = int(dev_rkc.maxm[()])
    names = {k[0] for k in _profiled_kernels(a, m)}
    assert any(k.startswith("rkc_chain5") for k in names) and "k_rkc_first" not in names, names


def _end_run(N, m, heat=False):
    """all stages + the end of the step (esq_rkc_stages_end) from a fixed state:
    y_{n+1}, f(t + h, y_{n+1}), the error estimate's sum of squares, kernel labels"""
    import ctypes as C
    from extensisq_amd._lib import PROF_RKC, PROF_SOLERR
    rhs = esq.Heat2D(N) if heat else esq.Diffusion3D(N)
    rho = rhs.spectral_radius()
    rng = np.random.default_rng(11 + N)
    y0 = (pb.heat2d_y0(N) + 0.1 * rng.standard_normal(N * N) if heat else
          pb.diff3d_y0(N) + 0.1 * rng.standard_normal(N ** 3))
    s = esq.SSV2stab(rhs, 0.0, y0, 1.0, rtol=1e-3, atol=1e-3, const_jac=True,
                     first_step=1e-6, rho_jac=lambda t, y: rho)
    h = (m * m - 1) / (1.54 * rho)
    out = C.c_double()
    s._dev.profile_reset()
    s._dev.profile_enable([PROF_RKC, PROF_SOLERR])
    yrow, fyrow = s._stages_end(0.0, h, m, out)
    s._dev.profile_enable(None)
    names = [k[0] for k in s._dev.profile_kernels()]
    # the launches are the ones the library describes without a GPU (CPU suite:
    # tests/test_step_plans.py::test_chebyshev_step_programs)
    if os.environ.get("ESQ_RKC_DEPTH") != "1" and "ESQ_RKC_LAST" not in os.environ:
        from extensisq_amd import _lib
        buf = C.create_string_buffer(1 << 14)
        depth, forms = rhs._rkc_chain_entry(_lib.load())[1] & 0xff, rhs._rkc_chain_forms
        chained = heat or N >= 48 or os.environ.get("ESQ_RKC_FORCE") == "1"
        assert _lib.load().esq_rkc_plan_describe(m, (depth if chained else 1) | forms,
                                                 6 if heat else 5, buf, len(buf)) == 0
        want = buf.value.decode().split(" | ")[0].split()
        got = {k[0]: k[2] for k in s._dev.profile_kernels()}
        assert got == {k: want.count(k) for k in set(want)}, (m, got, want)
    return s._dev.download(SLOT_K, yrow), s._dev.download(SLOT_K, fyrow), out.value, names


@pytest.mark.parametrize("N", [13, 24, 41, 57, 64])
def test_rkc_chain_takes_the_end_of_the_step_along(monkeypatch, N):
    """the chain sweep that ends a step also evaluates f(t + h, y_{n+1}) and the error
    estimate (LAST form of esq_rhs_rkc_chain_fn): y_{n+1} and its derivative bit for
    bit those of one launch per stage + the fused end sweep, the sum of squares (other
    workgroups, another summation tree) to 1e-13; chains of every length at the end,
    and the step that is one chain from start to end (FIRST wins, the end separate)"""
    monkeypatch.setenv("ESQ_RKC_FORCE", "1")
    for m in (2, 3, 4, 5, 6, 7, 8, 10, 23):
        monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
        y1, f1, e1, n1 = _end_run(N, m)
        monkeypatch.delenv("ESQ_RKC_DEPTH")
        y2, f2, e2, n2 = _end_run(N, m)
        monkeypatch.setenv("ESQ_RKC_LAST", "0")
        y3, f3, e3, n3 = _end_run(N, m)
        monkeypatch.delenv("ESQ_RKC_LAST")
        assert np.isfinite(y1).all() and np.isfinite(f1).all() and e1 > 0.0
        for y, f, e, tag in ((y2, f2, e2, "LAST"), (y3, f3, e3, "ESQ_RKC_LAST=0")):
            np.testing.assert_array_equal(y, y1, err_msg=f"m = {m}, {tag}")
            np.testing.assert_array_equal(f, f1, err_msg=f"m = {m}, {tag}")
            assert abs(e - e1) <= 1e-13 * e1, (m, tag, e, e1)
        assert not any(k.endswith("-end") for k in n1 + n3), (n1, n3)
        if m >= 7:                       # a last chain that is not the first one
            assert any(k.endswith("-end") for k in n2), n2


@pytest.mark.parametrize("N,rows", [(40, 5), (130, 12), (512, 0)])
def test_rkc_chain_2d_takes_the_end_of_the_step_along(monkeypatch, N, rows):
    """the same on the 2-D heat plugin (csrc/esq_rkc2d.hpp, LAST form): y_{n+1} and
    its derivative bit for bit, the sum of squares to 1e-13"""
    if rows:
        monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    for m in (2, 3, 4, 5, 6, 7, 8, 10, 12, 23):
        monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
        y1, f1, e1, n1 = _end_run(N, m, heat=True)
        monkeypatch.delenv("ESQ_RKC_DEPTH")
        y2, f2, e2, n2 = _end_run(N, m, heat=True)
        monkeypatch.setenv("ESQ_RKC_LAST", "0")
        y3, f3, e3, n3 = _end_run(N, m, heat=True)
        monkeypatch.delenv("ESQ_RKC_LAST")
        assert np.isfinite(y1).all() and np.isfinite(f1).all() and e1 > 0.0
        for y, f, e, tag in ((y2, f2, e2, "LAST"), (y3, f3, e3, "ESQ_RKC_LAST=0")):
            np.testing.assert_array_equal(y, y1, err_msg=f"m = {m}, {tag}")
            np.testing.assert_array_equal(f, f1, err_msg=f"m = {m}, {tag}")
            assert abs(e - e1) <= 1e-13 * e1, (m, tag, e, e1)
        assert not any(k.endswith("-end") for k in n1 + n3), (n1, n3)
        if m in (8, 10, 12):             # (a last chain of 2 .. 5 stages behind the first one)
            assert any(k.endswith("-end") for k in n2), (m, n2)


def _profiled_kernels(s, m):
    """kernel labels of one more run of the stages (the launch plan, by name)"""
    from extensisq_amd._lib import PROF_RKC
    s._dev.profile_reset()
    s._dev.profile_enable([PROF_RKC])
    rho = s.rho_jac(0.0, None)
    s._stages(0.0, (m * m - 1) / (1.54 * rho), m)
    s._dev.profile_enable(None)
    return s._dev.profile_kernels()


def test_rkc_chain_plan_and_whole_steps(monkeypatch):
    """whole adaptive steps (first stage, chains, fused tail, controller) with
    chain sweeps against one launch per stage: identical t, y, error norms and
    counters; and the launch plan of m = 100: 99 stages = 24 chains of 4 + one of
    3, the last one without its second output"""
    N = 48
    rhs = esq.Diffusion3D(N)
    rho = rhs.spectral_radius()
    h0 = (100 ** 2 - 1) / (1.54 * rho) * 0.999
    kw = dict(rtol=1e-3, atol=1e-3, const_jac=True, first_step=h0,
              rho_jac=lambda t, y: rho)
    y0 = pb.diff3d_y0(N)
    a = esq.SSV2stab(rhs, 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
    b = esq.SSV2stab(esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_DEPTH")
    for _ in range(3):
        assert a.step() is None and b.step() is None
        # (the end of the step in the last chain sweep sums the error estimate over
        # other workgroups than the separate sweep: equal to rounding, and so the
        # step sizes the controller derives from it)
        assert abs(a.errold - b.errold) <= 1e-13 * b.errold
        assert abs(a.t - b.t) <= 1e-13 * b.t and abs(a.absh - b.absh) <= 1e-13 * b.absh
        np.testing.assert_allclose(a.y, b.y, rtol=1e-11, atol=1e-13)
    assert a.nfev == b.nfev and int(dev_rkc.maxm[()]) == 100
    # ... and bit for bit with that form switched off
    monkeypatch.setenv("ESQ_RKC_LAST", "0")
    a0 = esq.SSV2stab(esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_LAST")
    monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
    b0 = esq.SSV2stab(esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_RKC_DEPTH")
    for _ in range(3):
        assert a0.step() is None and b0.step() is None
        assert a0.t == b0.t and a0.errold == b0.errold and a0.absh == b0.absh
        np.testing.assert_array_equal(a0.y, b0.y)
    tab = {k[0]: k[2] for k in _profiled_kernels(a, 100)}
    assert tab == {"rkc_chain4-first": 1, "rkc_chain4": 23, "rkc_chain3-last": 1}, tab
    tab = {k[0]: k[2] for k in _profiled_kernels(a, 6)}       # 5 = 3 + 2
    assert tab == {"rkc_chain3-first": 1, "rkc_chain2-last": 1}, tab
    tab = {k[0]: k[2] for k in _profiled_kernels(b, 6)}
    assert tab == {"k_rkc_first": 1, "rhs_rkc": 5}, tab


@pytest.mark.parametrize("case", ["heat8", "heat130", "diff12"])
def test_pde_steps_golden_rkc(golden_dir, case):
    """SSV2stab with the device plugins (sweep + Chebyshev recursion in one kernel)
    directly against the real reference's two steps (tools/gen_golden.py::gen_pde)"""
    g = np.load(os.path.join(golden_dir, "pde_steps.npz"))
    if case == "diff12":
        rhs, y0, rho = esq.Diffusion3D(12), pb.diff3d_y0(12), 12.0 * 13 ** 2
    else:
        N = 8 if case == "heat8" else 130
        rhs, y0, rho = esq.Heat2D(N), pb.heat2d_y0(N), pb.heat2d_rho(N)
    s = esq.SSV2stab(rhs, 0.0, y0, 1.0, rtol=1e-4, atol=1e-7, first_step=40.0 / rho,
                     rho_jac=lambda t, y: rho, const_jac=True)
    for _ in range(2):
        assert s.step() is None
    key = f"{case}/SSV2stab"
    assert s.nfev == int(g[key + "/nfev"])
    assert int(dev_rkc.maxm[()]) == int(g[key + "/maxm"])
    assert_allclose(s.t, float(g[key + "/t"]), rtol=1e-12)
    y = s.y if y0.size <= 2048 else s.y[::97]
    assert_allclose(y, g[key + "/y"], rtol=1e-11, atol=1e-14)


def test_lockstep_two_shards_with_y_dependent_spectral_radius():
    """SSV2stab in a lock-step batch on the real kernels: two solvers (two
    contexts on this GPU, one thread each) with a y-dependent `rho_jac`; the
    batch must use the larger bound and the summed error norm, i.e. reproduce
    the oracle's run on the concatenated state (reference sommeijer.py:174-204
    evaluates `rho_jac` on the whole state)"""
    import threading
    N, world = 12, 2
    n = N * N

    def rho_jac(t, y):
        return 300.0 * (1.0 + float(np.max(np.abs(y))))

    y0s = [(1.0 + r) * pb.heat2d_y0(N, seed=40 + r) for r in range(world)]
    slots = [0.0] * world
    barrier = threading.Barrier(world)
    local = threading.local()

    def reducer(values, op):
        out = []
        for v in values:
            slots[local.rank] = v
            barrier.wait()
            out.append(sum(slots) if op == "sum" else max(slots) if op == "max"
                       else min(slots))
            barrier.wait()
        return out

    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            local.rank = rank
            grp = esq.LockstepGroup(None, world * n, reduce_scalars=reducer)
            grp.debug = True                 # cross-check (t, h, m) every step
            s = esq.SSV2stab(esq.Heat2D(N), 0.0, y0s[rank], 2e-3, rtol=1e-4,
                             atol=1e-7, rho_jac=rho_jac, first_step=1e-5,
                             lockstep=grp)
            ts = []
            while s.status == "running":
                assert s.step() is None
                ts.append(s.t)
            results[rank] = (ts, s.y, s.nfev)
        except BaseException as exc:       # noqa: BLE001
            errors.append(exc)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    assert results[0][0] == results[1][0]
    f1 = pb.heat2d_rhs(N)

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(world)])

    ref = rkc_oracle.SSV2stab(fun, 0.0, np.concatenate(y0s), 2e-3, rtol=1e-4,
                              atol=1e-7, rho_jac=rho_jac, first_step=1e-5)
    ts = []
    while ref.status == "running":
        assert ref.step() is None
        ts.append(ref.t)
    assert_allclose(results[0][0], ts, rtol=1e-9)
    assert_allclose(np.concatenate([r[1] for r in results]), ref.y, rtol=1e-8,
                    atol=1e-12)
    assert results[0][2] == ref.nfev


def _run_lockstep_threads(world, make_solver):
    """`world` solvers of this process (one context + stream + thread each) in
    lock-step through a host reducer standing in for RCCL; returns
    [(ts, y, nfev, extra)] per rank"""
    import threading
    slots = [0.0] * world
    barrier = threading.Barrier(world)
    local = threading.local()

    def reducer(values, op):
        out = []
        for v in values:
            slots[local.rank] = v
            barrier.wait()
            out.append(sum(slots) if op == "sum" else max(slots) if op == "max"
                       else min(slots))
            barrier.wait()
        return out

    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            local.rank = rank
            s = make_solver(rank, reducer)
            ts = []
            while s.status == "running":
                assert s.step() is None
                ts.append(s.t)
            results[rank] = (ts, s.y, s.nfev, s)
        except BaseException as exc:       # noqa: BLE001
            errors.append(exc)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    return results


def test_lockstep_two_shards_power_iteration_and_estimated_first_step():
    """host-reducer lock-step with NOTHING handed to the solvers: `rho_jac=None`
    (nonlinear power iteration, sommeijer.py:331-398) and `first_step=None`
    (`_init_step_size`, :147-160).  Every norm of both procedures is a norm of
    the WHOLE batch, so both shards must reproduce the oracle's run on the
    concatenated state -- same steps, same nfev, same spectral-radius
    evaluations -- instead of silently leaving lock-step with shard-local
    norms divided by the total size."""
    N, world = 12, 2
    n = N * N
    y0s = [(1.0 + r) * pb.heat2d_y0(N, seed=40 + r) for r in range(world)]
    offsets = [0, n]

    def make(rank, reducer):
        grp = esq.LockstepGroup(None, world * n, reduce_scalars=reducer,
                                offset=offsets[rank])
        grp.debug = True                 # cross-check (t, h, m) every step
        return esq.SSV2stab(esq.Heat2D(N), 0.0, y0s[rank], 2e-3, rtol=1e-4,
                            atol=1e-7, lockstep=grp)

    results = _run_lockstep_threads(world, make)
    nfesig_dev = int(dev_rkc.nfesig[()])
    assert results[0][0] == results[1][0]                  # bitwise lock-step
    f1 = pb.heat2d_rhs(N)

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(world)])

    ref = rkc_oracle.SSV2stab(fun, 0.0, np.concatenate(y0s), 2e-3, rtol=1e-4,
                              atol=1e-7)
    ts = []
    while ref.status == "running":
        assert ref.step() is None
        ts.append(ref.t)
    assert len(results[0][0]) == len(ts)
    assert_allclose(results[0][0], ts, rtol=1e-7)
    assert_allclose(np.concatenate([r[1] for r in results]), ref.y, rtol=1e-6,
                    atol=1e-10)
    assert results[0][2] == ref.nfev and results[1][2] == ref.nfev
    # both threads count into the module-level counter
    assert nfesig_dev == world * int(rkc_oracle.nfesig[()])


@pytest.mark.parametrize("plugin,N", [("heat", 6), ("heat", 130), ("diff3d", 5),
                                      ("diff3d", 24), ("diff3d", 41)])
def test_rkc_fused_tail_matches_unfused(monkeypatch, plugin, N):
    """the end of a Chebyshev step in ONE sweep (ESQ_EPI_RKCERR: f(t+h, y), the
    estimate 0.8(yn - y) + 0.4h(fn + f) and its weighted partial sums,
    sommeijer.py:214-220) against RHS launch + error kernel: states and
    derivatives bit-identical, the error norm to rounding (its partial sums are
    grouped by the sweep's workgroups)"""
    if plugin == "heat":
        mk, y0 = (lambda: esq.Heat2D(N)), pb.heat2d_y0(N)
    else:
        mk, y0 = (lambda: esq.Diffusion3D(N)), pb.diff3d_y0(N)
    rho = mk().spectral_radius()
    h = 100.0 / rho                       # m ~ 13: the tail is ~10 % of a step
    kw = dict(rtol=1e-2, atol=1e-2, const_jac=True, rho_jac=lambda t, y: rho,
              first_step=h, max_step=h)
    a = esq.SSV2stab(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_FUSE", "stage,block,solerr,errnorm")     # no rkcerr
    b = esq.SSV2stab(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_FUSE")
    for _ in range(4):
        # the controllers see error norms that differ in the last bits: hand b
        # the scalars a's controller produced so that both take the same step
        b.absh, b.hold = a.absh, a.hold
        if a.hold is not None:
            b.errold = a.errold
        assert a.step() is None and b.step() is None
        assert a.t == b.t
        np.testing.assert_array_equal(a.y, b.y)
        fa = a._dev.download(SLOT_K, a._r["fn"])
        fb = b._dev.download(SLOT_K, b._r["fn"])
        np.testing.assert_array_equal(fa, fb)
        assert_allclose(a.errold, b.errold, rtol=1e-12)
    assert a.nfev == b.nfev


@pytest.mark.parametrize("mode", ["stored", "t_eval", "dense"])
def test_solve_ivp_with_deferred_states_rkc(monkeypatch, mode):
    """SSV2stab through plain `solve_ivp` on a state of 16 MB: `solver.y` as scipy's loop
    reads it is a deferred mirror (extensisq_amd/lazy.py) -- kept per step, ignored
    (t_eval / dense_output): results bit-identical to immediate downloads, and the
    mirrors of a direct reader are plain arrays"""
    from extensisq_amd.lazy import LazyState
    N = 128
    rhs = esq.Diffusion3D(N)
    rho = rhs.spectral_radius()
    y0 = pb.diff3d_y0(N)
    tf = 6e-4
    kw = dict(rtol=1e-3, atol=1e-3, const_jac=True, rho_jac=lambda t, y: rho)
    if mode == "t_eval":
        kw["t_eval"] = [0.4 * tf, tf]
    elif mode == "dense":
        kw["dense_output"] = True
    seen = []

    class Spy(esq.SSV2stab):
        def _step_impl(self):
            out = super()._step_impl()
            seen.append((self._lazy_on, self._lazy_eager))
            return out

    got = solve_ivp(esq.Diffusion3D(N), (0.0, tf), y0, method=Spy, **kw)
    assert seen and all(on for on, _e in seen)
    if mode == "stored" and len(seen) > 3:
        assert seen[-1][1]                     # the copies run beside the steps by then
    monkeypatch.setenv("ESQ_LAZY_Y", "0")
    ref = solve_ivp(esq.Diffusion3D(N), (0.0, tf), y0, method=esq.SSV2stab, **kw)
    assert got.success and ref.success and got.nfev == ref.nfev
    np.testing.assert_array_equal(got.t, ref.t)
    np.testing.assert_array_equal(got.y, ref.y)
    if mode == "dense":
        tc = np.linspace(0.0, tf, 5)
        np.testing.assert_array_equal(got.sol(tc), ref.sol(tc))
    monkeypatch.delenv("ESQ_LAZY_Y")
    s = esq.SSV2stab(esq.Diffusion3D(N), 0.0, y0, tf, **{k: v for k, v in kw.items()
                                                        if k not in ("t_eval", "dense_output")})
    assert s.step() is None and isinstance(s.y, np.ndarray) and not isinstance(s.y, LazyState)


def _ahead_stats(solver):
    used, dropped = C.c_long(), C.c_long()
    solver._chk(solver._lib.esq_rk_launch_ahead_stats(solver._ctx, C.byref(used),
                                                      C.byref(dropped)),
                "esq_rk_launch_ahead_stats")
    return used.value, dropped.value


@pytest.mark.parametrize("plugin,N", [("diff3d", 57), ("heat", 130)])
def test_rkc_opening_sweep_launched_ahead_is_bit_identical(plugin, N):
    """A run at max_step with a constant spectral radius: the next step's opening chain
    sweep goes into the queue behind the step's final sum (esq_rkc_guess_next) and the
    next esq_rkc_stages_end takes it up -- same states, same counters as with
    launch_ahead=0 (ref sommeijer.py:162-271: the reference has no such thing, the
    results are the plain sequence's).  Then a step that does NOT follow the guess (the
    state replaced, a smaller step): the sweep is dropped, the result is the plain one."""
    from extensisq_amd import workloads as wl
    rhs = esq.Diffusion3D(N) if plugin == "diff3d" else esq.Heat2D(N)
    if plugin == "diff3d":                            # smooth data: no rejections
        y0 = wl.diff3d_y0(N)
    else:
        x = (np.arange(N) + 1.0) / (N + 1.0)
        y0 = np.outer(np.sin(np.pi * x), np.sin(np.pi * x)).ravel()
    rho = rhs.spectral_radius()
    h = 400.0 / rho                                   # about 25 stages

    def make(**kw):
        return esq.SSV2stab(rhs, 0.0, y0, 1.0, first_step=h, max_step=h, rtol=1e-2, atol=1e-2,
                            rho_jac=lambda t, y: rho, const_jac=True, **kw)

    a, b = make(), make(esq_options={"launch_ahead": 0})
    for _ in range(6):
        assert a.step() is None and b.step() is None
        assert a.t == b.t
        assert np.array_equal(np.asarray(a.y), np.asarray(b.y))
    used, dropped = _ahead_stats(a)
    assert used >= 4 and dropped == 0, (used, dropped)
    assert _ahead_stats(b) == (0, 0)
    assert a.nfev == b.nfev
    # the guess fails: another state arrives between the steps
    y1 = np.asarray(a.y) * 0.5
    a.y = y1
    b.y = y1
    assert a.step() is None and b.step() is None
    assert np.array_equal(np.asarray(a.y), np.asarray(b.y))
    # ... and a step size below max_step (no guess is made, none is taken)
    c = esq.SSV2stab(rhs, 0.0, y0, 1.0, first_step=0.5 * h, max_step=h, rtol=1e-2, atol=1e-2,
                     rho_jac=lambda t, y: rho, const_jac=True)
    d = esq.SSV2stab(rhs, 0.0, y0, 1.0, first_step=0.5 * h, max_step=h, rtol=1e-2, atol=1e-2,
                     rho_jac=lambda t, y: rho, const_jac=True, esq_options={"launch_ahead": 0})
    for _ in range(5):
        assert c.step() is None and d.step() is None
        assert c.t == d.t and np.array_equal(np.asarray(c.y), np.asarray(d.y))


def test_rkc_opening_sweep_ahead_survives_a_rejection():
    """an attempt at max_step that is REJECTED after its successor's opening sweep went
    into the queue: the retry runs from (y_n, f_n) as if nothing had been launched"""
    from extensisq_amd import workloads as wl
    N = 57
    rhs = esq.Diffusion3D(N)
    y0 = wl.diff3d_y0(N)
    rho = rhs.spectral_radius()
    h = 400.0 / rho
    rough = np.random.default_rng(4).standard_normal(rhs.n) * 50.0

    def run(**kw):
        s = esq.SSV2stab(rhs, 0.0, y0, 1.0, first_step=h, max_step=h, rtol=1e-2, atol=1e-2,
                         rho_jac=lambda t, y: rho, const_jac=True, **kw)
        for _ in range(3):                            # smooth data: the run sits at max_step
            assert s.step() is None
        assert s.absh == s.max_step
        s.y = rough                                   # ... and now the next attempt fails
        before = int(dev_rkc.nrejct)
        for _ in range(3):
            assert s.step() is None
        return s, int(dev_rkc.nrejct) - before

    a, rej_a = run()
    ta, ya = a.t, np.asarray(a.y).copy()
    b, rej_b = run(esq_options={"launch_ahead": 0})
    assert rej_a == rej_b and rej_a >= 1
    assert ta == b.t and np.array_equal(ya, np.asarray(b.y))
    assert a.nfev == b.nfev
    used, dropped = _ahead_stats(a)
    assert used >= 1 and dropped >= 1, (used, dropped)    # the rejected attempt's successor
    assert _ahead_stats(b) == (0, 0)
