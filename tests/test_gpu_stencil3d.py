"""GPU: csrc/esq_stencil3d.hpp with USER functors -- the reference's own two demo
problems (docs/Demo_SSV2stab.ipynb) as device plugins
(examples/ssv2stab_demo_plugins.hip, compiled here with hipcc):

* the right-hand sides against their NumPy twins (oracle/problems.py); the twins call
  NumPy's tanh / exp / power, the plugins the device's: agreement to a few ulp of the
  largest term, not bit for bit (the Laplacian part alone IS bit for bit);
* every fused epilogue kind and the Chebyshev stage entry on the generic sweep (any
  boundary condition, several pointwise-coupled fields) bit-identical to the
  entry-free run (ESQ_CHAIN=0: esq_rhs_fn + the library's own kernels);
* the published integer tables of both problems (docs/Demo_SSV2stab.ipynb:350-356,
  207-211) in device-RHS mode through plain solve_ivp;
* explicit pairs on both plugins against the oracle's step."""
import json
import os
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal
from scipy.integrate import solve_ivp

import extensisq_amd as esq
from extensisq_amd import sommeijer as dev_rkc
from oracle import problems as pb
from oracle import rk_oracle, rkc_oracle

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import examples.ssv2stab_demo_plugins as demo  # noqa: E402

pytestmark = pytest.mark.gpu


def _state(n, seed, lo=0.0):
    rng = np.random.default_rng(seed)
    return lo + rng.random(n)


@pytest.mark.parametrize("N", [5, 16, 39])
def test_tanh_heat_rhs_matches_the_twin(N):
    rhs, y0, _rho = demo.tanh_heat(N)
    fun, y0_twin, _ = pb.tanh3d_problem(N)
    assert_allclose(y0, y0_twin, rtol=0, atol=0)
    for t, y in ((0.0, y0), (0.3, _state(N ** 3, N) - 0.5)):
        got, want = rhs(t, y), fun(t, y)
        # (the largest term of the sum: (N + 1)^2 * 6 |u| and the source's 362.5)
        scale = 6.0 * (N + 1) ** 2 * np.abs(y).max() + 400.0
        assert_allclose(got, want, rtol=0, atol=8 * np.finfo(float).eps * scale)
    # the Laplacian with its time-dependent Dirichlet data is exact where tanh is:
    # t and the boundary such that every ghost value is tanh(0) ... not reachable;
    # instead: a state equal to the exact solution on the grid gives the PDE residual
    x = np.linspace(0.0, 1.0, N + 2)
    X, Y, Z = np.meshgrid(x, x, x)
    ex = np.tanh(5 * X + 10 * Y + 7.5 * Z - (2.5 + 5 * 0.2))[1:-1, 1:-1, 1:-1].reshape(-1)
    assert_allclose(rhs(0.2, ex), fun(0.2, ex), rtol=0,
                    atol=8 * np.finfo(float).eps * (6.0 * (N + 1) ** 2 + 400.0))


@pytest.mark.parametrize("N", [4, 13, 40])
def test_combustion_rhs_matches_the_twin(N):
    rhs, y0 = demo.combustion(N)
    fun, y0_twin = pb.combustion3d_problem(N)
    assert_equal(y0, y0_twin)
    n3 = N ** 3
    # (c in (0, 1], T in [1, 2): the reaction term stays finite)
    y = np.concatenate([_state(n3, N), _state(n3, N + 1, lo=1.0)])
    for t, state in ((0.0, y0), (0.1, y)):
        got, want = rhs(t, state), fun(t, state)
        scale = 6.0 * (N + 0.5) ** 2 * np.abs(state).max() + np.abs(want).max()
        assert_allclose(got, want, rtol=0, atol=8 * np.finfo(float).eps * scale)
    # no reaction (c = 0): the two Laplacians with mirror / Dirichlet faces, bit for bit
    y[:n3] = 0.0
    got, want = rhs(0.0, y), fun(0.0, y)
    assert_equal(got[:n3], want[:n3])
    assert_equal(got[n3:], want[n3:])


@pytest.mark.parametrize("problem", ["tanh", "combustion"])
@pytest.mark.parametrize("name", ["Pr8", "Ts5", "BS5", "Pr9"])
def test_explicit_pairs_on_the_generic_sweep(monkeypatch, problem, name):
    """every fused epilogue kind (stage argument, blocked accumulation, solution +
    error sums, FSAL error norm) on k_stencil3d_points -- one and two fields, ghost
    values from the functor -- bit-identical to the entry-free run and within the
    single-step bounds of the oracle on the NumPy twin"""
    from oracle.tolerances import check_step
    if problem == "tanh":
        N = 21
        rhs, y0, _ = demo.tanh_heat(N)
        plain_rhs, _y, _ = demo.tanh_heat(N)
        twin = pb.tanh3d_problem(N)[0]
        rho = 12.0 * (N + 1) ** 2
    else:
        N = 16
        rhs, y0 = demo.combustion(N)
        plain_rhs, _y = demo.combustion(N)
        twin = pb.combustion3d_problem(N)[0]
        # (temperatures at which the reaction is slower than the diffusion)
        y0 = np.concatenate([_state(N ** 3, 3), 1.0 + 0.05 * _state(N ** 3, 4)])
        rho = 12.0 * (N + 0.5) ** 2 / 0.9 + 1e3
    h = 0.25 / rho
    # (tolerances at which no attempt is rejected: every run takes bitwise the same h)
    tol = 1e-1
    kw = dict(first_step=h, max_step=h, rtol=tol, atol=tol, nfev_stiff_detect=0)
    cls = getattr(esq, name)
    fused = cls(rhs, 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_CHAIN", "0")
    plain = cls(plain_rhs, 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN")
    plain._prelaunch = False
    o = rk_oracle.METHODS[name](twin, 0.0, y0, 1.0, **kw)
    y_old = o.y
    assert fused.step() is None and plain.step() is None and o.step() is None
    row_sum = max(1.0, float(np.abs(cls.A).sum(axis=1).max()))
    assert int(esq.NFS[()]) == 0 and int(rk_oracle.NFS[()]) == 0
    check_step(fused, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous, tol, tol,
               k_rtol=1e-12, lipschitz=rho * row_sum)
    from extensisq_amd._lib import PROF_RHS, PROF_SOLERR, PROF_STAGE
    fused._dev.profile_reset()
    fused._dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR])
    for _ in range(2):
        assert fused.step() is None and plain.step() is None
        assert fused.t == plain.t
        assert_allclose(fused.error_norm_old, plain.error_norm_old, rtol=1e-11)
    fused._dev.profile_enable(None)
    labels = {row[0].split("<")[0] for row in fused._dev.profile_kernels()}
    assert "rhs+stage" in labels, labels
    assert_equal(np.asarray(fused.y), np.asarray(plain.y))
    assert_equal(fused.K, plain.K)
    assert fused.nfev == plain.nfev


@pytest.mark.parametrize("problem", ["tanh", "combustion"])
def test_chebyshev_stage_entry_on_the_generic_sweep(monkeypatch, problem):
    """SSV2stab on the user plugins: the stage sweep with the three-term recursion
    (esq_rhs_rkc_fn) and the end of the step (ESQ_EPI_RKCERR) against two kernels per
    stage (ESQ_RKC_CHAIN=0) -- states bit for bit"""
    if problem == "tanh":
        mk = lambda: demo.tanh_heat(17)                    # noqa: E731
        rhs, y0, rho_jac = mk()
        kw = dict(rho_jac=rho_jac, const_jac=True)
        span = 0.05
    else:
        mk = lambda: demo.combustion(12) + (None,)         # noqa: E731
        rhs, y0, _ = mk()
        kw = {}
        span = 0.3
    a = esq.SSV2stab(rhs, 0.0, y0, span, rtol=1e-5, atol=1e-5, **kw)
    monkeypatch.setenv("ESQ_RKC_CHAIN", "0")
    b = esq.SSV2stab(mk()[0], 0.0, y0, span, rtol=1e-5, atol=1e-5, **kw)
    monkeypatch.delenv("ESQ_RKC_CHAIN")
    for _ in range(5):
        assert a.step() is None and b.step() is None
        assert a.t == b.t and a.errold == b.errold
        assert_equal(np.asarray(a.y), np.asarray(b.y))
    assert a.nfev == b.nfev and a.nfev > 10


@pytest.mark.parametrize("N,planes", [(13, 3), (40, 0), (64, 5)])
def test_user_functor_with_the_fast_sweeps_and_chain_sweeps(monkeypatch, N, planes):
    """a one-field homogeneous functor compiled by the user (anisotropic diffusion): its
    RHS bit for bit its NumPy twin (16-byte pair sweep, odd and even N); explicit pairs
    with the 3-D chain sweeps of the header (esq_chain3d.hpp) bit-identical to one sweep
    per stage and to the entry-free run, within the oracle's single-step bounds;
    SSV2stab's Chebyshev chain sweeps (esq_rkc3d.hpp) bit-identical to one launch per
    stage -- everything `Diffusion3D` has, from a 12-line functor"""
    from oracle.tolerances import check_step
    from extensisq_amd._lib import PROF_RHS, PROF_RKC, PROF_SOLERR, PROF_STAGE
    monkeypatch.setenv("ESQ_RKC_FORCE", "1")
    monkeypatch.setenv("ESQ_RKC_PLANES", str(planes))
    rhs, twin, rho = demo.aniso_diffusion(N)
    rng = np.random.default_rng(N)
    y0 = rng.standard_normal(N ** 3)
    assert_equal(rhs(0.0, y0), twin(0.0, y0))
    h = 1.0 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-1, atol=1e-1, nfev_stiff_detect=0)
    for name in ("Pr8", "Ts5"):
        cls = getattr(esq, name)
        chained = cls(demo.aniso_diffusion(N)[0], 0.0, y0, 1.0, **kw)
        monkeypatch.setenv("ESQ_CHAIN_DEPTH", "1")
        single = cls(demo.aniso_diffusion(N)[0], 0.0, y0, 1.0, **kw)
        monkeypatch.delenv("ESQ_CHAIN_DEPTH")
        monkeypatch.setenv("ESQ_CHAIN", "0")
        bare = cls(demo.aniso_diffusion(N)[0], 0.0, y0, 1.0, **kw)
        monkeypatch.delenv("ESQ_CHAIN")
        o = rk_oracle.METHODS[name](twin, 0.0, y0, 1.0, **kw)
        y_old = o.y
        for s in (chained, single, bare, o):
            assert s.step() is None
        row_sum = max(1.0, float(np.abs(cls.A).sum(axis=1).max()))
        check_step(chained, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous, 1e-1, 1e-1,
                   k_rtol=2e-13, lipschitz=rho * row_sum)
        chained._dev.profile_reset()
        chained._dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR])
        for _ in range(2):
            for s in (chained, single, bare):
                assert s.step() is None
        chained._dev.profile_enable(None)
        labels = [row[0] for row in chained._dev.profile_kernels()]
        assert any(lab.startswith("chain") for lab in labels), labels
        for other in (single, bare):
            assert_equal(np.asarray(chained.y), np.asarray(other.y))
            assert_equal(chained.K, other.K)
    # SSV2stab: chain sweeps of four stages against one launch per stage
    skw = dict(rtol=1e-3, atol=1e-3, const_jac=True, rho_jac=lambda t, y: rho,
               first_step=(30 ** 2 - 1) / (1.54 * rho) * 0.999)
    a = esq.SSV2stab(demo.aniso_diffusion(N)[0], 0.0, y0, 1.0, **skw)
    monkeypatch.setenv("ESQ_RKC_DEPTH", "1")
    b = esq.SSV2stab(demo.aniso_diffusion(N)[0], 0.0, y0, 1.0, **skw)
    monkeypatch.delenv("ESQ_RKC_DEPTH")
    monkeypatch.setenv("ESQ_RKC_LAST", "0")      # (the end of the step: other summation tree)
    a0 = esq.SSV2stab(demo.aniso_diffusion(N)[0], 0.0, y0, 1.0, **skw)
    monkeypatch.delenv("ESQ_RKC_LAST")
    a._dev.profile_enable([PROF_RKC])
    for _ in range(3):
        assert a.step() is None and b.step() is None and a0.step() is None
        assert a0.t == b.t and a0.errold == b.errold
        assert_equal(np.asarray(a0.y), np.asarray(b.y))
        assert abs(a.t - b.t) <= 1e-12 * b.t
        assert_allclose(np.asarray(a.y), np.asarray(b.y), rtol=1e-10, atol=1e-12)
    assert any(row[0].startswith("rkc_chain") for row in a._dev.profile_kernels())


@pytest.mark.parametrize("tol,expect", [
    (1e-1, (6, 1, 402, 132)),      # docs/Demo_SSV2stab.ipynb:350-356
    (1e-2, (15, 4, 729, 85)),
    (1e-3, (27, 2, 786, 40)),
    (1e-4, (57, 0, 1087, 26)),
    (1e-5, (129, 1, 1682, 20)),
    (1e-6, (262, 0, 2445, 12)),
])
def test_published_heat_table_device_rhs(golden_dir, tol, expect):
    """the reference's published table of the 3-D tanh heat problem (n = 59 319) with
    the right-hand side ON THE DEVICE (user plugin): steps (failed) / f-evals / s-max
    exact, accepted times and the solution as the reference's"""
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)[f"tanh3d_tol{tol:.0e}"]
    rhs, y0, rho = demo.tanh_heat(39)
    res = solve_ivp(rhs, (0, 0.7), y0, method=esq.SSV2stab, rtol=tol, atol=tol,
                    const_jac=True, rho_jac=rho)
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
def test_published_combustion_table_device_rhs(golden_dir, tol, expect):
    """the combustion table (n = 128 000, two fields, mirror faces, spectral radius by
    the device-resident power iteration) with the right-hand side on the device"""
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)[f"combustion_tol{tol:.0e}"]
    rhs, y0 = demo.combustion(40)
    res = solve_ivp(rhs, (0, 0.3), y0, method=esq.SSV2stab, rtol=tol, atol=tol)
    nfs = int(dev_rkc.nrejct[()])
    got = (int(res.t.size - 1 + nfs), nfs, int(res.nfev),
           int(dev_rkc.nfesig[()]), int(dev_rkc.maxm[()]))
    assert got == expect
    assert_allclose(res.t, gold["t"], rtol=1e-8)
    assert_allclose(res.y[::4001, -1], gold["y_probe"], rtol=1e-6)
