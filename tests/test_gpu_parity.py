"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called
through the C ABI, against (a) the golden vectors recorded from the real
reference and (b) the CPU oracle on the same seeded inputs.

Tolerances (fp64; SURVEY.md §7 "parity definition"): from identical (t, y, f, h)
K and y_new agree to 1e-13 relative (max-norm, scaled by max|K|), error_norm
and the next step size to 1e-10 relative; trajectories: identical accepted /
rejected step counts and nfev, t_k to 1e-9 relative.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal
from scipy.integrate import solve_ivp

import extensisq_amd as esq
from oracle import problems as pb
from oracle import rk_oracle
from oracle.tolerances import check_step
from tools_cases import bruss1d, compare_trajectory, single_step_cases

pytestmark = pytest.mark.gpu

ERK = ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "CK5", "Me4", "CFMR7osc"]
DEV = {n: getattr(esq, n) for n in ERK}


@pytest.fixture(scope="module")
def single(golden_dir):
    return np.load(os.path.join(golden_dir, "erk_single_step.npz"))


@pytest.fixture(scope="module")
def traces(golden_dir):
    with open(os.path.join(golden_dir, "erk_traces.json")) as fh:
        return json.load(fh)


def test_library_is_the_hip_build():
    from extensisq_amd import _lib
    lib = _lib.load()
    assert lib.esq_abi_version() == _lib.ABI_VERSION
    assert os.path.basename(_lib.LIB_PATH) == "libextensisq_amd.so"


# ------------------------------------------------ golden single-step vectors
@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("pname", list(single_step_cases()))
@pytest.mark.parametrize("direction", ["fwd", "bwd"])
def test_single_step_golden(single, name, pname, direction):
    """host-RHS mode (Python callable): all RK arithmetic on the GPU"""
    fun, t0, y0, h = single_step_cases()[pname]
    sign = 1 if direction == "fwd" else -1
    key = f"{name}/{pname}/{direction}"
    s = DEV[name](fun, t0, y0, t0 + sign * 10.0, first_step=abs(h), rtol=1e-6,
                  atol=1e-9, nfev_stiff_detect=0)
    assert s.step() is None
    if int(single[key + "/nfs"]) == 0:
        assert s.t == float(single[key + "/t_new"])
        check_step(s, single[key + "/K"], single[key + "/y_new"],
                   float(single[key + "/error_norm"]),
                   float(single[key + "/h_abs_next"]),
                   np.asarray(y0, dtype=float), float(single[key + "/h"]),
                   1e-6, 1e-9)
    else:
        # the reference retried with a step derived from the (cancelling)
        # error norm of the rejected attempt: h agrees to ~1e-9 only
        K = single[key + "/K"]
        assert_allclose(s.t, float(single[key + "/t_new"]), rtol=1e-10)
        assert_allclose(s.K, K, rtol=0, atol=1e-8 * np.abs(K).max())
        assert_allclose(s.y, single[key + "/y_new"], rtol=1e-8, atol=1e-12)
        assert_allclose(s.error_norm_old, float(single[key + "/error_norm"]),
                        rtol=1e-5)
    assert s.nfev == int(single[key + "/nfev"])
    assert int(esq.NFS[()]) == int(single[key + "/nfs"])


# -------------------------------------------- device RHS vs the CPU oracle
def _pair(name, dev_rhs, cpu_rhs, t0, y0, tb, **kw):
    d = DEV[name](dev_rhs, t0, y0, tb, **kw)
    o = rk_oracle.METHODS[name](cpu_rhs, t0, y0, tb, **kw)
    return d, o


@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 511, 513, 4099, 70001])
def test_device_rhs_step_sizes(name, n):
    """ragged sizes around the padding / block boundaries, device RHS"""
    rng = np.random.default_rng(n)
    lam = -rng.random(n) * 2.0
    y0 = rng.standard_normal(n)
    kw = dict(first_step=0.05, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    d, o = _pair(name, esq.DiagonalLinear(lam, 1.0),
                 lambda t, y: lam * y + np.sin(t), 0.3, y0, 5.0, **kw)
    # one step from identical input: tight comparison
    y_old = o.y
    assert d.step() is None and o.step() is None
    assert d.t == o.t
    check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-6, 1e-9, k_rtol=2e-13)
    # two more steps.  The controllers see error norms that differ in the last
    # digits (cancelling sum), so the proposed steps would drift apart by ~1e-8
    # relative: pin the device to the oracle's step to keep comparing tightly.
    drifted = False
    for _ in range(2):
        d.h_abs, d.error_norm_old = o.h_abs, o.error_norm_old
        nfs0 = int(rk_oracle.NFS[()])
        assert d.step() is None and o.step() is None
        drifted = drifted or int(rk_oracle.NFS[()]) != nfs0
        if not drifted:                         # no retry inside a step so far
            assert d.t == o.t
            assert_allclose(d.y, o.y, rtol=1e-12, atol=1e-14)
        else:                                   # retried with h from err_norm
            assert_allclose(d.t, o.t, rtol=1e-9)
            assert_allclose(d.y, o.y, rtol=1e-8, atol=1e-11)
        assert_allclose(d.error_norm_old, o.error_norm_old, rtol=1e-4)
    assert d.nfev == o.nfev


@pytest.mark.parametrize("name", ERK)
def test_atol_vector_and_rejections(name):
    """per-component atol + a first step so large that steps are rejected"""
    n = 1000
    rng = np.random.default_rng(5)
    lam = -rng.random(n) * 30.0
    y0 = rng.standard_normal(n)
    atol = 10.0 ** rng.uniform(-10, -6, n)
    kw = dict(first_step=1.0, rtol=1e-7, atol=atol, nfev_stiff_detect=0)
    d, o = _pair(name, esq.DiagonalLinear(lam), lambda t, y: lam * y, 0.0, y0,
                 3.0, **kw)
    nfs = []
    for _ in range(4):
        assert d.step() is None
        nfs.append(int(esq.NFS[()]))
    for k in range(4):
        assert o.step() is None
    assert nfs[-1] == int(rk_oracle.NFS[()]) and nfs[-1] > 0
    assert d.nfev == o.nfev
    assert_allclose(d.t, o.t, rtol=1e-9)
    assert_allclose(d.y, o.y, rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("name,rhs,cpu,y0f,N", [
    ("Ts5", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 37),
    ("Pr8", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 50),
    ("Pr9", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 64),
    ("BS5", esq.Diffusion3D, pb.diff3d_rhs, pb.diff3d_y0, 13),
    ("Pr7", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 31),
    # even grids: the vectorised + chained plugin entries (the kernels the
    # bench runs), incl. the FSAL "last sweep also forms y_new" path, DIRECTLY
    # against the oracle; 130 / 258 span more than one wave tile per grid row
    ("Ts5", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 36),
    ("Ts5", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 258),
    ("Ts5", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 130),
    ("BS5", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 130),
    ("BS5", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 48),
    ("Pr7", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 30),
    ("Pr7", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 258),
    ("Pr8", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 258),
    ("Pr9", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 130),
    ("CK5", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 130),
    ("Me4", esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 36),
    ("CFMR7osc", esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 36),
])
def test_pde_workloads(name, rhs, cpu, y0f, N):
    """the BASELINE.json workloads at small N, 5 steps, device RHS"""
    dev_rhs = rhs(N)
    y0 = y0f(N)
    h = 0.5 / dev_rhs.spectral_radius()
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6,
              nfev_stiff_detect=0)
    d, o = _pair(name, dev_rhs, cpu(N), 0.0, y0, 1.0, **kw)
    y_old = o.y
    assert d.step() is None and o.step() is None
    check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-3, 1e-6, k_rtol=2e-13, lipschitz=dev_rhs.spectral_radius())
    for _ in range(4):      # h is pinned by max_step: states stay comparable
        assert d.step() is None and o.step() is None
        assert d.t == o.t
        assert_allclose(d.K, o.K, rtol=0, atol=1e-11 * np.abs(o.K).max())
        assert_allclose(d.y, o.y, rtol=1e-12, atol=1e-14)
    assert d.nfev == o.nfev
    assert int(esq.NFS[()]) == int(rk_oracle.NFS[()]) == 0


@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("case", ["bruss8", "heat8", "bruss130", "heat130"])
def test_pde_steps_golden(golden_dir, name, case):
    """the device RHS plugins and fused sweeps DIRECTLY against numbers recorded
    from the real reference (tools/gen_golden.py::gen_pde): three fixed steps of
    every ERK method on the benchmark's workloads at even grid sizes"""
    g = np.load(os.path.join(golden_dir, "pde_steps.npz"))
    N = int(case[-3:]) if case.endswith("130") else 8
    if case.startswith("bruss"):
        rhs, y0 = esq.Brusselator2D(N), pb.bruss2d_y0(N)
    else:
        rhs, y0 = esq.Heat2D(N), pb.heat2d_y0(N)
    key = f"{case}/{name}"
    h = float(g[key + "/h"])
    s = DEV[name](rhs, 0.0, y0, 1.0, first_step=h, max_step=h, rtol=1e-3,
                  atol=1e-6, nfev_stiff_detect=0)
    for _ in range(3):
        assert s.step() is None
    small = y0.size <= 512
    assert s.t == float(g[key + "/t"]) and s.nfev == int(g[key + "/nfev"])
    assert int(esq.NFS[()]) == int(g[key + "/nfs"])
    y = s.y if small else s.y[::97]
    K = s.K[:s.n_stages + s.FSAL]
    K = K if small else K[:, ::97]
    assert_allclose(y, g[key + "/y"], rtol=1e-13, atol=1e-15)
    assert_allclose(K, g[key + "/K"], rtol=0, atol=1e-11 * np.abs(g[key + "/K"]).max())
    assert_allclose(s.error_norm_old, g[key + "/err"][-1], rtol=1e-6)


@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("plugin,N,t_end", [("bruss", 10, 1.5), ("heat", 24, 0.02)])
def test_device_rhs_long_trajectory(name, plugin, N, t_end):
    """whole adaptive integrations with a device RHS (fused sweeps, accept-time
    first stage, rejections, the end-of-interval rule) against the oracle with
    the NumPy twin: identical numbers of accepted and rejected steps and RHS
    evaluations, accepted times and final state to the accuracy the (cancelling)
    error norms allow"""
    if plugin == "bruss":
        dev, cpu, y0 = esq.Brusselator2D(N), pb.bruss2d_rhs(N), pb.bruss2d_y0(N)
    else:
        dev, cpu, y0 = esq.Heat2D(N), pb.heat2d_rhs(N), pb.heat2d_y0(N)
    kw = dict(rtol=1e-7, atol=1e-10) if plugin == "bruss" else dict(rtol=1e-5, atol=1e-8)
    got = solve_ivp(dev, (0.0, t_end), y0, method=DEV[name], **kw)
    nfs_dev = int(esq.NFS[()])
    ref = solve_ivp(cpu, (0.0, t_end), y0, method=rk_oracle.METHODS[name], **kw)
    nfs_ref = int(rk_oracle.NFS[()])
    assert got.success and ref.success
    assert got.t.size == ref.t.size and got.t.size > 20
    assert got.nfev == ref.nfev and nfs_dev == nfs_ref
    # these runs are stability-limited: the error norms are rounding-sensitive
    # and the reference itself moves by 1.3e-3 in t_k (3e-5 in y) with its BLAS
    # thread count; the integer counts above are the strong test
    assert_allclose(got.t, ref.t, rtol=5e-2)
    scale = np.abs(ref.y[:, -1]).max()
    assert_allclose(got.y[:, -1], ref.y[:, -1], rtol=0, atol=1e-4 * scale)


@pytest.mark.parametrize("rhs,cpu,y0f,N", [
    (esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 2),
    (esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 4),
    (esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 37),
    (esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 300),
    (esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 514),
    (esq.Heat2D, pb.heat2d_rhs, pb.heat2d_y0, 1030),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 3),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 4),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 6),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 50),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 301),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 514),
    (esq.Brusselator2D, pb.bruss2d_rhs, pb.bruss2d_y0, 1030),
    (esq.Diffusion3D, pb.diff3d_rhs, pb.diff3d_y0, 13),
    (esq.Diffusion3D, pb.diff3d_rhs, pb.diff3d_y0, 40),
])
def test_builtin_rhs_bitwise(rhs, cpu, y0f, N):
    """built-in device RHS == NumPy twin, bit for bit (same operation order,
    library built with -ffp-contract=off)"""
    r = rhs(N)
    y = y0f(N) + 0.01 * np.random.default_rng(N).standard_normal(r.n)
    assert_equal(r(0.0, y), cpu(N)(0.0, y))


# ----------------------------------------------------------- trajectories
CASES = {
    "readme": (lambda t, y: -0.5 * y, [0, 10], [2, 4, 8], {}),
    "duffing": (pb.duffing_rhs, [0.0, 20.0], [0.0, 0.0], {}),
    "duffing_tight": (pb.duffing_rhs, [0.0, 20.0], [0.0, 0.0],
                      dict(rtol=1e-9, atol=1e-12)),
    "rational_fwd": (pb.rational_rhs, [5, 9], [1 / 3, 2 / 9],
                     dict(rtol=1e-3, atol=1e-6)),
    "rational_bwd": (pb.rational_rhs, [5, 1], [1 / 3, 2 / 9],
                     dict(rtol=1e-3, atol=1e-6)),
    "complex": (lambda t, y: -y, [0, 1], [0.5 + 1j],
                dict(rtol=1e-3, atol=1e-6)),
    "bruss1d": (bruss1d()[0], [0, 0.5], bruss1d()[1],
                dict(rtol=1e-6, atol=1e-9)),
}


@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("case", list(CASES))
def test_trajectory_golden(traces, name, case):
    """solve_ivp(method=<device class>) reproduces the reference's run"""
    fun, t_span, y0, kw = CASES[case]
    res = solve_ivp(fun, t_span, y0, method=DEV[name], **kw)
    compare_trajectory(res, int(esq.NFS[()]), traces[name][case],
                       kw.get("rtol", 1e-3), t_rtol=1e-7,
                       noisy=case == "duffing_tight")


def test_published_known_answers():
    """README.md:29-30 example; docs/Demo_BS5.ipynb:137,175 nfev"""
    res = solve_ivp(lambda t, y: -0.5 * y, [0, 10], [2, 4, 8], method=esq.BS5)
    assert res.nfev == 40 and int(esq.NFS[()]) == 0
    assert_allclose(res.t, [0, 0.38027594845942564, 3.5949909992307925,
                            6.671517050825296, 8.335758525412647, 10.0],
                    rtol=1e-10)
    assert_allclose(res.y[:, -1], [0.013389145036386444, 0.026778290072772888,
                                   0.053556580145545776], rtol=1e-11)
    assert solve_ivp(pb.duffing_rhs, [0, 20], [0, 0], method=esq.BS5).nfev == 212
    assert solve_ivp(pb.duffing_rhs, [0, 20], [0, 0], method=esq.Ts5).nfev == 341


@pytest.mark.parametrize("name", ERK)
def test_dense_output_golden(traces, name):
    g = traces[name]["rational_dense"]
    res = solve_ivp(pb.rational_rhs, [5, 9], [1 / 3, 2 / 9], method=DEV[name],
                    dense_output=True)
    assert_allclose(res.sol(np.array(g["tc"])), g["yc"], rtol=1e-10)
    pmax = np.abs(DEV[name].P).max()
    assert_allclose(res.sol(res.t), res.y, rtol=pmax * 1e-15, atol=pmax * 1e-15)


@pytest.mark.parametrize("interp", ["free", "low", "best"])
def test_bs5_interpolants_golden(traces, interp):
    g = traces["BS5"]["dense_" + interp]
    res = solve_ivp(pb.rational_rhs, [5, 9], [1 / 3, 2 / 9], method=esq.BS5,
                    dense_output=True, interpolant=interp)
    assert res.nfev == g["nfev"]
    assert_allclose(res.sol(np.array(g["tc"])), g["yc"], rtol=1e-10)


# --------------------------------------- reference test-suite re-expressions
@pytest.mark.parametrize("name", [n for n in ERK if n != "Me4"])
def test_error_estimation(name):
    """tests/test_rk.py:75-89 (Me4 is excluded there too: its estimate is of
    fifth order)"""
    step = 0.2
    s = DEV[name](lambda t, y: y, 0, [1], 1, first_step=step)
    s.step()
    est = s._estimate_error(s.K, step)
    assert np.abs(s.y - np.exp([step])) < np.abs(est)


@pytest.mark.parametrize("name", ERK)
def test_error_estimation_complex(name):
    """tests/test_rk.py:92-98"""
    h = 0.2
    s = DEV[name](lambda t, y: 1j * y, 0, [1j], 1, first_step=h)
    s.step()
    assert np.isrealobj(s._estimate_error_norm(s.K, h, scale=[1]))


@pytest.mark.parametrize("name", ERK)
def test_integration_rational(name):
    """tests/test_ivp.py:150-213 (both directions, vectorized or not)"""
    rtol, atol, y0 = 1e-3, 1e-6, [1 / 3, 2 / 9]

    def err(y, y_true):
        scale = np.abs(np.atleast_2d(y_true)).max(axis=1)[:, None]
        e = (y - y_true) / (atol + rtol * scale)
        return np.linalg.norm(e, axis=0) / np.sqrt(e.shape[0])

    def fun_vec(t, y):
        return np.vstack((y[1] / t,
                          y[1] * (y[0] + 2 * y[1] - 1) / (t * (y[0] - 1))))
    for vectorized in (False, True):
        for t_span in ([5, 9], [5, 1]):
            res = solve_ivp(fun_vec if vectorized else pb.rational_rhs, t_span,
                            y0, rtol=rtol, atol=atol, method=DEV[name],
                            dense_output=True, vectorized=vectorized)
            assert res.t[0] == t_span[0] and res.success and res.status == 0
            assert res.nfev < 44 and res.njev == 0 and res.nlu == 0
            assert np.all(err(res.y, pb.rational_sol(res.t)) < 5)
            tc = np.linspace(*t_span)
            assert np.all(err(res.sol(tc), pb.rational_sol(tc)) < 5)


@pytest.mark.parametrize("name", ERK)
def test_integration_complex(name):
    """tests/test_ivp.py:216-259"""
    res = solve_ivp(lambda t, y: -y, [0, 1], [0.5 + 1j], method=DEV[name],
                    dense_output=True, rtol=1e-3, atol=1e-6)
    assert res.success and res.status == 0
    assert res.nfev < (40 if name in ("Pr8", "Pr9") else 28)
    y_true = ((0.5 + 1j) * np.exp(-res.t)).reshape(1, -1)
    scale = np.abs(y_true).max()
    assert np.all(np.abs(res.y - y_true) / (1e-6 + 1e-3 * scale) < 5)


@pytest.mark.parametrize("name", ERK)
def test_step_limits_and_failures(name):
    """tests/test_ivp.py:582-665: max_step, first_step, ctor errors, TOO_SMALL"""
    y0 = [1 / 3, 2 / 9]
    cls = DEV[name]
    for t_span in ([5, 9], [5, 1]):
        res = solve_ivp(pb.rational_rhs, t_span, y0, max_step=0.5, method=cls,
                        rtol=1e-3, atol=1e-6, first_step=0.1)
        assert res.t[-1] == t_span[-1] and res.success
        assert np.all(np.abs(np.diff(res.t)) <= 0.5 + 1e-15)
        assert_allclose(0.1, np.abs(res.t[1] - 5))
        with pytest.raises(ValueError):
            cls(pb.rational_rhs, t_span[0], y0, t_span[1], max_step=-1)
        with pytest.raises(ValueError):
            cls(pb.rational_rhs, t_span[0], y0, t_span[1], first_step=-1)
        with pytest.raises(ValueError):
            cls(pb.rational_rhs, t_span[0], y0, t_span[1], first_step=5)
        s = cls(pb.rational_rhs, t_span[0], y0, t_span[1], rtol=1e-3, atol=1e-6,
                max_step=1e-20)
        message = s.step()
        assert s.status == 'failed' and "step size is less" in message
        with pytest.raises(RuntimeError):
            s.step()


@pytest.mark.parametrize("name", ERK)
def test_classes_contract(name):
    """tests/test_ivp.py:838-868 attribute contract; :785-825 corner cases"""
    y0 = [1 / 3, 2 / 9]
    s = DEV[name](pb.rational_rhs, 5, y0, np.inf)
    assert s.n == 2 and s.status == 'running' and s.t_bound == np.inf
    assert s.direction == 1 and s.t == 5 and s.step_size is None
    assert_equal(s.y, y0)
    assert s.nfev > 0 and s.njev >= 0 and s.nlu == 0
    with pytest.raises(RuntimeError):
        s.dense_output()
    assert s.step() is None and s.status == 'running' and s.t > 5
    assert not np.all(np.equal(s.y, y0)) and s.step_size > 0
    assert_allclose(s.dense_output()(5), y0, rtol=1e-15, atol=0)
    # t0 == tf
    sol = solve_ivp(lambda t, y: -y, [4, 4], [2, 3], method=DEV[name],
                    dense_output=True)
    assert_equal(sol.sol(4), [2, 3])
    # empty state
    sol = solve_ivp(lambda t, y: np.zeros((0,)), [0, 10], np.zeros((0,)),
                    method=DEV[name], dense_output=True)
    assert_equal(sol.sol(10), np.zeros((0,)))
    # zero RHS keeps y (tiny_err -> max_factor branch), test_ivp.py:1100-1105
    res = solve_ivp(lambda t, y: np.zeros_like(y), [0, 10], np.ones(3),
                    method=DEV[name])
    assert res.success
    assert_allclose(res.y, 1.0, rtol=1e-15)


def test_nan_propagates_to_failure():
    """a NaN anywhere in the state must fail the step, not be dropped by a
    max-style reduction"""
    n = 5000
    lam = -np.ones(n)
    lam[1234] = np.nan          # one derivative component is NaN
    s = esq.Pr8(esq.DiagonalLinear(lam), 0.0, np.ones(n), 1.0, first_step=0.1)
    message = s.step()
    assert s.status == 'failed' and "Overflow" in message


def test_user_defined_tableau():
    """docs/Demo_own_RK.ipynb contract: a tableau given as class attributes"""
    class Heun(esq.RungeKutta):
        n_stages, order, order_secondary = 2, 2, 1
        A = np.array([[0, 0], [1, 0]])
        B = np.array([0.5, 0.5])
        C = np.array([0, 1])
        E = np.array([0.5, -0.5, 0])
    res = solve_ivp(lambda t, y: -y, [0, 1], [1.0], method=Heun, rtol=1e-4,
                    atol=1e-7)
    assert res.success
    assert_allclose(res.y[0, -1], np.exp(-1), rtol=1e-3)


# ------------------------------------------------ full-size properties
def test_full_size_pr8_step_matches_oracle():
    """BASELINE.json configs[2] size: one Pr8 step, n = 9 999 392, against the
    oracle's step (a few seconds of NumPy)"""
    N = 2236
    y0 = pb.bruss2d_y0(N)
    h = 1.0 / pb.bruss2d_rho(N)
    kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
              nfev_stiff_detect=0)
    d, o = _pair("Pr8", esq.Brusselator2D(N), pb.bruss2d_rhs(N), 0.0, y0, 1.0,
                 **kw)
    y_old = o.y
    assert d.step() is None and o.step() is None
    check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-6, 1e-9, k_rtol=2e-13, lipschitz=pb.bruss2d_rho(N))
    assert d.nfev == o.nfev == 14


@pytest.mark.parametrize("name,plugin,N,plan", [
    ("Pr8", "bruss", 2236, ["chain4+solerr-K<8>", "chain4<4>", "chain5<0>"]),
    ("Pr9", "heat", 2236, None)])
def test_full_size_three_steps_match_oracle(name, plugin, N, plan):
    """BASELINE.json configs[2] / configs[4] sizes, THREE steps against the oracle:
    the first step starts from the constructor's K[0]; from the second on the
    step is the launch sequence bench.py times -- the end-point derivative as
    stage 0 of the first chain sweep, the chain that forms its own input from the
    rows it reads, the K rows nothing reads left unwritten.  Nothing is read from
    the device solver in between (that would evaluate those rows the plain way);
    at the end `K` (restored on demand) and `y` meet the oracle's.  The two runs
    start steps 2 and 3 from states that differ in the last digits, so the
    single-step bounds are widened by the RHS's amplification over two steps."""
    if plugin == "bruss":
        rhs, cpu, y0, rho = (esq.Brusselator2D(N), pb.bruss2d_rhs(N), pb.bruss2d_y0(N),
                             pb.bruss2d_rho(N))
    else:
        rhs, cpu, y0, rho = (esq.Heat2D(N), pb.heat2d_rhs(N), pb.heat2d_y0(N),
                             pb.heat2d_rho(N))
    h = 1.0 / rho
    # (tolerances at which the controller keeps h at max_step: both runs then take
    # bitwise the same step sizes)
    rtol, atol = (1e-6, 1e-9) if plugin == "bruss" else (1e-3, 1e-6)
    kw = dict(first_step=h, max_step=h, rtol=rtol, atol=atol, nfev_stiff_detect=0)
    d, o = _pair(name, rhs, cpu, 0.0, y0, 1.0, **kw)
    from extensisq_amd._lib import PROF_RHS, PROF_SOLERR, PROF_STAGE
    errs = []
    for k in range(3):
        if k == 2:                      # the launch plan of the third step, by name
            d._dev.profile_reset()
            d._dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR])
        assert d.step() is None and o.step() is None
        assert d.t == o.t
        errs.append((d.error_norm_old, o.error_norm_old))
    d._dev.profile_enable(None)
    labels = sorted(row[0] for row in d._dev.profile_kernels())
    assert d.nfev == o.nfev and int(esq.NFS[()]) == 0
    missing = C.c_int()
    d._chk(d._lib.esq_rk_lazy_rows(d._ctx, C.byref(missing), None, None, None, None),
           "esq_rk_lazy_rows")
    if plan is not None:
        assert missing.value > 0          # the rows were NOT in memory until now
    kmax, ymax = np.abs(o.K).max(), np.abs(o.y).max()
    k_atol = 10 * (2e-13 * kmax + 8 * np.finfo(float).eps * rho * ymax)
    assert_allclose(d.y, o.y, rtol=1e-11, atol=h * k_atol)
    assert_allclose(d.K, o.K, rtol=0, atol=k_atol)
    for got, ref in errs:
        assert_allclose(got, ref, rtol=1e-5)
    if plan is not None:
        assert labels == plan


def test_full_size_free_controller_with_rejections_matches_oracle():
    """configs[2] size, the controller left alone from a first step 40 x the
    stability limit: the attempts that are rejected, the retries and the accepted
    steps that follow are the oracle's -- same counts of accepted steps, rejected
    steps and RHS evaluations, same times -- with the chain sweeps doing the
    work (a rejected step re-uses K[0]; its retry starts from the rows)"""
    N = 2236
    y0 = pb.bruss2d_y0(N)
    h_stab = 1.0 / pb.bruss2d_rho(N)
    kw = dict(first_step=40 * h_stab, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    d, o = _pair("Pr8", esq.Brusselator2D(N), pb.bruss2d_rhs(N), 0.0, y0,
                 60 * h_stab, **kw)
    accepted = 0
    while o.status == "running" and accepted < 5:
        assert o.step() is None
        accepted += 1
    nfs_ref = int(rk_oracle.NFS[()])
    for _ in range(accepted):
        assert d.step() is None
    assert int(esq.NFS[()]) == nfs_ref and nfs_ref >= 1
    assert d.nfev == o.nfev
    assert_allclose(d.t, o.t, rtol=1e-9)
    assert_allclose(d.h_abs, o.h_abs, rtol=1e-6)
    assert_allclose(d.y, o.y, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["Pr8", "Ts5"])
def test_free_controller_with_rejections_on_the_3d_plugin_matches_oracle(name):
    """explicit pairs on `Diffusion3D` (N = 96: the 3-D chain sweeps by default), the
    controller left alone from a first step 30 x the stability limit: rejected attempts,
    their retries (K[0] re-used, no first launch ahead of a rejected attempt) and the
    accepted steps that follow are the oracle's -- same counts of accepted steps,
    rejected steps and RHS evaluations, same times"""
    N = 96
    rng = np.random.default_rng(9)
    y0 = pb.diff3d_y0(N) + 1e-2 * rng.standard_normal(N ** 3)
    h_stab = 1.0 / (12.0 * (N + 1) ** 2)
    kw = dict(first_step=30 * h_stab, rtol=1e-5, atol=1e-8, nfev_stiff_detect=0)
    d, o = _pair(name, esq.Diffusion3D(N), pb.diff3d_rhs(N), 0.0, y0, 80 * h_stab, **kw)
    accepted = 0
    while o.status == "running" and accepted < 6:
        assert o.step() is None
        accepted += 1
    nfs_ref = int(rk_oracle.NFS[()])
    d._dev.profile_enable([0, 1, 2])
    for _ in range(accepted):
        assert d.step() is None
    labels = [row[0] for row in d._dev.profile_kernels()]
    assert any(lab.startswith("chain") for lab in labels), labels
    assert int(esq.NFS[()]) == nfs_ref and nfs_ref >= 1
    assert d.nfev == o.nfev
    assert_allclose(d.t, o.t, rtol=1e-9)
    assert_allclose(d.h_abs, o.h_abs, rtol=1e-6)
    assert_allclose(np.asarray(d.y), o.y, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name,mk,y0f", [
    ("Pr8", lambda: esq.Brusselator2D(64), lambda: pb.bruss2d_y0(64)),
    ("Pr9", lambda: esq.Heat2D(96), lambda: pb.heat2d_y0(96)),
    ("Ts5", lambda: esq.Heat2D(96), lambda: pb.heat2d_y0(96))])
def test_assigning_the_state_keeps_the_old_derivative(monkeypatch, name, mk, y0f):
    """`solver.y = value` between steps (events, callbacks, user code): in the
    reference `self.f` stays what it was (common.py:298), so the next step's
    K[0] is the derivative of the OLD state.  With the end-point derivative left
    to the next step's first chain sweep (ESQ_LAZY_END) the library evaluates it
    before the upload changes the state: runs with and without the deferral are
    bit-identical, and K[0] is the stale derivative"""
    monkeypatch.setenv("ESQ_CHAIN_ROWS", "8")          # chains on these small grids
    y0 = y0f()
    rho = mk().spectral_radius()
    h = 0.5 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    a = DEV[name](mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_LAZY_END", "0")
    b = DEV[name](mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_LAZY_END")
    cpu = pb.bruss2d_rhs(64) if name == "Pr8" else pb.heat2d_rhs(96)
    o = rk_oracle.METHODS[name](cpu, 0.0, y0, 1.0, **kw)
    for k in range(3):
        for s in (a, b, o):
            assert s.step() is None
            s.y = (0.5 + 0.25 * k) * s.y           # no K[0] re-evaluation by hand
    for s in (a, b, o):
        assert s.step() is None
    assert a.t == b.t
    assert_allclose(a.t, o.t, rtol=1e-12)
    assert_equal(a.y, b.y)
    assert_equal(a.K, b.K)
    assert a.nfev == b.nfev == o.nfev
    # K[0] of the last step is f(t, y BEFORE the last assignment), as in the oracle
    assert_allclose(a.K[0], o.K[0], rtol=0, atol=1e-9 * np.abs(o.K[0]).max())
    assert_allclose(a.y, o.y, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("N", [13, 40, 64, 159])
@pytest.mark.parametrize("name", ["Pr8", "Ts5", "BS5"])
def test_diffusion3d_erk_fused_sweeps(monkeypatch, name, N):
    """explicit pairs on the 3-D plugin (the reference treats every `fun` alike,
    common.py:353-356; its own demo problems are 3-D): with the plugin's fused
    entry every RHS sweep carries the stage arithmetic that follows it (stage
    argument, blocked accumulation, solution + error sums, FSAL error norm) --
    bit-identical to the one-kernel-per-operation sequence (ESQ_CHAIN=0), the first
    step within the single-step bounds of the oracle, odd and even grids, up to
    the BASELINE.json configs[3] grid N = 159 (n = 4 019 679)"""
    y0 = pb.diff3d_y0(N)
    if N < 100:
        y0 = y0 + 1e-3 * np.random.default_rng(N).standard_normal(N ** 3)
    rho = 12.0 * (N + 1) ** 2
    h = 1.0 / rho
    # (tolerances at which no attempt is rejected: every run takes bitwise the same h)
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    # (one sweep per stage: the chain sweeps have test_diffusion3d_erk_chain_sweeps)
    monkeypatch.setenv("ESQ_CHAIN_DEPTH", "1")
    fused = DEV[name](esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN_DEPTH")
    monkeypatch.setenv("ESQ_CHAIN", "0")
    plain = DEV[name](esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN")
    plain._prelaunch = False
    o = rk_oracle.METHODS[name](pb.diff3d_rhs(N), 0.0, y0, 1.0, **kw)
    y_old = o.y
    assert fused.step() is None and plain.step() is None and o.step() is None
    assert int(esq.NFS[()]) == 0 and int(rk_oracle.NFS[()]) == 0
    check_step(fused, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-3, 1e-6, k_rtol=2e-13, lipschitz=rho)
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
    if name == "Pr8":       # every RHS evaluation of the step is a fused sweep
        assert "rhs_plugin" not in labels and "k_lincomb" not in labels, labels
        assert {"rhs+block", "rhs+solerr"} <= labels, labels
    assert_equal(fused.y, plain.y)
    assert_equal(fused.K, plain.K)
    assert fused.nfev == plain.nfev


@pytest.mark.parametrize("N,planes", [(5, 0), (13, 3), (24, 0), (41, 7), (57, 0), (64, 5),
                                      (70, 16)])
@pytest.mark.parametrize("depth", [2, 3, 4])
def test_diffusion3d_erk_chain_sweeps_are_bit_identical(monkeypatch, N, planes, depth):
    """D consecutive stages per marching sweep on the 3-D plugin (esq_rhs_diff3d_chain,
    csrc/esq_chain3d.hpp) against one sweep per stage: K and y bit for bit over three
    steps (the second and third start with the end-point derivative as stage 0 of
    their first chain, rows are left unwritten and restored for the comparison), the
    error norm to rounding; grids of one and of several patches per plane, odd and
    even, forced tile depths (run-in planes, plane ranges that do not divide N)"""
    from extensisq_amd._lib import PROF_RHS, PROF_SOLERR, PROF_STAGE
    monkeypatch.setenv("ESQ_RKC_FORCE", "1")
    monkeypatch.setenv("ESQ_RKC_PLANES", str(planes))
    rng = np.random.default_rng(100 + N)
    y0 = pb.diff3d_y0(N) + 0.1 * rng.standard_normal(N ** 3)
    h = 1.0 / (12.0 * (N + 1) ** 2)
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    chained = []
    for name in ("Pr8", "Ts5", "BS5", "Pr9", "CK5"):
        monkeypatch.setenv("ESQ_CHAIN_DEPTH", str(depth))
        a = DEV[name](esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
        monkeypatch.setenv("ESQ_CHAIN_DEPTH", "1")
        b = DEV[name](esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
        a._dev.profile_reset()
        a._dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR])
        for k in range(3):
            assert a.step() is None and b.step() is None
            assert a.t == b.t
            assert_allclose(a.error_norm_old, b.error_norm_old, rtol=1e-11)
        a._dev.profile_enable(None)
        labels = [row[0] for row in a._dev.profile_kernels()]
        # (the planner takes a chain where its words x tile amplification cost less)
        chained += [name] if any(lab.startswith("chain") for lab in labels) else []
        assert_equal(np.asarray(a.y), np.asarray(b.y), err_msg=name)
        assert_equal(a.K, b.K, err_msg=name)
        assert a.nfev == b.nfev
    assert "Pr8" in chained and len(chained) >= 3, chained


@pytest.mark.parametrize("name", ["Pr8", "Ts5", "BS5", "Pr9"])
@pytest.mark.parametrize("N", [64, 159])
def test_diffusion3d_erk_chain_sweeps_match_oracle(name, N):
    """the default launch plan of an explicit pair on the 3-D plugin (chain sweeps from
    N = 48) against the oracle's step from the same (t, y, h), and bit-identical to the
    entry-free run (ESQ_CHAIN=0) over three steps; BASELINE.json configs[3]'s grid"""
    import os
    y0 = pb.diff3d_y0(N)
    if N < 100:
        y0 = y0 + 1e-3 * np.random.default_rng(N).standard_normal(N ** 3)
    rho = 12.0 * (N + 1) ** 2
    h = 1.0 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    chained = DEV[name](esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    os.environ["ESQ_CHAIN"] = "0"
    try:
        plain = DEV[name](esq.Diffusion3D(N), 0.0, y0, 1.0, **kw)
    finally:
        del os.environ["ESQ_CHAIN"]
    plain._prelaunch = False
    o = rk_oracle.METHODS[name](pb.diff3d_rhs(N), 0.0, y0, 1.0, **kw)
    y_old = o.y
    assert chained.step() is None and plain.step() is None and o.step() is None
    # (a stage argument y + h sum_j a_ij K_j is rounded at eps * sum_j |a_ij| |h K_j| ~
    # eps * sum_j |a_ij| * |y| where h L ~ 1, and the RHS amplifies that by L)
    row_sum = max(1.0, float(np.abs(DEV[name].A).sum(axis=1).max()))
    check_step(chained, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-3, 1e-6, k_rtol=2e-13, lipschitz=rho * row_sum)
    from extensisq_amd._lib import PROF_RHS, PROF_SOLERR, PROF_STAGE
    chained._dev.profile_reset()
    chained._dev.profile_enable([PROF_STAGE, PROF_RHS, PROF_SOLERR])
    for _ in range(2):
        assert chained.step() is None and plain.step() is None
        assert_allclose(chained.error_norm_old, plain.error_norm_old, rtol=1e-11)
    chained._dev.profile_enable(None)
    labels = [row[0] for row in chained._dev.profile_kernels()]
    assert any(lab.startswith("chain") for lab in labels), labels
    assert_equal(np.asarray(chained.y), np.asarray(plain.y))
    assert_equal(chained.K, plain.K)
    assert chained.nfev == plain.nfev


@pytest.mark.parametrize("name,plugin,N", [("Pr8", "bruss", 2236), ("Ts5", "heat", 1000),
                                           ("Pr9", "heat", 2236)])
def test_full_size_fused_equals_unfused(monkeypatch, name, plugin, N):
    """size-independent property at the BASELINE.json sizes: the fully fused
    step (one kernel per RHS evaluation, ~10 000 workgroup partials in the error
    norm, first stage formed at accept time) and the one-kernel-per-operation
    step give bit-identical states and stage derivatives"""
    if plugin == "bruss":
        mk, y0, rho = (lambda: esq.Brusselator2D(N)), pb.bruss2d_y0(N), pb.bruss2d_rho(N)
    else:
        mk, y0, rho = (lambda: esq.Heat2D(N)), pb.heat2d_y0(N), pb.heat2d_rho(N)
    h = 1.0 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    cls = DEV[name]
    if name == "Pr9":
        monkeypatch.setenv("ESQ_SRC", "1")      # the on-the-fly first stage at full size
    fused = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_SRC", raising=False)
    monkeypatch.setenv("ESQ_CHAIN", "0")
    plain = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN")
    plain._prelaunch = False
    for _ in range(3):
        assert fused.step() is None and plain.step() is None
        assert fused.t == plain.t
        assert_allclose(fused.error_norm_old, plain.error_norm_old, rtol=1e-11)
    assert_equal(fused.y, plain.y)
    s = cls.n_stages
    for row in (1, s // 2, s - 1, s):
        assert_equal(fused._dev.download_last_K(row), plain._dev.download_last_K(row))
    assert fused.nfev == plain.nfev


def test_full_size_linearity_and_exactness():
    """size-independent properties at n = 1e7: for f = lam*y the Pr8 step is
    linear in y0 (step(a*y0) == a*step(y0) with atol scaled alike) and the
    result matches exp(lam*t) to the tolerance"""
    n = 10_000_000
    rng = np.random.default_rng(11)
    lam = -rng.random(n)
    y0 = rng.standard_normal(n)
    rhs = esq.DiagonalLinear(lam)
    outs = []
    for a in (1.0, 4.0):
        s = esq.Pr8(rhs, 0.0, a * y0, 0.5, first_step=0.1, rtol=1e-8,
                    atol=a * 1e-10, nfev_stiff_detect=0)
        while s.status == 'running':
            s.step()
        outs.append((s.y, s.nfev, s.t))
    assert outs[0][1] == outs[1][1]
    assert_equal(4.0 * outs[0][0], outs[1][0])        # exact: power of two
    assert_allclose(outs[0][0], y0 * np.exp(lam * 0.5), rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------ lock-step / RCCL
def test_rccl_single_rank_lockstep():
    """the RCCL leg of the lock-step mode on one GPU: a 1-rank communicator is
    created through the C ABI (dlopen librccl, ncclCommInitRank) and every
    error evaluation goes through ncclAllReduce on the solver's stream; the
    run must equal the communicator-free run bit for bit"""
    from extensisq_amd import lockstep
    N = 64
    y0 = pb.heat2d_y0(N)
    group = lockstep.init_lockstep(0, 1, 0, y0.size)
    assert group.n_total == y0.size
    kw = dict(rtol=1e-6, atol=1e-9, nfev_stiff_detect=0, first_step=1e-6)
    a = esq.Pr9(esq.Heat2D(N), 0.0, y0, 1e-3, lockstep=group, **kw)
    b = esq.Pr9(esq.Heat2D(N), 0.0, y0, 1e-3, **kw)
    for _ in range(6):
        assert a.step() is None and b.step() is None
        assert a.t == b.t and a.error_norm_old == b.error_norm_old
    assert_equal(a.y, b.y)
    r = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1e-3, rtol=1e-4, atol=1e-6,
                     lockstep=group)
    s = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1e-3, rtol=1e-4, atol=1e-6)
    for _ in range(3):
        assert r.step() is None and s.step() is None
        assert r.t == s.t
    assert_equal(r.y, s.y)
    # host scalars through RCCL (esq_allreduce_scalars): the debug cross-check
    # of (t, h[, m]) and the batch maximum of a user spectral-radius bound
    assert lockstep.comm_size(group) == 1
    assert group.allreduce(a._dev, [1.5, -2.0], "max") == [1.5, -2.0]
    assert group.allreduce(a._dev, [1.5, -2.0], "sum") == [1.5, -2.0]
    group.debug = True
    rho = 8.0 * (N + 1) ** 2
    u = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1e-3, rtol=1e-4, atol=1e-6,
                     lockstep=group, rho_jac=lambda t, y: rho + float(np.max(y)))
    v = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 1e-3, rtol=1e-4, atol=1e-6,
                     rho_jac=lambda t, y: rho + float(np.max(y)))
    for _ in range(3):
        assert u.step() is None and v.step() is None and a.step() is None
        assert u.t == v.t and u.sprad == v.sprad
    group.debug = False
    del a, r, u
    lockstep.destroy_lockstep(group)


def test_lockstep_eight_shards_on_one_gpu_equal_the_concatenated_reference(golden_dir):
    """BASELINE.json configs[4] in miniature, on the real kernels: eight Pr9
    solvers (eight contexts and streams on this GPU, one thread each, standing
    in for eight ranks) integrate their own heat problem in lock-step; the
    per-shard error sums of squares are summed by a host reducer where RCCL
    would all-reduce them.  Every shard must take exactly the steps the REAL
    reference took on the concatenated state (tests/golden/lockstep.npz)."""
    import threading
    g = np.load(os.path.join(golden_dir, "lockstep.npz"))
    N, world = int(g["N"]), 8
    n = N * N
    slots = [0.0] * world
    barrier = threading.Barrier(world)
    local = threading.local()

    def reducer(values, op):
        # fixed-order reduction over the eight "ranks" (deterministic)
        assert len(values) == 1
        slots[local.rank] = values[0]
        barrier.wait()
        out = sum(slots) if op == "sum" else max(slots) if op == "max" else min(slots)
        barrier.wait()
        return [out]

    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            local.rank = rank
            y0 = pb.heat2d_y0(N, seed=int(g["seeds"][rank]))
            grp = esq.LockstepGroup(None, world * n, reduce_scalars=reducer)
            s = esq.Pr9(esq.Heat2D(N), 0.0, y0, float(g["t_end"]),
                        first_step=float(g["h0"]), rtol=1e-6, atol=1e-9,
                        nfev_stiff_detect=0, lockstep=grp)
            ts, errs = [], []
            while s.status == "running":
                assert s.step() is None
                ts.append(s.t)
                errs.append(s.error_norm_old)
            results[rank] = (ts, errs, s.y, s.nfev)
        except BaseException as exc:       # noqa: BLE001
            errors.append(exc)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    for ts, errs, _y, nfev in results:
        assert ts == results[0][0]                      # bitwise identical decisions
        assert_allclose(ts, g["t"], rtol=1e-10)
        assert_allclose(errs, g["err"], rtol=1e-6)
        assert nfev == int(g["nfev"]) - 4               # the golden run estimated h0 itself
    assert_allclose(np.concatenate([r[2] for r in results]), g["y_end"], rtol=1e-9,
                    atol=1e-12)


def test_lockstep_total_size_changes_the_norm():
    """n_total of the batch enters the RMS norm (two ranks' worth of elements
    halves the mean square when the other shard contributes nothing)"""
    n = 1000
    lam = -np.ones(n)
    y0 = np.ones(n)
    a = esq.Pr8(esq.DiagonalLinear(lam), 0.0, y0, 1.0, first_step=0.3)
    a.step()
    b = esq.Pr8(esq.DiagonalLinear(lam), 0.0, y0, 1.0, first_step=0.3,
                lockstep=esq.LockstepGroup(None, 2 * n))
    b.step()
    assert_allclose(b.error_norm_old, a.error_norm_old / np.sqrt(2), rtol=1e-12)


# ------------------------------------------------ device-resident interpolant
@pytest.mark.parametrize("name", ERK)
def test_device_dense_output_large_n(name):
    """n >= 4096: Qh = h*K.T@P is formed in one fused pass and stays in HBM,
    evaluations are Horner kernels; compare with the
    oracle's host interpolant of the same step"""
    n = 5000
    rng = np.random.default_rng(17)
    lam = -rng.random(n) * 2.0
    y0 = rng.standard_normal(n)
    kw = dict(first_step=0.05, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    d, o = _pair(name, esq.DiagonalLinear(lam, 1.0),
                 lambda t, y: lam * y + np.sin(t), 0.3, y0, 5.0, **kw)
    assert d.step() is None and o.step() is None
    sd, so = d.dense_output(), o.dense_output()
    from extensisq_amd.common import DeviceHornerDenseOutput
    if name != "BS5":       # BS5's default 'low' interpolant is Horner too
        assert isinstance(sd, DeviceHornerDenseOutput)
    tc = np.linspace(o.t_old, o.t, 5)
    assert_allclose(sd(tc), so(tc), rtol=1e-11, atol=1e-13)
    assert_allclose(sd(tc[2]), so(tc[2]), rtol=1e-11, atol=1e-13)
    assert sd(tc).shape == (n, 5)
    del d                   # the interpolant owns its memory
    assert_allclose(sd(o.t), o.y, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("interp", ["free", "low", "best"])
def test_bs5_interpolants_device_resident(interp):
    n = 6000
    rng = np.random.default_rng(3)
    lam = -rng.random(n)
    y0 = rng.standard_normal(n)
    kw = dict(first_step=0.1, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0,
              interpolant=interp)
    d = esq.BS5(esq.DiagonalLinear(lam), 0.0, y0, 5.0, **kw)
    o = rk_oracle.BS5(lambda t, y: lam * y, 0.0, y0, 5.0, **kw)
    assert d.step() is None and o.step() is None
    sd, so = d.dense_output(), o.dense_output()
    tc = np.linspace(o.t_old, o.t, 4)
    assert_allclose(sd(tc), so(tc), rtol=1e-11, atol=1e-13)
    assert d.nfev == o.nfev


# ------------------------------------------------------- device starting step
@pytest.mark.parametrize("case", ["decay", "duffing", "heat", "zero", "complex",
                                  "backward", "scalar", "atol_vec", "big"])
@pytest.mark.parametrize("mode", ["host_rhs", "device_rhs"])
def test_device_h_start(case, mode):
    """the device-resident starting-step estimate (perturbation vectors in K
    rows, norms by reduction kernels) equals the oracle's `first_step_size`
    (common.py:519-763; pinned against the reference's golden first steps in
    tests/test_oracle_golden.py and tests/test_host_logic.py)"""
    from extensisq_amd.common import validate_tol
    atol = 1e-6
    dev_fun = None
    if case == "decay":
        fun, a, b, y = (lambda t, y: -0.5 * y), 0.0, 10.0, np.array([2., 4., 8.])
    elif case == "duffing":
        fun, a, b, y = pb.duffing_rhs, 0.0, 20.0, np.array([0.0, 0.0])
    elif case == "heat":
        fun, a, b, y = pb.heat2d_rhs(12), 0.0, 1.0, pb.heat2d_y0(12)
        dev_fun = esq.Heat2D(12)
    elif case == "zero":
        fun, a, b, y = (lambda t, y: np.zeros_like(y)), 0.0, 10.0, np.ones(3)
    elif case == "complex":
        fun, a, b, y = (lambda t, y: -y), 0.0, 1.0, np.array([0.5 + 1j, -2j])
    elif case == "backward":
        fun, a, b, y = pb.rational_rhs, 5.0, 1.0, np.array([1 / 3, 2 / 9])
    elif case == "scalar":
        fun, a, b, y = (lambda t, y: np.cos(t) * y), 1.0, 3.0, np.array([0.7])
    elif case == "atol_vec":
        rng = np.random.default_rng(2)
        lam = -rng.random(777) * 5
        fun, a, b, y = (lambda t, y: lam * y + np.sin(t)), 0.5, 2.0, \
            rng.standard_normal(777)
        y[::7] = 0.0
        atol = 10.0 ** rng.uniform(-9, -5, 777)
        dev_fun = esq.DiagonalLinear(lam, 1.0)
    else:
        fun, a, b, y = pb.bruss2d_rhs(64), 0.0, 1.0, pb.bruss2d_y0(64)
        dev_fun = esq.Brusselator2D(64)
    if mode == "device_rhs" and dev_fun is None:
        pytest.skip("no device twin for this RHS")
    for cls in (esq.Ts5, esq.Pr8):
        s = cls(dev_fun if mode == "device_rhs" else fun, a, y, b, rtol=1e-4,
                atol=atol)
        y_arr = np.asarray(y, dtype=complex if np.iscomplexobj(y) else float)
        rtol_v, atol_v = validate_tol(1e-4, atol, y_arr)
        want = abs(rk_oracle.first_step_size(fun, a, b, y_arr, np.asarray(fun(a, y_arr)),
                                             cls.order_secondary, rtol_v, atol_v))
        assert_allclose(s.h_abs, want, rtol=1e-11)


# ----------------------------------------- solve_ivp end to end, device RHS
@pytest.mark.parametrize("name", ["Ts5", "BS5", "Pr8"])
def test_solve_ivp_device_rhs_t_eval_and_events(name):
    """plain `solve_ivp(DeviceRHS, ..., method=<class>, t_eval=..., events=...)`
    (reference tests/test_ivp.py:369-541, 757-782 with a device RHS):
    default first step (device h_start), lazy `solver.y` mirror, device-resident
    interpolant (n = 6400 >= 4096) for t_eval and the event root-finder"""
    N = 80
    y0 = pb.heat2d_y0(N, seed=7)
    t_eval = np.array([0.0, 2e-5, 7e-5, 1e-4])

    def event(t, y):                       # the centre value decays through 0.9
        return y[(N // 2) * N + N // 2] - 0.9
    event.terminal = False
    kw = dict(rtol=1e-5, atol=1e-8, t_eval=t_eval, events=event)
    got = solve_ivp(esq.Heat2D(N), (0.0, 1e-4), y0, method=DEV[name], **kw)
    ref = solve_ivp(pb.heat2d_rhs(N), (0.0, 1e-4), y0,
                    method=rk_oracle.METHODS[name], **kw)
    assert got.success and ref.success
    assert got.nfev == ref.nfev and got.t.shape == ref.t.shape
    assert_allclose(got.y, ref.y, rtol=1e-8, atol=1e-11)
    assert len(got.t_events[0]) == len(ref.t_events[0])
    assert_allclose(got.t_events[0], ref.t_events[0], rtol=1e-7)


# ------------------------------------- large downloads: the process's download stream
@pytest.mark.parametrize("n", [2 * 1048576 + 0, 3 * 1048576 + 1, 3 * 1048576 + 2, 4194304 + 777])
def test_large_downloads_are_the_same_bytes(n):
    """esq_download and esq_snapshot_copy of >= 8 MiB run on the process's download
    stream, by the DMA engines: odd lengths, a destination that is not 16-byte aligned,
    page-locked by the caller (as the warm buffers do) and not"""
    import ctypes as C
    from extensisq_amd import _lib
    from extensisq_amd.device import DeviceContext
    lib = _lib.load()
    rng = np.random.default_rng(n)
    data = rng.standard_normal(n)
    dev = DeviceContext(n, 2)
    dev.upload(_lib.SLOT_Y, 0, data)
    before = _lib.copy_lane_info(0)
    np.testing.assert_array_equal(dev.download(_lib.SLOT_Y, 0), data)
    for shift, pin in ((0, True), (0, False), (1, True)):
        raw = np.full(n + 2, np.nan)
        out = raw[shift:shift + n]                   # shift 1: 8 mod 16
        token = C.c_void_p()
        assert lib.esq_snapshot_begin(dev.handle, _lib.SLOT_Y, 0, C.byref(token)) == 0
        ptr = out.ctypes.data_as(C.c_void_p)
        locked = pin and lib.esq_host_pin(ptr, out.nbytes) == 0
        assert lib.esq_snapshot_copy(token, ptr, int(locked)) == 0
        np.testing.assert_array_equal(out, data)
        assert np.isnan(raw[:shift]).all() and np.isnan(raw[shift + n:]).all()
    after = _lib.copy_lane_info(0)
    assert after["engine_copies"] >= before["engine_copies"] + 4
    assert after["best_gbs"] > 1.0 and after["last_gbs"] > 1.0
    dev.close()


def test_device_memory_of_a_destroyed_context_serves_the_next_one(monkeypatch):
    """the slab of a closed context is kept and handed to the next context of that size
    (csrc/esq_core.hip: memory that hipMalloc hands out a second time is slow for the
    DMA engines); a context of another size gets memory of its own; the cache can be
    emptied; reused memory starts zeroed like fresh memory"""
    from extensisq_amd import _lib
    from extensisq_amd.device import DeviceContext
    lib = _lib.load()
    _lib.release_cached_memory()
    n = 3 * 1048576 + 5

    a = DeviceContext(n, 4)
    a.upload(_lib.SLOT_Y, 0, np.full(n, 7.0))
    a.close()
    held = _lib.release_cached_memory()
    assert held >= 8 * n * 4                              # the slab was in the cache ...
    assert _lib.release_cached_memory() == 0              # ... and is not any more
    b = DeviceContext(n, 4)
    b.upload(_lib.SLOT_Y, 0, np.full(n, 7.0))
    b.upload(_lib.SLOT_YNEW, 0, np.full(n, 9.0))
    b.close()
    c = DeviceContext(n, 4)                               # takes b's slab
    assert _lib.release_cached_memory() == 0              # (nothing left behind)
    np.testing.assert_array_equal(c.download(_lib.SLOT_Y, 0), np.zeros(n))
    np.testing.assert_array_equal(c.download(_lib.SLOT_YNEW, 0), np.zeros(n))
    d = DeviceContext(n + 2048, 4)                        # another size: its own memory
    c.upload(_lib.SLOT_Y, 0, np.full(n, 1.0))
    d.upload(_lib.SLOT_Y, 0, np.full(n + 2048, 2.0))
    np.testing.assert_array_equal(c.download(_lib.SLOT_Y, 0), np.full(n, 1.0))
    np.testing.assert_array_equal(d.download(_lib.SLOT_Y, 0), np.full(n + 2048, 2.0))
    c.close()
    d.close()
    assert _lib.release_cached_memory() >= 8 * (2 * n + 2048) * 4
    # the cap: nothing is kept with ESQ_SLAB_CACHE_MB=0; the oldest block goes first
    monkeypatch.setenv("ESQ_SLAB_CACHE_MB", "0")
    e = DeviceContext(n, 4)
    e.close()
    assert _lib.release_cached_memory() == 0
    slab_mb = 8 * n * (4 + 6) / 2 ** 20                   # (rows + fixed slots, roughly)
    monkeypatch.setenv("ESQ_SLAB_CACHE_MB", str(int(1.5 * slab_mb)))
    f, g = DeviceContext(n, 4), DeviceContext(n + 4096, 4)
    f.close()
    g.close()                                             # f's slab has to make room
    held = _lib.release_cached_memory()
    assert 8 * (n + 4096) * 4 <= held < 8 * (2 * n) * 4 + (64 << 20)


# ------------------------------------- deferred mirrors of large states (lazy.py)
def _lazy_pair(monkeypatch, cls, N=1024, **kw):
    """two solvers of the same IVP (heat, n = N^2 >= 8 MB / 8): `solver.y` deferred /
    downloaded at once"""
    rho = esq.Heat2D(N).spectral_radius()
    kw = dict(dict(rtol=1e-6, atol=1e-9, first_step=0.5 / rho, max_step=1.0 / rho,
                   nfev_stiff_detect=0), **kw)
    y0 = pb.heat2d_y0(N, seed=3)
    # (by default only scipy's solve_ivp loop gets the mirror: `always` = every reader)
    monkeypatch.setenv("ESQ_LAZY_Y", "always")
    a = cls(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_LAZY_Y", "0")
    b = cls(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_LAZY_Y")
    assert a._lazy_on and a._lazy_always and not b._lazy_on
    # a solver made without the switch hands a direct reader the plain array
    c = cls(esq.Heat2D(N), 0.0, y0, 1.0, **kw)
    assert c._lazy_on and isinstance(c.y, np.ndarray)
    return a, b


@pytest.mark.parametrize("name", ["Pr8", "Ts5", "BS5"])
def test_lazy_state_mirror_matches_the_immediate_download(monkeypatch, name):
    """`solver.y` of a large device-resident state is a LazyState: nothing is copied
    until it is used; used one step later, or kept by the caller over many steps
    (what plain solve_ivp does, ivp.py:702), it holds the state of ITS step, bit for
    bit what a solver that downloads at once returns -- distinct arrays, as the
    reference's (common.py:343)"""
    from extensisq_amd.lazy import LazyState
    a, b = _lazy_pair(monkeypatch, DEV[name])
    kept, want = [], []
    for k in range(7):
        assert a.step() is None and b.step() is None
        ya, yb = a.y, b.y
        assert isinstance(ya, LazyState) and not ya.materialized
        assert isinstance(yb, np.ndarray)
        assert a.y is ya                              # one mirror per state
        kept.append(ya)
        want.append(yb)
        if k == 0:
            assert not a._lazy_eager
        if k >= 2:
            # the caller stores its states: from the third step on the copies run
            # beside the steps (esq_snapshot_*), started when the mirror is made
            assert a._lazy_eager and ya._pending is not None
    for k, (ya, yb) in enumerate(zip(kept, want)):
        np.testing.assert_array_equal(np.asarray(ya), yb, err_msg=f"state {k}")
    arrays = [np.asarray(m) for m in kept]
    assert len({x.ctypes.data for x in arrays}) == len(arrays)
    assert a.t == b.t and a.nfev == b.nfev
    # used late: the state before the current one is still on the device
    assert a.step() is None and b.step() is None
    ya, yb = a.y, b.y
    del kept, arrays
    assert a.step() is None and b.step() is None
    np.testing.assert_array_equal(ya + 0.0, yb)
    # not used at all: nothing is copied, and the eager copies stop again
    for _ in range(4):
        assert a.step() is None and b.step() is None
        a.y, b.y
    assert not a._lazy_eager
    np.testing.assert_array_equal(np.asarray(a.y), b.y)


def test_esq_options_reach_the_library_through_solve_ivp():
    """`solve_ivp(..., method=Pr8, esq_options={...})`: the ESQ_* switches as a keyword
    of the drop-in call (scipy hands unknown options to the solver's constructor,
    ivp.py:621) -- here: one sweep per stage instead of chain sweeps; same results"""
    import os
    N = 96
    rho = esq.Brusselator2D(N).spectral_radius()
    y0 = pb.bruss2d_y0(N)
    kw = dict(rtol=1e-6, atol=1e-9, first_step=0.25 / rho, max_step=0.25 / rho,
              nfev_stiff_detect=0)
    seen = []

    class Spy(esq.Pr8):
        def _step_impl(self):
            out = super()._step_impl()
            seen.append(self)
            return out

    a = solve_ivp(esq.Brusselator2D(N), (0.0, 3.0 / rho), y0, method=Spy,
                  esq_options={"chain_rows": 12}, **kw)
    chained = seen[-1]
    b = solve_ivp(esq.Brusselator2D(N), (0.0, 3.0 / rho), y0, method=Spy,
                  esq_options={"chain_rows": 12, "chain_depth": 1}, **kw)
    single = seen[-1]
    assert "ESQ_CHAIN_ROWS" not in os.environ and "ESQ_CHAIN_DEPTH" not in os.environ
    np.testing.assert_array_equal(a.y, b.y)
    for s, want in ((chained, True), (single, False)):
        s._dev.profile_reset()
        s._dev.profile_enable([0, 1, 2])
        s.status = "running"
        s.t_bound = s.t + 1.0
        assert s.step() is None
        labels = [row[0] for row in s._dev.profile_kernels()]
        assert any(lab.startswith("chain") for lab in labels) == want, labels


def test_lazy_state_survives_an_assignment_to_the_state(monkeypatch):
    """`solver.y = value` uploads a new state: a mirror of the old one that somebody
    holds has been downloaded before"""
    a, b = _lazy_pair(monkeypatch, esq.Pr8)
    assert a.step() is None and b.step() is None
    ya, yb = a.y, b.y
    a.y = 2.0 * np.asarray(yb)
    b.y = 2.0 * yb
    assert ya.materialized
    np.testing.assert_array_equal(np.asarray(ya), yb)
    assert a.step() is None and b.step() is None
    np.testing.assert_array_equal(np.asarray(a.y), b.y)


@pytest.mark.parametrize("mode", ["stored", "t_eval", "dense", "events"])
def test_solve_ivp_with_deferred_states(monkeypatch, mode):
    """the drop-in call on a large state, every way scipy uses `solver.y`
    (ivp.py:665-736): kept per step, ignored (t_eval / dense_output), handed to an
    event function -- results bit-identical to immediate downloads"""
    N = 1024
    rho = esq.Heat2D(N).spectral_radius()
    y0 = pb.heat2d_y0(N, seed=5)
    tf = 9.3 / rho
    kw = dict(rtol=1e-6, atol=1e-9, first_step=0.5 / rho, max_step=1.0 / rho,
              nfev_stiff_detect=0)
    seen = []
    if mode == "t_eval":
        kw["t_eval"] = [0.3 * tf, tf]
    elif mode == "dense":
        kw["dense_output"] = True
    elif mode == "events":
        mid = (N // 2) * N + N // 2

        def event(t, y):
            seen.append(type(y).__name__)
            return y[mid] - 0.5 * (y0[mid] + y0[mid] * np.exp(-2 * np.pi ** 2 * tf))
        kw["events"] = event
    got = solve_ivp(esq.Heat2D(N), (0.0, tf), y0, method=esq.Pr8, **kw)
    if mode == "events":
        # user callbacks get what the reference gives them: plain ndarrays
        assert seen and set(seen) == {"ndarray"}, set(seen)
    monkeypatch.setenv("ESQ_LAZY_Y", "0")
    ref = solve_ivp(esq.Heat2D(N), (0.0, tf), y0, method=esq.Pr8, **kw)
    assert got.success and ref.success and got.nfev == ref.nfev
    np.testing.assert_array_equal(got.t, ref.t)
    np.testing.assert_array_equal(got.y, ref.y)
    assert isinstance(got.y, np.ndarray)
    if mode == "dense":
        tc = np.linspace(0.0, tf, 5)
        np.testing.assert_array_equal(got.sol(tc), ref.sol(tc))
    if mode == "events":
        np.testing.assert_array_equal(got.t_events[0], ref.t_events[0])
        np.testing.assert_array_equal(got.y_events[0], ref.y_events[0])


# ---------------------------------------------------- user-compiled RHS plugin
PLUGIN_SRC = r'''
#include <hip/hip_runtime.h>
// f_i = -k * y_i + cos(t)  -- a user plugin following include/extensisq_amd.h
__global__ void k_user(const double* y, double* f, size_t n, double k, double c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = -k * y[i] + c;
}
extern "C" int user_rhs(void* user, double t, const double* y, double* f, size_t n,
                        void* stream) {
    const double k = *(const double*)user;
    hipLaunchKernelGGL(k_user, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, y, f, n, k, cos(t));
    return (int)hipGetLastError();
}
'''


def test_user_plugin_compiled_with_hipcc(tmp_path):
    """a user's own `esq_rhs_fn`, built with hipcc and passed
    as a C function pointer through `CFunctionRHS`"""
    import ctypes
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "user_rhs.hip"
    so = tmp_path / "libuser_rhs.so"
    src.write_text(PLUGIN_SRC)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-fPIC", "-shared",
                    "-ffp-contract=off", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    kval = ctypes.c_double(0.75)
    n = 3001
    rhs = esq.CFunctionRHS(ctypes.cast(lib.user_rhs, ctypes.c_void_p).value,
                           ctypes.addressof(kval), n)
    y0 = np.linspace(-1.0, 2.0, n)
    cpu = lambda t, y: -0.75 * y + np.cos(t)  # noqa: E731
    assert_allclose(rhs(0.3, y0), cpu(0.3, y0), rtol=1e-15)
    got = solve_ivp(rhs, (0.0, 2.0), y0, method=esq.Pr7, rtol=1e-7, atol=1e-10)
    ref = solve_ivp(cpu, (0.0, 2.0), y0, method=rk_oracle.Pr7, rtol=1e-7,
                    atol=1e-10)
    assert got.nfev == ref.nfev
    # step sizes follow the (cancelling) error norms: ~1e-7 relative agreement
    assert_allclose(got.t, ref.t, rtol=1e-6)
    assert_allclose(got.y[:, -1], ref.y[:, -1], rtol=1e-7, atol=1e-11)


FUSED_PLUGIN_SRC = r'''
// A user plugin WITH a fused entry, written against the public helper headers
// (extensisq_amd/csrc/esq_plugin.hpp, esq_epilogue.hpp): f_i = -k*y_i + cos(t).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "esq_plugin.hpp"

__global__ void k_plain(const double* y, double* f, size_t n, double k, double c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = -k * y[i] + c;
}
template <class Epi>
__global__ __launch_bounds__(256) void k_sweep(const double* y, double* f, Epi epi,
                                               size_t n, size_t n2, double k, double c) {
    double local = 0.0;
    for (size_t i2 = (size_t)blockIdx.x * 256 + threadIdx.x; i2 < n2;
         i2 += (size_t)gridDim.x * 256) {
        typename Epi::In in;
        epi.load(in, i2);
        const double2 yc = esq::ld2(y, i2);
        double2 fy = make_double2(0.0, 0.0);          // the padding stays zero
        if (2 * i2 < n) fy.x = -k * yc.x + c;
        if (2 * i2 + 1 < n) fy.y = -k * yc.y + c;
        epi.store_f(f, i2, fy);
        epi.finish(in, fy, yc, i2, local);
    }
    if (Epi::kReduce) esq::block_partial(local, epi.red.partials);
}
extern "C" int user_rhs(void* user, double t, const double* y, double* f, size_t n,
                        void* stream) {
    const double k = *(const double*)user;
    hipLaunchKernelGGL(k_plain, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, y, f, n, k, cos(t));
    return (int)hipGetLastError();
}
extern "C" int user_rhs_fused(void* user, double t, const double* y, double* f,
                              const esq_epilogue* epi, size_t n, void* stream,
                              void* start_event, void* stop_event) {
    const double k = *(const double*)user;
    const size_t n2 = ((n + 511) / 512) * 512 / 2;    // vectors are padded to 512
    unsigned grid = (unsigned)((n2 + 255) / 256);
    if (grid > 1024) grid = 1024;
    if (esq::epilogue_reduces(epi)) {
        if ((int)grid > epi->partials_cap) return ESQ_ENOTSUP;
        *epi->partials_used = (int)grid;
    }
    const int rc = esq::dispatch_epilogue(epi, [&](auto ep) {
        hipExtLaunchKernelGGL((k_sweep<decltype(ep)>), dim3(grid), dim3(256), 0,
                              (hipStream_t)stream, (hipEvent_t)start_event,
                              (hipEvent_t)stop_event, 0, y, f, ep, n, n2, k, cos(t));
    });
    return rc ? rc : (int)hipGetLastError();
}
'''


@pytest.mark.parametrize("name", ["Pr8", "Ts5", "BS5"])
def test_user_plugin_with_fused_entry(tmp_path_factory, name):
    """a user's `esq_rhs_fused_fn` written with the public
    helper headers (`esq_plugin.hpp`: `dispatch_epilogue`, `epilogue_reduces`;
    `esq_epilogue.hpp`: the device-side epilogues).  The fused run must equal
    the run through the plain entry bit for bit, and the oracle to tolerance."""
    import ctypes
    import shutil
    import subprocess
    from extensisq_amd import _lib
    so = _user_fused_plugin(tmp_path_factory, ctypes, shutil, subprocess)
    lib = ctypes.CDLL(str(so))
    kval = ctypes.c_double(0.75)
    n = 3001
    plain_ptr = ctypes.cast(lib.user_rhs, ctypes.c_void_p).value
    fused_ptr = ctypes.cast(lib.user_rhs_fused, ctypes.c_void_p)

    class Fused(esq.CFunctionRHS):
        _fuse_default = True

        def _fused_entry(self, lib_):
            return fused_ptr

    y0 = np.linspace(-1.0, 2.0, n)
    kw = dict(first_step=0.05, rtol=1e-7, atol=1e-10)
    cls = DEV[name]
    a = cls(Fused(plain_ptr, ctypes.addressof(kval), n), 0.0, y0, 2.0, **kw)
    b = cls(esq.CFunctionRHS(plain_ptr, ctypes.addressof(kval), n), 0.0, y0, 2.0,
            **kw)
    o = rk_oracle.METHODS[name](lambda t, y: -0.75 * y + np.cos(t), 0.0, y0, 2.0,
                                **kw)
    a._dev.profile_enable([_lib.PROF_STAGE, _lib.PROF_SOLERR, _lib.PROF_RHS])
    for _ in range(5):
        assert a.step() is None and b.step() is None and o.step() is None
        assert_allclose(a.t, b.t, rtol=1e-13)
        # vs the oracle: the step sizes follow error norms that are cancelling
        # sums (rounding-level differences show up as ~1e-5 relative in h)
        assert_allclose(a.t, o.t, rtol=1e-4)
    a._dev.profile_enable(None)
    kernels = {row[0].split("<")[0] for row in a._dev.profile_kernels()}
    assert "rhs+stage" in kernels                      # the fused entry really ran
    assert ("rhs+solerr" in kernels) or ("rhs+errnorm" in kernels)
    assert a.nfev == b.nfev == o.nfev
    # the first step starts from identical data: bit-identical K and y; later
    # steps inherit the last-bit difference of the error norm through h
    assert_allclose(a.y, b.y, rtol=1e-11, atol=1e-14)
    assert_allclose(a.y, o.y, rtol=1e-4, atol=1e-6)


_FUSED_SO = {}


def _user_fused_plugin(tmp_path_factory, ctypes, shutil, subprocess):
    if "so" not in _FUSED_SO:
        d = tmp_path_factory.mktemp("fused_plugin")
        src = d / "user_fused.hip"
        so = d / "libuser_fused.so"
        src.write_text(FUSED_PLUGIN_SRC)
        root = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC",
                        "-shared", "-ffp-contract=off",
                        "-I", os.path.join(root, "extensisq_amd", "csrc"),
                        str(src), "-o", str(so)], check=True)
        _FUSED_SO["so"] = so
    return _FUSED_SO["so"]


# --------------------------------- a user's stencil plugin with all three entries
# INTEGRATION.md §4 prints this source: a third 2-D stencil -- the Gray-Scott
# reaction-diffusion system, two fields, periodic -- whose author writes the
# pointwise functor and six lines per entry point; the sweeps, the epilogues, the
# marching chain sweeps and their launch geometry come from esq_stencil2d.hpp.
GRAY_SCOTT_PLUGIN_SRC = r'''
#include "esq_stencil2d.hpp"            // -I <repo>/extensisq_amd/csrc

// u_t = Du lap(u) - u v^2 + F (1 - u),   v_t = Dv lap(v) + u v^2 - (F + k) v
// on an N x N periodic grid, state = [u.ravel(), v.ravel()]
struct GrayScott {
    double du, dv, F, k;                // Du / dx^2, Dv / dx^2, feed, kill
    __device__ __forceinline__ void eval(const double2 (&c)[2], const double2 (&lap)[2],
                                         double2 (&f)[2]) const {
        f[0] = eval_one(0, c, lap[0]);
        f[1] = eval_one(1, c, lap[1]);
    }
    __device__ __forceinline__ double2 eval_one(int field, const double2 (&c)[2],
                                                double2 lap) const {
        const double uvvx = c[0].x * c[1].x * c[1].x, uvvy = c[0].y * c[1].y * c[1].y;
        if (field == 0)
            return make_double2((du * lap.x - uvvx) + F * (1.0 - c[0].x),
                                (du * lap.y - uvvy) + F * (1.0 - c[0].y));
        return make_double2((dv * lap.x + uvvx) - (F + k) * c[1].x,
                            (dv * lap.y + uvvy) - (F + k) * c[1].y);
    }
};
struct User { int N; GrayScott fn; };
using P = esq::Stencil2D<2, /*PERIODIC=*/true, GrayScott>;

extern "C" int gs_rhs(void *user, double, const double *y, double *f, size_t n, void *stream) {
    const User *u = (const User *)user;
    if (n != 2 * (size_t)u->N * u->N) return ESQ_EINVAL;
    return P::rhs(u->fn, u->N, y, f, stream);
}
extern "C" int gs_fused(void *user, double, const double *y, double *f, const esq_epilogue *epi,
                        size_t, void *stream, void *e0, void *e1) {
    const User *u = (const User *)user;
    return P::fused(u->fn, u->N, y, f, epi, stream, e0, e1);
}
extern "C" int gs_chain(void *user, const double *y, const esq_chain *chain, size_t,
                        void *stream, void *e0, void *e1) {
    const User *u = (const User *)user;
    return P::chain(u->fn, u->N, y, chain, stream, e0, e1);
}
'''

_GS_SO = {}


def _gray_scott_plugin(tmp_path_factory):
    import shutil
    import subprocess
    if "so" not in _GS_SO:
        d = tmp_path_factory.mktemp("gs_plugin")
        src, so = d / "gray_scott.hip", d / "libgray_scott.so"
        src.write_text(GRAY_SCOTT_PLUGIN_SRC)
        root = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-shared", "-ffp-contract=off",
                        "-I", os.path.join(root, "extensisq_amd", "csrc"),
                        str(src), "-o", str(so)], check=True)
        _GS_SO["so"] = so
    return _GS_SO["so"]


def _gray_scott_numpy(N, du, dv, F, k):
    """NumPy twin of the functor, same operation order"""
    def fun(t, y):
        u, v = y[:N * N].reshape(N, N), y[N * N:].reshape(N, N)

        def lap(a):
            return ((np.roll(a, 1, 0) + np.roll(a, -1, 0))
                    + (np.roll(a, 1, 1) + np.roll(a, -1, 1))) - 4.0 * a
        uvv = u * v * v
        fu = (du * lap(u) - uvv) + F * (1.0 - u)
        fv = (dv * lap(v) + uvv) - (F + k) * v
        return np.concatenate([fu.ravel(), fv.ravel()])
    return fun


@pytest.mark.parametrize("name", ["Pr8", "Ts5"])
def test_user_compiled_chain_plugin(monkeypatch, tmp_path_factory, name):
    """a THIRD stencil, compiled by the user with hipcc against esq_stencil2d.hpp
    only, with all three entries registered through CFunctionRHS: its RHS agrees
    bit for bit with the NumPy twin of the functor, its chained steps (marching
    sweeps of up to 4-5 stages, one field per wave) are bit-identical to its own
    one-sweep-per-stage steps (ESQ_CHAIN_DEPTH=1) and to the entry-free run, and
    within the single-step bounds of the oracle on the twin"""
    import ctypes
    from extensisq_amd import _lib
    lib = ctypes.CDLL(str(_gray_scott_plugin(tmp_path_factory)))

    class GS(ctypes.Structure):
        _fields_ = [("du", ctypes.c_double), ("dv", ctypes.c_double),
                    ("F", ctypes.c_double), ("k", ctypes.c_double)]

    class User(ctypes.Structure):
        _fields_ = [("N", ctypes.c_int), ("fn", GS)]

    N = 96
    du, dv, F, k = 0.16 * 4.0, 0.08 * 4.0, 0.035, 0.06
    user = User(N, GS(du, dv, F, k))
    ptr = lambda f: ctypes.cast(f, ctypes.c_void_p)      # noqa: E731
    rhs_ptr = ptr(lib.gs_rhs).value

    class Plugin(esq.CFunctionRHS):
        _fuse_default = True
        _chain_caps = 15

        def _fused_entry(self, lib_):
            return ptr(lib.gs_fused)

        def _chain_entry(self, lib_):
            return ptr(lib.gs_chain)

    rng = np.random.default_rng(5)
    u0 = 1.0 - 0.5 * rng.random((N, N))
    v0 = 0.25 * rng.random((N, N))
    y0 = np.concatenate([u0.ravel(), v0.ravel()])
    twin = _gray_scott_numpy(N, du, dv, F, k)
    plain = esq.CFunctionRHS(rhs_ptr, ctypes.addressof(user), y0.size)
    assert_equal(plain(0.0, y0), twin(0.0, y0))          # the RHS: bit for bit
    rho = 8.0 * du + 1.0
    h = 0.5 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    cls = DEV[name]
    monkeypatch.setenv("ESQ_CHAIN_ROWS", "12")           # chains on this small grid
    chained = cls(Plugin(rhs_ptr, ctypes.addressof(user), y0.size), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_CHAIN_DEPTH", "1")
    single = cls(Plugin(rhs_ptr, ctypes.addressof(user), y0.size), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN_DEPTH")
    bare = cls(plain, 0.0, y0, 1.0, **kw)
    o = rk_oracle.METHODS[name](twin, 0.0, y0, 1.0, **kw)
    y_old = o.y
    for s in (chained, single, bare, o):
        assert s.step() is None
    check_step(chained, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-6, 1e-9, k_rtol=2e-13, lipschitz=rho)
    chained._dev.profile_reset()
    chained._dev.profile_enable([_lib.PROF_STAGE, _lib.PROF_SOLERR, _lib.PROF_RHS])
    for _ in range(3):
        for s in (chained, single, bare):
            assert s.step() is None
        assert chained.t == single.t == bare.t
    chained._dev.profile_enable(None)
    labels = {row[0].split("<")[0] for row in chained._dev.profile_kernels()}
    assert any(lab.startswith("chain") for lab in labels), labels
    assert "rhs_plugin" not in labels, labels
    assert_equal(chained.y, single.y)
    assert_equal(chained.y, bare.y)
    assert_equal(chained.K, single.K)
    assert_equal(chained.K, bare.K)
    assert chained.nfev == single.nfev == bare.nfev


# -------------------------------------------------- CKdisc (variable order)
@pytest.mark.parametrize("case", ["readme", "duffing", "rational_bwd", "complex",
                                  "sawtooth", "kink", "bruss1d"])
def test_ckdisc_golden(golden_dir, case):
    """variable-order Cash-Karp on the device (reference cash.py:253-395): every
    order assessment is one fused `esq_rk_custom_sol_err` pass; trajectories,
    accepted orders and dense output against the reference's run"""
    from test_oracle_golden import _ckdisc_check
    with open(os.path.join(golden_dir, "ckdisc_traces.json")) as fh:
        gold = json.load(fh)
    _ckdisc_check(esq.CKdisc, case, gold, esq.NFS, 1e-7)


def test_ckdisc_device_rhs_matches_oracle():
    n = 5000
    rng = np.random.default_rng(8)
    lam = -rng.random(n) * 3
    y0 = rng.standard_normal(n)
    kw = dict(rtol=1e-5, atol=1e-8)
    d = esq.CKdisc(esq.DiagonalLinear(lam, 1.0), 0.0, y0, 2.0, **kw)
    o = rk_oracle.CKdisc(lambda t, y: lam * y + np.sin(t), 0.0, y0, 2.0, **kw)
    assert_allclose(d.h_abs, o.h_abs, rtol=1e-11)
    for _ in range(6):
        assert d.step() is None and o.step() is None
        assert d.order_accepted == o.order_accepted
        assert_allclose(d.t, o.t, rtol=1e-7)
        assert_allclose(d.y, o.y, rtol=1e-7, atol=1e-10)
    assert d.nfev == o.nfev
    tc = np.linspace(o.t_old, o.t, 4)
    assert_allclose(d.dense_output()(tc), o.dense_output()(tc), rtol=1e-7,
                    atol=1e-10)


# --------------------------------- the other BASELINE.json configs, full size
def test_full_size_ts5_heat_step_matches_oracle():
    """configs[1]: Ts5, 2-D heat N = 1000 (n = 1e6), two steps vs the oracle"""
    N = 1000
    rhs = esq.Heat2D(N)
    y0 = pb.heat2d_y0(N)
    h = 1.0 / rhs.spectral_radius()
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    d, o = _pair("Ts5", rhs, pb.heat2d_rhs(N), 0.0, y0, 1.0, **kw)
    y_old = o.y
    assert d.step() is None and o.step() is None
    check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-3, 1e-6, k_rtol=2e-13, lipschitz=rhs.spectral_radius())
    assert int(esq.NFS[()]) == 0
    assert d.step() is None and o.step() is None
    assert d.t == o.t and d.nfev == o.nfev
    assert_allclose(d.y, o.y, rtol=1e-12, atol=1e-14)


def test_full_size_pr9_heat_step_matches_oracle():
    """configs[4] shard: Pr9, 2-D heat N = 2236 (n = 4 999 696), one step"""
    N = 2236
    rhs = esq.Heat2D(N)
    y0 = pb.heat2d_y0(N)
    h = 1.0 / rhs.spectral_radius()
    kw = dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    d, o = _pair("Pr9", rhs, pb.heat2d_rhs(N), 0.0, y0, 1.0, **kw)
    y_old = o.y
    assert d.step() is None and o.step() is None
    check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y_old, o.h_previous,
               1e-3, 1e-6, k_rtol=2e-13, lipschitz=rhs.spectral_radius())
    assert int(esq.NFS[()]) == 0
    assert d.nfev == o.nfev == 18


# ------------------------------------------------------ blocked accumulation
@pytest.mark.parametrize("name", ERK + ["CKdisc"])
@pytest.mark.parametrize("mode", ["device_rhs", "host_rhs"])
def test_blocked_accumulation_is_bit_identical(monkeypatch, name, mode):
    """ESQ_BLOCK_ACC (default on): leading columns of A are accumulated once
    for all later stages and each stage resumes the same FMA chain -- K rows,
    states, error norms must equal the one-kernel-per-stage path bit for bit"""
    n = 3001
    rng = np.random.default_rng(21)
    lam = -rng.random(n) * 2.0
    y0 = rng.standard_normal(n)
    cls = getattr(esq, name)

    def make():
        fun = (esq.DiagonalLinear(lam, 1.0) if mode == "device_rhs"
               else (lambda t, y: lam * y + np.sin(t)))
        return cls(fun, 0.2, y0, 5.0, first_step=0.05, rtol=1e-6, atol=1e-9)
    blocked = make()
    monkeypatch.setenv("ESQ_BLOCK_ACC", "0")
    plain = make()
    monkeypatch.delenv("ESQ_BLOCK_ACC")
    for _ in range(4):
        assert blocked.step() is None and plain.step() is None
        assert blocked.t == plain.t and blocked.h_abs == plain.h_abs
        assert_equal(blocked.K, plain.K)
        assert_equal(blocked.y, plain.y)
    assert blocked.nfev == plain.nfev
    assert int(esq.NFS[()]) >= 0


def test_blocked_accumulation_plans(monkeypatch):
    """the column boundaries the library derives from each tableau's sparsity
    and the 8-byte words per element and step the stage kernels then move"""
    import ctypes
    monkeypatch.delenv("ESQ_BLOCK_ACC", raising=False)
    expect = {"Ts5": ([], 25, 25), "BS5": ([4], 33, 31), "Pr7": ([6], 58, 51),
              "Pr8": ([7], 93, 75), "Pr9": ([8, 13], 154, 109),
              "CK5": ([], 25, 25), "Me4": ([], 16, 16), "CFMR7osc": ([5], 46, 42)}
    for name, (bounds, plain, blocked) in expect.items():
        s = getattr(esq, name)(lambda t, y: -y, 0.0, np.ones(7), 1.0,
                               first_step=0.1)
        b = (ctypes.c_int * 8)()
        wp, wb = ctypes.c_int(), ctypes.c_int()
        k = s._lib.esq_rk_block_plan(s._ctx, b, 8, ctypes.byref(wp),
                                     ctypes.byref(wb))
        assert (list(b[:k]), wp.value, wb.value) == (bounds, plain, blocked), name


# ------------------------------------------------------------- chained stages
def _plugin(plugin, N):
    if plugin == "bruss":
        return (lambda: esq.Brusselator2D(N)), pb.bruss2d_y0(N), pb.bruss2d_rho(N)
    return (lambda: esq.Heat2D(N)), pb.heat2d_y0(N), pb.heat2d_rho(N)


@pytest.mark.parametrize("name", ERK + ["CKdisc"])
@pytest.mark.parametrize("plugin,N", [("bruss", 4), ("bruss", 50), ("bruss", 258),
                                      ("heat", 6), ("heat", 130)])
def test_chained_stages_are_bit_identical(monkeypatch, name, plugin, N):
    """ESQ_CHAIN (default on): every RHS sweep also does the Runge-Kutta
    arithmetic that follows it (all epilogue kinds, first stage of the next
    step formed at accept time); K rows and states must equal the
    one-kernel-per-operation path bit for bit (also combined with blocked
    accumulation), the error norm to rounding (its partial sums are grouped by
    the sweep's workgroups instead of the grid-stride blocks)"""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    monkeypatch.setenv("ESQ_CHAIN", "1")
    chained = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_CHAIN", "0")
    plain = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN")
    chained._prelaunch, plain._prelaunch = True, False
    for _ in range(3):
        assert plain.step() is None and chained.step() is None
        assert chained.t == plain.t
        assert_allclose(chained.h_abs, plain.h_abs, rtol=1e-12)
        if name != "CKdisc":
            assert_allclose(chained.error_norm_old, plain.error_norm_old, rtol=1e-12)
        assert_equal(chained.K, plain.K)
        assert_equal(chained.y, plain.y)
    assert chained.nfev == plain.nfev
    sd, sp = chained.dense_output(), plain.dense_output()
    tc = np.linspace(plain.t_old, plain.t, 3)
    assert_equal(sd(tc), sp(tc))


@pytest.mark.parametrize("fuse", ["stage", "stage,block", "stage,solerr",
                                  "stage,errnorm", "block,solerr,errnorm",
                                  "stage,src", "stage,block,solerr,errnorm,src"])
@pytest.mark.parametrize("name,plugin,N", [
    ("Pr8", "bruss", 50), ("Pr8", "bruss", 258), ("Pr9", "heat", 130),
    ("Pr7", "heat", 36), ("Ts5", "heat", 258), ("Ts5", "bruss", 48),
    ("BS5", "heat", 130), ("CFMR7osc", "bruss", 36)])
def test_each_epilogue_kind_is_bit_identical(monkeypatch, fuse, name, plugin, N):
    """ESQ_FUSE selects the epilogue kinds one by one: (a) next stage argument,
    (b) blocked accumulation inside the boundary stage's sweep, (c) solution +
    error norm inside the last stage's sweep, FSAL error norm inside the
    end-point sweep, "src": the first sweep of a step forms its own input from
    y and K[0] -- each must leave K rows and states bit-identical"""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    monkeypatch.setenv("ESQ_FUSE", fuse)
    monkeypatch.setenv("ESQ_SRC", "1")          # "src" at any size
    fused = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_FUSE")
    monkeypatch.delenv("ESQ_SRC")
    monkeypatch.setenv("ESQ_CHAIN", "0")
    plain = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN")
    for _ in range(3):
        assert fused.step() is None and plain.step() is None
        assert fused.t == plain.t
        assert_allclose(fused.error_norm_old, plain.error_norm_old, rtol=1e-12)
        assert_equal(fused.K, plain.K)
        assert_equal(fused.y, plain.y)
    assert fused.nfev == plain.nfev


@pytest.mark.parametrize("name", ["Pr8", "Ts5", "BS5", "Pr9"])
def test_prelaunched_first_stage_is_used_only_when_valid(monkeypatch, name):
    """esq_rk_accept(h_next) forms the next step's first stage argument ahead of
    time.  It must be dropped when (1) the next step asks for another h (the
    user changed h_abs), (2) a vector was written in between (solver.y = ...),
    (3) the attempt is repeated after a rejection -- and used otherwise; every
    variant must equal the run without it bit for bit"""
    N = 48
    mk, y0, rho = _plugin("heat", N)
    cls = getattr(esq, name)
    kw = dict(first_step=0.5 / rho, rtol=1e-5, atol=1e-8)
    # without "src" (with it the first sweep needs no stage argument at all)
    monkeypatch.setenv("ESQ_FUSE", "stage,block,solerr,errnorm")

    def run(prelaunch):
        s = cls(mk(), 0.0, y0, 1.0, **kw)
        s._prelaunch = prelaunch
        out = []
        for k in range(8):
            if k == 2:
                s.h_abs = 0.7 * s.h_abs                # (1) another step size
            if k == 4:
                s.y = 0.5 * s.y                        # (2) the state is replaced
                s._chk(s._lib.esq_rk_eval_rhs(s._ctx, 0, float(s.t), 1, 0),
                       "esq_rk_eval_rhs")                # ... and K[0] = f(t, y)
            if k == 6:
                s.h_abs = 40.0 * s.h_abs               # (3) forces rejections
            assert s.step() is None
            out.append((s.t, s.h_abs, s.y.copy(), s.K.copy()))
        return out, int(esq.NFS[()])
    a, nfs_a = run(True)
    b, nfs_b = run(False)
    assert nfs_a == nfs_b and nfs_a > 0
    for (ta, ha, ya, Ka), (tb, hb, yb, Kb) in zip(a, b):
        assert ta == tb and ha == hb
        assert_equal(ya, yb)
        assert_equal(Ka, Kb)


@pytest.mark.parametrize("N", [2, 3, 5, 7])
def test_plugin_fallback_paths_small_and_odd_grids(N):
    """grids the vectorised / chained plugin entries decline (odd N, N < 4):
    the library must fall back to the plain kernels and still match the oracle"""
    y0 = pb.heat2d_y0(N)
    kw = dict(rtol=1e-5, atol=1e-8)
    for name in ("Ts5", "Pr7"):
        d, o = _pair(name, esq.Heat2D(N), pb.heat2d_rhs(N), 0.0, y0, 0.05, **kw)
        while o.status == "running":
            assert d.step() is None and o.step() is None
        assert d.status == "finished" and d.nfev == o.nfev
        assert_allclose(d.t, o.t, rtol=1e-12)
        assert_allclose(d.y, o.y, rtol=1e-8, atol=1e-12)
    from oracle import rkc_oracle
    d = esq.SSV2stab(esq.Heat2D(N), 0.0, y0, 0.05, **kw)
    o = rkc_oracle.SSV2stab(pb.heat2d_rhs(N), 0.0, y0, 0.05, **kw)
    while o.status == "running":
        assert d.step() is None and o.step() is None
    assert d.status == "finished"
    assert_allclose(d.y, o.y, rtol=1e-7, atol=1e-11)


# ------------------------------------------------------- small host-RHS problems
@pytest.mark.parametrize("name", ERK + ["CKdisc", "SSV2stab"])
def test_host_slab_mode_is_bit_identical(monkeypatch, name):
    """small problems with a Python RHS keep their vectors in pinned,
    device-mapped host memory (ESQ_CREATE_HOST_SLAB): same kernels, uploads and
    downloads become memcpy -- the run must equal the device-slab run bit for bit"""
    cls = getattr(esq, name)
    fun, t_span, y0, kw = CASES["bruss1d"]

    def run(flag):
        monkeypatch.setenv("ESQ_HOST_SLAB", flag)
        s = cls(fun, t_span[0], y0, t_span[1], **kw)
        assert s._dev.host_slab == (flag == "1")
        out = []
        for _ in range(6):
            assert s.step() is None
            out.append((s.t, s.y.copy()))
        dense = s.dense_output()(0.5 * (s.t_old + s.t))
        return out, dense, s.nfev
    a, da, na = run("1")
    b, db, nb = run("0")
    assert na == nb
    for (ta, ya), (tb, yb) in zip(a, b):
        assert ta == tb
        assert_equal(ya, yb)
    assert_equal(da, db)


def test_host_slab_mode_is_faster_for_small_host_rhs(monkeypatch, capsys):
    """n = 400, Python RHS (the regime of the reference's own tests): wall time
    per Pr8 step with the pinned host slab against the device slab (and the
    NumPy oracle for scale).  The host slab must not be slower."""
    import time
    n = 400
    rng = np.random.default_rng(0)
    lam = -rng.random(n)
    y0 = rng.standard_normal(n)
    fun = lambda t, y: lam * y          # noqa: E731
    kw = dict(first_step=1e-3, max_step=1e-3, rtol=1e-6, atol=1e-9,
              nfev_stiff_detect=0)

    def per_step(make, steps=200):
        s = make()
        for _ in range(20):
            s.step()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step()
        return (time.perf_counter() - t0) / steps
    monkeypatch.setenv("ESQ_HOST_SLAB", "1")
    t_host = per_step(lambda: esq.Pr8(fun, 0.0, y0, 1e9, **kw))
    monkeypatch.setenv("ESQ_HOST_SLAB", "0")
    t_dev = per_step(lambda: esq.Pr8(fun, 0.0, y0, 1e9, **kw))
    t_ora = per_step(lambda: rk_oracle.Pr8(fun, 0.0, y0, 1e9, **kw))
    with capsys.disabled():
        print(f"\n[small-n] Pr8 n=400 host RHS: host slab {1e6 * t_host:.0f} us/step, "
              f"device slab {1e6 * t_dev:.0f} us/step, NumPy oracle "
              f"{1e6 * t_ora:.0f} us/step")
    assert t_host < 1.1 * t_dev


def test_lockstep_host_reducer_estimates_the_first_step_on_the_whole_batch():
    """host-reducer lock-step with `first_step=None`: Watts' starting-step
    procedure (common.py:519-763) runs on norms, log-tolerance sums and minima of
    the WHOLE batch, so every shard starts with the step the oracle chooses on
    the concatenated state and the batch stays in lock-step from there."""
    import threading
    N, world = 16, 2
    n = N * N
    y0s = [(1.0 + 0.5 * r) * pb.heat2d_y0(N, seed=7 + r) for r in range(world)]
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
            grp.debug = True
            s = esq.Pr7(esq.Heat2D(N), 0.0, y0s[rank], 2e-3, rtol=1e-5, atol=1e-8,
                        nfev_stiff_detect=0, lockstep=grp)
            h0 = s.h_abs
            ts = []
            while s.status == "running":
                assert s.step() is None
                ts.append(s.t)
            results[rank] = (h0, ts, s.y, s.nfev)
        except BaseException as exc:       # noqa: BLE001
            errors.append(exc)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    assert results[0][0] == results[1][0] and results[0][1] == results[1][1]
    f1 = pb.heat2d_rhs(N)

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(world)])

    ref = rk_oracle.Pr7(fun, 0.0, np.concatenate(y0s), 2e-3, rtol=1e-5, atol=1e-8,
                        nfev_stiff_detect=0)
    assert_allclose(results[0][0], ref.h_abs, rtol=1e-9)
    ts = []
    while ref.status == "running":
        assert ref.step() is None
        ts.append(ref.t)
    assert len(ts) == len(results[0][1])
    assert_allclose(results[0][1], ts, rtol=1e-7)
    assert_allclose(np.concatenate([r[2] for r in results]), ref.y, rtol=1e-7,
                    atol=1e-11)
    assert results[0][3] == ref.nfev


# ------------------------------------------ complex states on the device path
# The reference's RungeKutta accepts complex128 states (common.py:187-190,
# norm = real(x @ conj(x)), :64-66; tests/test_rk.py:92-98,
# tests/test_ivp.py:216-259).  Here they run with a DEVICE right-hand side: the
# state never leaves HBM, the reducing epilogues use the complex modulus.
def _complex_problem(n, seed=5):
    rng = np.random.default_rng(seed)
    lam = -rng.random(n) * 2.0 + 1j * (rng.random(n) * 6.0 - 3.0)
    y0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    amp = 0.7 - 0.4j
    return lam, y0, amp


@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("n", [1, 37, 256, 4099])
def test_complex_device_rhs_step(name, n):
    """one step of every ERK class on a complex state with the complex
    DiagonalLinear plugin: K, y_new, the error norm and the next step size
    against the oracle from identical (t, y, f, h)"""
    lam, y0, amp = _complex_problem(n)
    kw = dict(first_step=0.05, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    rhs = esq.DiagonalLinear(lam, amp)
    assert rhs.is_complex
    d, o = _pair(name, rhs, lambda t, y: lam * y + amp * np.sin(t), 0.3, y0, 5.0,
                 **kw)
    assert d.y.dtype == np.complex128
    assert d.step() is None and o.step() is None
    assert d.t == o.t
    check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y0, o.h_previous, 1e-6,
               1e-9, k_rtol=4e-13)
    assert d.nfev == o.nfev
    assert np.isrealobj(d.error_norm_old)


@pytest.mark.parametrize("name", ERK)
def test_complex_fused_equals_unfused(name, monkeypatch):
    """the complex reducing epilogues (solution + error norm inside the last
    stage's sweep, FSAL error norm inside the end-point sweep) against the
    one-kernel-per-operation sequence: K rows and the state bit-identical, the
    error norm to rounding"""
    lam, y0, amp = _complex_problem(3001)
    kw = dict(first_step=0.04, rtol=1e-7, atol=1e-10, nfev_stiff_detect=0)
    out = {}
    for chain in ("0", "1"):
        monkeypatch.setenv("ESQ_CHAIN", chain)
        s = DEV[name](esq.DiagonalLinear(lam, amp), 0.1, y0, 3.0, **kw)
        for _ in range(3):
            assert s.step() is None
        out[chain] = (s.t, s.y.copy(), s.K.copy(), s.error_norm_old, s.nfev)
    a, b = out["0"], out["1"]
    assert a[0] == b[0] and a[4] == b[4]
    assert_equal(a[1], b[1])
    assert_equal(a[2], b[2])
    assert_allclose(a[3], b[3], rtol=1e-9)


@pytest.mark.parametrize("name", ERK)
def test_complex_device_rhs_trajectory_golden(traces, name):
    """the reference's complex decay trace (y' = -y, y0 = 0.5 + 1j, [0, 1];
    tests/test_ivp.py:216-259) with the state resident on the device"""
    res = solve_ivp(esq.DiagonalLinear(np.array([-1.0 + 0.0j])), [0, 1],
                    [0.5 + 1j], method=DEV[name], rtol=1e-3, atol=1e-6)
    compare_trajectory(res, int(esq.NFS[()]), traces[name]["complex"], 1e-3,
                       t_rtol=1e-7)
    assert np.iscomplexobj(res.y)


@pytest.mark.parametrize("name", ERK)
def test_error_estimation_complex_device_rhs(name):
    """tests/test_rk.py:92-98 with a device RHS: the error norm of a complex
    state is real"""
    h = 0.2
    s = DEV[name](esq.DiagonalLinear(np.array([1j])), 0, [1j], 1, first_step=h)
    assert s.step() is None
    err_norm = s._estimate_error_norm(s.K, h, scale=[1])
    assert np.isrealobj(err_norm)
    assert np.isrealobj(s.error_norm_old) and s.error_norm_old >= 0


CPLX_PLUGIN_SRC = r'''
#include <hip/hip_runtime.h>
// complex state, f_k = (a + i b) * y_k: n counts DOUBLES (two per element,
// interleaved re, im) -- include/extensisq_amd.h, esq_rhs_fn
__global__ void k_rot(const double2* y, double2* f, size_t n_cplx, double a, double b) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_cplx) {
        const double2 v = y[k];
        f[k] = make_double2(a * v.x - b * v.y, a * v.y + b * v.x);
    }
}
extern "C" int user_rhs_c(void* user, double t, const double* y, double* f, size_t n,
                          void* stream) {
    const double* ab = (const double*)user;
    const size_t nc = n / 2;
    hipLaunchKernelGGL(k_rot, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const double2*)y, (double2*)f, nc, ab[0],
                       ab[1]);
    return (int)hipGetLastError();
}
'''


def test_user_plugin_complex_state(tmp_path):
    """`CFunctionRHS(..., is_complex=True)`: a user's plain `esq_rhs_fn` on a
    complex state -- the stand-alone complex kernels (stage accumulate,
    solution + error norm with the complex modulus) against the oracle"""
    import ctypes
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "user_rhs_c.hip"
    so = tmp_path / "libuser_rhs_c.so"
    src.write_text(CPLX_PLUGIN_SRC)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-fPIC", "-shared",
                    "-ffp-contract=off", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    ab = (ctypes.c_double * 2)(-0.3, 2.0)
    n = 1501
    rhs = esq.CFunctionRHS(ctypes.cast(lib.user_rhs_c, ctypes.c_void_p).value,
                           ctypes.addressof(ab), n, is_complex=True)
    rng = np.random.default_rng(11)
    y0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    lam = -0.3 + 2.0j
    cpu = lambda t, y: lam * y  # noqa: E731
    assert_allclose(rhs(0.0, y0), cpu(0.0, y0), rtol=1e-15)
    for name in ("Ts5", "Pr8"):
        kw = dict(first_step=0.05, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
        d, o = _pair(name, rhs, cpu, 0.0, y0, 2.0, **kw)
        assert d.step() is None and o.step() is None
        check_step(d, o.K, o.y, o.error_norm_old, o.h_abs, y0, o.h_previous, 1e-6,
                   1e-9, k_rtol=4e-13)
    got = solve_ivp(rhs, (0.0, 1.0), y0, method=esq.Pr7, rtol=1e-7, atol=1e-10)
    ref = solve_ivp(cpu, (0.0, 1.0), y0, method=rk_oracle.Pr7, rtol=1e-7, atol=1e-10)
    assert got.nfev == ref.nfev
    assert_allclose(got.t, ref.t, rtol=1e-6)
    assert_allclose(got.y[:, -1], ref.y[:, -1], rtol=1e-7, atol=1e-11)


# --------------------------------------- several stages per marching sweep
@pytest.mark.parametrize("name,plugin,N", [
    ("Pr8", "bruss", 16), ("Pr8", "bruss", 50), ("Pr8", "bruss", 116),
    ("Pr8", "bruss", 120), ("Pr8", "bruss", 124), ("Pr8", "bruss", 258),
    ("Pr9", "heat", 130), ("Pr9", "heat", 250), ("Pr7", "heat", 36),
    ("Pr7", "bruss", 130), ("Ts5", "heat", 258), ("Ts5", "bruss", 48),
    ("BS5", "heat", 130), ("CK5", "bruss", 64), ("Me4", "heat", 100),
    ("CFMR7osc", "bruss", 36)])
@pytest.mark.parametrize("depth,rows", [(2, 5), (2, 32), (3, 7), (3, 64), (4, 9),
                                        (4, 30), (4, 200), (5, 11), (6, 13), (6, 40)])
def test_chained_stage_sweeps_are_bit_identical(monkeypatch, name, plugin, N, depth,
                                                rows):
    """ESQ_CHAIN_DEPTH: up to `depth` consecutive stages in ONE marching sweep
    (stage k runs k grid rows behind stage 0, the arguments in between never in
    memory; csrc/esq_chain.hpp), the solution/error sweep included -- K rows and
    states must equal the one-sweep-per-stage path bit for bit, at tile heights
    that put the halo rows everywhere (ESQ_CHAIN_ROWS), on periodic and
    Dirichlet grids whose width is and is not a multiple of the tile width"""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    # a forced tile height also lifts the "grid too small for this depth" rule
    monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    monkeypatch.setenv("ESQ_CHAIN_DEPTH", str(depth))
    chained = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_CHAIN_DEPTH", "1")
    single = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN_DEPTH")
    chained._dev.profile_enable([0, 1, 2])
    for _ in range(3):
        assert chained.step() is None and single.step() is None
        assert chained.t == single.t
        assert_allclose(chained.error_norm_old, single.error_norm_old, rtol=1e-12)
        assert_equal(chained.K, single.K)
        assert_equal(chained.y, single.y)
    assert chained.nfev == single.nfev
    labels = [row[0] for row in chained._dev.profile_kernels()]
    if name in ("Pr7", "Pr8", "Pr9", "Ts5"):     # tableaux with chainable stages
        assert any(lab.startswith("chain") for lab in labels), labels
    if name == "Pr8" and depth >= 3:
        assert any(lab.startswith(("chain3", "chain4", "chain5", "chain6"))
                   for lab in labels), labels


# ------------------------- rows of K only their own sweep reads (lazy rows)
def _lazy_state(solver):
    import ctypes
    missing, keeps, restores = ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
    solver._chk(solver._lib.esq_rk_lazy_rows(solver._ctx, ctypes.byref(missing),
                                             ctypes.byref(keeps),
                                             ctypes.byref(restores), None, None),
                "esq_rk_lazy_rows")
    return missing.value, bool(keeps.value), restores.value


def _end_point_counts(solver):
    import ctypes
    fused, plain = ctypes.c_long(), ctypes.c_long()
    solver._chk(solver._lib.esq_rk_lazy_rows(solver._ctx, None, None, None,
                                             ctypes.byref(fused), ctypes.byref(plain)),
                "esq_rk_lazy_rows")
    return fused.value, plain.value


@pytest.mark.parametrize("name,plugin,N,rows", [
    ("Pr8", "bruss", 50, 9), ("Pr8", "bruss", 124, 30), ("Pr8", "heat", 36, 8),
    ("Pr8", "heat", 130, 30), ("Pr8", "bruss", 258, 40)])
def test_rows_only_their_sweep_reads_are_restored_on_demand(monkeypatch, name, plugin,
                                                            N, rows):
    """The stages of a step's last chain sweep (non-FSAL pairs) are read by nothing
    but that sweep's own solution / error sums; esq_rk_stages does not write them
    (label `chain<D>+solerr-K`).  Whoever reads rows of K (`solver.K`, the dense
    output, the error vector) gets them re-evaluated first, bit-identical to a
    context that always writes them (ESQ_LAZY_ROWS=0); a second reader within four
    steps makes the context keep its rows (ref: `self.K[s] = ...`,
    common.py:353-356, where every row is always in memory)."""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    lazy = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_LAZY_ROWS", "0")
    monkeypatch.setenv("ESQ_LAZY_END", "0")
    eager = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_LAZY_ROWS")
    monkeypatch.delenv("ESQ_LAZY_END")
    assert _lazy_state(eager)[1] and not _lazy_state(lazy)[1]
    lazy._dev.profile_enable([0, 1, 2])
    # steps nobody looks into: rows missing, states identical
    for _ in range(3):
        assert lazy.step() is None and eager.step() is None
        assert lazy.t == eager.t
        # (the partial sums are grouped by whichever sweep forms them)
        assert_allclose(lazy.error_norm_old, eager.error_norm_old, rtol=1e-12)
    missing, keeps, restores = _lazy_state(lazy)
    assert missing >= 2 and not keeps and restores == 0
    assert _lazy_state(eager) == (0, True, 0)
    labels = [row[0] for row in lazy._dev.profile_kernels()]
    assert any("+solerr-K<" in lab for lab in labels), labels
    assert_equal(lazy.y, eager.y)                    # reading y restores nothing
    assert _lazy_state(lazy)[0] == missing
    # first reader: the dense output of the step just accepted
    sl, se = lazy.dense_output(), eager.dense_output()
    tc = np.linspace(eager.t_old, eager.t, 4)
    assert_equal(sl(tc), se(tc))
    assert _lazy_state(lazy) == (0, False, 1)
    assert_equal(lazy.K, eager.K)
    # five quiet steps, then one reader: still lazy
    for _ in range(5):
        assert lazy.step() is None and eager.step() is None
    assert_equal(lazy.K, eager.K)
    assert _lazy_state(lazy) == (0, False, 2)
    # a reader on the very next step too: the context keeps its rows from now on
    assert lazy.step() is None and eager.step() is None
    assert _lazy_state(lazy)[0] >= 2
    assert_equal(lazy.K, eager.K)
    assert _lazy_state(lazy) == (0, True, 3)
    for _ in range(2):
        assert lazy.step() is None and eager.step() is None
        # at most the end-point derivative (heat: no depth-5 kernels, evaluated
        # at accept time)
        assert _lazy_state(lazy) in ((1, True, 3), (0, True, 3))
        assert_equal(lazy.K, eager.K)
        assert _lazy_state(lazy) == (0, True, 3)
        assert_equal(lazy.y, eager.y)
    assert lazy.nfev == eager.nfev


@pytest.mark.parametrize("name", ["Pr8", "Pr7"])
def test_lazy_rows_solve_ivp_with_dense_output(monkeypatch, name):
    """solve_ivp(dense_output=True, t_eval=...) reads every step's interpolant:
    same result, bit for bit, whether the rows were restored or always written;
    also across rejected steps (free controller)"""
    from scipy.integrate import solve_ivp
    N = 64
    mk, y0, rho = _plugin("bruss", N)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", "16")
    t_end = 30.0 / rho
    t_eval = np.linspace(0.0, t_end, 23)
    kw = dict(method=getattr(esq, name), rtol=1e-5, atol=1e-8, t_eval=t_eval,
              dense_output=True)
    a = solve_ivp(mk(), (0.0, t_end), y0, **kw)
    monkeypatch.setenv("ESQ_LAZY_ROWS", "0")
    monkeypatch.setenv("ESQ_LAZY_END", "0")
    b = solve_ivp(mk(), (0.0, t_end), y0, **kw)
    assert a.success and b.success
    assert_equal(a.t, b.t)
    assert_equal(a.y, b.y)
    assert a.nfev == b.nfev
    tc = np.linspace(0.0, t_end, 7)
    assert_equal(a.sol(tc), b.sol(tc))


@pytest.mark.parametrize("name,plugin,N,rows,first", [
    ("Pr8", "bruss", 50, 9, "chain5<0>"), ("Pr8", "bruss", 124, 30, "chain5<0>"),
    ("Pr7", "bruss", 64, 16, "chain"), ("Pr9", "bruss", 64, 16, "chain"),
    ("Pr8", "heat", 130, 30, ""), ("CK5", "bruss", 64, 16, ""), ("Me4", "heat", 100, 12, "")])
def test_end_point_derivative_as_stage_zero_of_the_next_step(monkeypatch, name, plugin, N,
                                                             rows, first):
    """Non-FSAL pairs: `K[-1] = fun(t_new, y_new)` (ref common.py:300-301) is not
    evaluated when the step is accepted but as one more stage in front of the next
    step's first chain sweep (label `chain<D+1><0>`: the stage reads the state
    itself).  States, K rows, `solver.f`, the dense output (its last row is that
    derivative) and nfev equal a context that evaluates it at accept time
    (ESQ_LAZY_END=0), bit for bit; a rejected step re-uses the K[0] it has."""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    # (the counts below are those of a context that launches nothing ahead of time:
    # test_first_launch_ahead_of_time has the other half)
    monkeypatch.setenv("ESQ_LAUNCH_AHEAD", "0")
    lazy = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_LAZY_END", "0")
    eager = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_LAZY_END")
    lazy._dev.profile_enable([0, 1, 2])
    for _ in range(4):
        assert lazy.step() is None and eager.step() is None
        assert lazy.t == eager.t
        # (the partial sums are grouped by whichever sweep forms them)
        assert_allclose(lazy.error_norm_old, eager.error_norm_old, rtol=1e-12)
    fused, plain = _end_point_counts(lazy)
    labels = [row[0] for row in lazy._dev.profile_kernels()]
    if first:
        assert fused == 3 and plain == 0, (fused, plain, labels)
        assert any(lab.startswith(first) and lab.endswith("<0>") for lab in labels), labels
    assert _end_point_counts(eager) == (0, 0)
    # readers in between: the derivative is evaluated for them, once
    assert_equal(lazy.f, eager.f)
    assert _end_point_counts(lazy)[0] == fused
    assert_equal(lazy.y, eager.y)
    sl, se = lazy.dense_output(), eager.dense_output()
    tc = np.linspace(eager.t_old, eager.t, 4)
    assert_equal(sl(tc), se(tc))
    assert_equal(lazy.K, eager.K)
    n_plain = _end_point_counts(lazy)[1]
    for _ in range(3):
        assert lazy.step() is None and eager.step() is None
    assert_equal(lazy.K, eager.K)
    assert_equal(lazy.y, eager.y)
    assert lazy.nfev == eager.nfev
    if first:
        assert _end_point_counts(lazy) == (fused + 2, n_plain + 1)
    # a step that is rejected and retried: K[0] is in memory for the retry
    kw2 = dict(first_step=40 * h, rtol=1e-6, atol=1e-9)
    a = cls(mk(), 0.0, y0, 1.0, **kw2)
    monkeypatch.setenv("ESQ_LAZY_END", "0")
    b = cls(mk(), 0.0, y0, 1.0, **kw2)
    monkeypatch.delenv("ESQ_LAZY_END")
    nfs0 = int(esq.NFS[()])
    # (the two decompose an attempt into different sweeps, so their error norms --
    # and with them the step sizes -- agree to rounding, not to the bit)
    for _ in range(6):
        assert a.step() is None and b.step() is None
        assert_allclose(a.t, b.t, rtol=1e-12)
        assert_allclose(a.h_abs, b.h_abs, rtol=1e-10)
    assert_allclose(a.y, b.y, rtol=1e-9, atol=1e-12)
    assert a.nfev == b.nfev
    assert int(esq.NFS[()]) > nfs0


def _ahead_stats(solver):
    import ctypes
    used, dropped = ctypes.c_long(), ctypes.c_long()
    solver._chk(solver._lib.esq_rk_launch_ahead_stats(solver._ctx, ctypes.byref(used),
                                                      ctypes.byref(dropped)),
                "esq_rk_launch_ahead_stats")
    return used.value, dropped.value


@pytest.mark.parametrize("name,plugin,N,rows", [
    ("Pr8", "bruss", 124, 30), ("Pr8", "heat", 130, 30), ("Pr7", "bruss", 64, 16),
    ("Pr9", "heat", 96, 12), ("Ts5", "heat", 96, 8), ("Ts5", "bruss", 64, 12),
    ("CK5", "bruss", 64, 16)])
def test_first_launch_ahead_of_time(monkeypatch, name, plugin, N, rows):
    """The first launch of the NEXT step goes into the queue behind the error norm of
    the step in flight (before the host has seen the norm where the run sits at
    max_step, else when the step is accepted); its K rows land in spare physical rows
    and the y_new of a whole-step chain in a spare state vector, so everything a
    caller may read after `step()` -- y, f, K, the dense output, the old state -- is
    what a context that launches nothing ahead (ESQ_LAUNCH_AHEAD=0) holds, bit for
    bit; readers in between, a rejected attempt and a changed step size drop the
    launch and the step is taken again in full."""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    a = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_LAUNCH_AHEAD", "0")
    b = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_LAUNCH_AHEAD")
    assert a._launch_ahead and not b._launch_ahead
    for _ in range(5):                     # nobody looks: every launch ahead is used
        assert a.step() is None and b.step() is None
        assert a.t == b.t and a.error_norm_old == b.error_norm_old
    used, dropped = _ahead_stats(a)
    # (CK5: the cheapest program evaluates f(t, y) by a launch of its own at accept
    # time and runs the whole step as ONE chain: nothing to launch ahead)
    every = 0 if name == "CK5" else 1
    assert used == 4 * every and dropped == 0, (used, dropped)
    assert _ahead_stats(b) == (0, 0)
    assert_equal(a.y, b.y)                 # (reading y drops nothing)
    assert a.step() is None and b.step() is None
    assert _ahead_stats(a) == (5 * every, 0)
    # readers after a step: the accepted step's rows, not the next step's
    assert_equal(a.K, b.K)
    assert_equal(a.f, b.f)
    sa, sb = a.dense_output(), b.dense_output()
    tc = np.linspace(b.t_old, b.t, 5)
    assert_equal(sa(tc), sb(tc))
    for _ in range(3):                     # ... and the run goes on, bit for bit
        assert a.step() is None and b.step() is None
        assert a.t == b.t
    assert_equal(a.y, b.y)
    assert_equal(a.K, b.K)
    assert a.nfev == b.nfev
    # a step size the guess did not foresee (the controller shrinks it: the attempt
    # from 40 h is rejected, the retries grow back): launches are dropped, never wrong
    kw2 = dict(first_step=40 * h, max_step=40 * h, rtol=1e-6, atol=1e-9)
    c = cls(mk(), 0.0, y0, 1.0, **kw2)
    monkeypatch.setenv("ESQ_LAUNCH_AHEAD", "0")
    d = cls(mk(), 0.0, y0, 1.0, **kw2)
    monkeypatch.delenv("ESQ_LAUNCH_AHEAD")
    nfs0 = int(esq.NFS[()])
    for _ in range(6):
        assert c.step() is None and d.step() is None
        assert c.t == d.t and c.h_abs == d.h_abs
    assert int(esq.NFS[()]) > nfs0
    assert_equal(c.y, d.y)
    assert_equal(c.K, d.K)
    assert c.nfev == d.nfev
    assert _ahead_stats(c)[1] >= every


@pytest.mark.parametrize("name,plugin,N,rows", [
    ("Pr8", "bruss", 124, 30), ("Pr8", "bruss", 50, 9), ("Pr8", "heat", 130, 30),
    ("Pr7", "bruss", 64, 16)])
def test_last_chain_forms_its_own_input(monkeypatch, name, plugin, N, rows):
    """The chain that ends a step reads (almost) every earlier K row for the solution
    and error sums anyway; where the argument of its first stage needs no other row
    it forms that argument itself, T_0 = y + h * a_i . K (`esq_chain.from_rows`), and
    the chain before it does not write it (`out = NULL`): one vector less written,
    one less read.  The plan is made from the plugin's answers to the library's
    queries (no learning step); states and K rows equal ESQ_CHAIN_FROM_ROWS=0 bit for
    bit, the bytes the launches are designed to move do not."""
    mk, y0, rho = _plugin(plugin, N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    cls = getattr(esq, name)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", str(rows))
    a = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.setenv("ESQ_CHAIN_FROM_ROWS", "0")
    b = cls(mk(), 0.0, y0, 1.0, **kw)
    monkeypatch.delenv("ESQ_CHAIN_FROM_ROWS")
    for s in (a, b):
        assert s.step() is None                       # the first step starts from K[0]
        s._dev.profile_reset()
        s._dev.profile_enable([0, 1, 2])
    for _ in range(4):
        assert a.step() is None and b.step() is None
        assert a.t == b.t
        assert_allclose(a.error_norm_old, b.error_norm_old, rtol=1e-12)
    assert_equal(a.y, b.y)
    assert_equal(a.K, b.K)
    assert a.nfev == b.nfev
    moved = [sum(r[5] for r in s._dev.profile_kernels()) for s in (a, b)]
    labels = [r[0] for r in a._dev.profile_kernels()]
    if any("+solerr" in lab and lab.startswith("chain") for lab in labels):
        # one vector less written and read per step, two more halo rows of the other
        # rows: a clear saving on tall tiles, a small one on short tiles
        vec = 8.0 * y0.size * 4
        # (the planner may use the form to restructure the whole step -- Pr7: a plain
        # RHS launch between the chains -- so the saving is asserted as such, and as
        # "one vector less each way" where both plans have the same launches)
        same = sorted(r[0] for r in a._dev.profile_kernels()) == \
            sorted(r[0] for r in b._dev.profile_kernels())
        assert moved[0] < moved[1] - (1.0 if rows >= 16 and same else 0.0) * vec, (moved, labels)
    else:
        assert moved[0] == moved[1]


def test_chain_entry_without_declared_capabilities(monkeypatch):
    """`esq_set_rhs_chain(ctx, fn, caps)`: the library asks a chain entry only for the
    optional forms it has declared (ESQ_CHAIN_CAP_*).  With caps = 0 -- a plugin
    written against plain chains -- every K row is written, every chain reads its
    input from y_in, the end-point derivative has its own sweep; states and K rows
    equal the full-capability run bit for bit."""
    N = 124
    mk, y0, rho = _plugin("bruss", N)
    h = 0.4 / rho
    kw = dict(first_step=h, max_step=h, rtol=1e-4, atol=1e-7)
    monkeypatch.setenv("ESQ_CHAIN_ROWS", "30")

    class PlainChains(esq.Brusselator2D):
        _chain_caps = 0

    full = esq.Pr8(mk(), 0.0, y0, 1.0, **kw)
    plain = esq.Pr8(PlainChains(N), 0.0, y0, 1.0, **kw)
    for s in (full, plain):
        s._dev.profile_enable([0, 1, 2])
    for _ in range(4):
        assert full.step() is None and plain.step() is None
        assert full.t == plain.t
        assert_allclose(full.error_norm_old, plain.error_norm_old, rtol=1e-12)
    assert _lazy_state(plain)[0] == 0 and _lazy_state(full)[0] >= 2
    assert_equal(full.y, plain.y)
    assert_equal(full.K, plain.K)
    labels = [r[0] for r in plain._dev.profile_kernels()]
    assert any(lab.startswith("chain") for lab in labels), labels
    assert not any("-K<" in lab or lab.startswith("chain5<0>") for lab in labels), labels
    assert _end_point_counts(plain) == (0, 0)
    moved = [sum(r[5] for r in s._dev.profile_kernels()) for s in (full, plain)]
    assert moved[0] < moved[1]
