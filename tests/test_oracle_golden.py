"""Pin the CPU oracle (oracle/) against golden vectors recorded from the real
reference (tools/gen_golden.py) and against the reference's published known
answers (README example; docs/Demo_BS5.ipynb:137,175 nfev; the integer table of
docs/Demo_SSV2stab.ipynb:350-356).

Tolerances: single step from identical (t, y, f, h): K and y_new to 1e-13
relative (max-norm), error_norm to 1e-12 relative; trajectories: identical step
counts / nfev, t_k to 1e-9 relative (SURVEY.md §7 "parity definition" -- the
reference is not bit-stable against its own BLAS thread count).
"""
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy.integrate import solve_ivp

from oracle import problems as pb
from oracle import rk_oracle, rkc_oracle
from tools_cases import bruss1d, compare_trajectory, single_step_cases

ERK = ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "CK5", "Me4", "CFMR7osc"]


@pytest.fixture(scope="module")
def single(golden_dir):
    return np.load(os.path.join(golden_dir, "erk_single_step.npz"))


@pytest.fixture(scope="module")
def traces(golden_dir):
    with open(os.path.join(golden_dir, "erk_traces.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", ERK)
@pytest.mark.parametrize("pname", list(single_step_cases()))
@pytest.mark.parametrize("direction", ["fwd", "bwd"])
def test_single_step(single, name, pname, direction):
    fun, t0, y0, h = single_step_cases()[pname]
    sign = 1 if direction == "fwd" else -1
    key = f"{name}/{pname}/{direction}"
    s = rk_oracle.METHODS[name](fun, t0, y0, t0 + sign * 10.0,
                                first_step=abs(h), rtol=1e-6, atol=1e-9,
                                nfev_stiff_detect=0)
    assert s.step() is None
    K = single[key + "/K"]
    scale = np.abs(K).max()
    assert_allclose(s.K, K, rtol=0, atol=1e-13 * scale)
    assert_allclose(s.y, single[key + "/y_new"], rtol=1e-13, atol=1e-300)
    assert s.t == float(single[key + "/t_new"])
    assert_allclose(s.error_norm_old, float(single[key + "/error_norm"]),
                    rtol=1e-12)
    assert_allclose(s.h_abs, float(single[key + "/h_abs_next"]), rtol=1e-12)
    assert s.nfev == int(single[key + "/nfev"])
    assert int(rk_oracle.NFS[()]) == int(single[key + "/nfs"])


def _run(name, fun, t_span, y0, **kw):
    res = solve_ivp(fun, t_span, y0, method=rk_oracle.METHODS[name], **kw)
    return res, int(rk_oracle.NFS[()])


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
def test_trajectory(traces, name, case):
    fun, t_span, y0, kw = CASES[case]
    res, nfs = _run(name, fun, t_span, y0, **kw)
    compare_trajectory(res, nfs, traces[name][case], kw.get("rtol", 1e-3),
                       t_rtol=1e-9, noisy=case == "duffing_tight")


def test_published_known_answers(traces):
    # README.md:29-30 example: t grid and nfev of BS5
    g = traces["BS5"]["readme"]
    assert g["nfev"] == 40 and g["nfs"] == 0
    assert_allclose(g["t"], [0, 0.38027594845942564, 3.5949909992307925,
                             6.671517050825296, 8.335758525412647, 10.0],
                    rtol=1e-14)
    # docs/Demo_BS5.ipynb:137,175
    assert traces["BS5"]["duffing"]["nfev"] == 212
    assert traces["Ts5"]["duffing"]["nfev"] == 341
    res, _ = _run("BS5", pb.duffing_rhs, [0.0, 20.0], [0.0, 0.0])
    assert res.nfev == 212
    res, _ = _run("Ts5", pb.duffing_rhs, [0.0, 20.0], [0.0, 0.0])
    assert res.nfev == 341


@pytest.mark.parametrize("name", ERK)
def test_dense_output(traces, name):
    g = traces[name]["rational_dense"]
    res = solve_ivp(pb.rational_rhs, [5, 9], [1 / 3, 2 / 9],
                    method=rk_oracle.METHODS[name], dense_output=True)
    assert_allclose(res.sol(np.array(g["tc"])), g["yc"], rtol=1e-10)


@pytest.mark.parametrize("interp", ["free", "low", "best"])
def test_bs5_interpolants(traces, interp):
    g = traces["BS5"]["dense_" + interp]
    res = solve_ivp(pb.rational_rhs, [5, 9], [1 / 3, 2 / 9],
                    method=rk_oracle.BS5, dense_output=True,
                    interpolant=interp)
    assert res.nfev == g["nfev"]
    assert_allclose(res.sol(np.array(g["tc"])), g["yc"], rtol=1e-10)


# ---------------------------------------------------------------- SSV2stab --
@pytest.mark.parametrize("m", [2, 3, 10, 100, 132])
def test_rkc_stages(golden_dir, m):
    g = np.load(os.path.join(golden_dir, "rkc_stages.npz"))
    lam, yn, fn = g[f"m{m}/lam"], g[f"m{m}/yn"], g[f"m{m}/fn"]
    h = float(g[f"m{m}/h"])
    fun = lambda t, y: lam * y + np.sin(t)  # noqa: E731
    s = rkc_oracle.SSV2stab(fun, 0.0, yn, 1.0, first_step=1e-3,
                            rho_jac=lambda t, y: 50.0)
    y, w1, w2 = np.empty_like(yn), np.empty_like(yn), np.empty_like(yn)
    s._stages(0.0, yn.copy(), fn, h, m, y, w1, w2)
    assert_allclose(y, g[f"m{m}/y"], rtol=1e-13, atol=1e-13 * np.abs(y).max())


@pytest.mark.parametrize("tol,expect", [
    (1e-1, (6, 1, 402, 132)),      # docs/Demo_SSV2stab.ipynb:350-356
    (1e-2, (15, 4, 729, 85)),
    (1e-3, (27, 2, 786, 40)),
    (1e-4, (57, 0, 1087, 26)),
    (1e-5, (129, 1, 1682, 20)),
    (1e-6, (262, 0, 2445, 12)),
])
def test_rkc_published_table(golden_dir, tol, expect):
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)[f"tanh3d_tol{tol:.0e}"]
    assert (gold["steps"], gold["nfs"], gold["nfev"], gold["maxm"]) == expect
    fun, y0, rho = pb.tanh3d_problem(39)
    res = solve_ivp(fun, (0, 0.7), y0, method=rkc_oracle.SSV2stab, rtol=tol,
                    atol=tol, const_jac=True, rho_jac=rho)
    nfs = int(rkc_oracle.nrejct[()])
    got = (int(res.t.size - 1 + nfs), nfs, int(res.nfev),
           int(rkc_oracle.maxm[()]))
    assert got == expect
    assert_allclose(res.t, gold["t"], rtol=1e-9)
    assert_allclose(res.y[::5000, -1], gold["y_probe"], rtol=1e-7)


@pytest.mark.parametrize("tol,expect", [
    (1e-4, (51, 1, 525, 21, 36)),  # docs/Demo_SSV2stab.ipynb:207-211:
    (1e-5, (124, 0, 781, 27, 29)),  # steps (failed) / f-evals / f-sigma / s-max
    (1e-6, (270, 0, 1270, 39, 20)),
    (1e-7, (581, 0, 2147, 65, 14)),
])
def test_rkc_published_combustion_table(golden_dir, tol, expect):
    """the 3-D combustion problem (n = 128 000, spectral radius by the power
    iteration): the published integers and the reference's accepted times"""
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)[f"combustion_tol{tol:.0e}"]
    assert (gold["steps"], gold["nfs"], gold["nfev"], gold["nfesig"],
            gold["maxm"]) == expect
    fun, y0 = pb.combustion3d_problem(40)
    res = solve_ivp(fun, (0, 0.3), y0, method=rkc_oracle.SSV2stab, rtol=tol,
                    atol=tol)
    nfs = int(rkc_oracle.nrejct[()])
    got = (int(res.t.size - 1 + nfs), nfs, int(res.nfev),
           int(rkc_oracle.nfesig[()]), int(rkc_oracle.maxm[()]))
    assert got == expect
    assert_allclose(res.t, gold["t"], rtol=1e-8)
    assert_allclose(res.y[::4001, -1], gold["y_probe"], rtol=1e-6)


def test_rkc_power_iteration(golden_dir):
    with open(os.path.join(golden_dir, "rkc_traces.json")) as fh:
        gold = json.load(fh)["heat2d_rho_power"]
    N = 24
    res = solve_ivp(pb.heat2d_rhs(N), (0, 0.01), pb.heat2d_y0(N, seed=1234),
                    method=rkc_oracle.SSV2stab, rtol=1e-4, atol=1e-6)
    nfs = int(rkc_oracle.nrejct[()])
    assert (int(res.t.size - 1 + nfs), nfs, int(res.nfev),
            int(rkc_oracle.maxm[()]), int(rkc_oracle.nfesig[()])) == (
        gold["steps"], gold["nfs"], gold["nfev"], gold["maxm"], gold["nfesig"])
    assert_allclose(res.t, gold["t"], rtol=1e-9)
    assert_allclose(res.y[:, -1], gold["y_end"], rtol=1e-8, atol=1e-12)


# ---------------------------------------------------------------- lockstep --
def test_lockstep_reference(golden_dir):
    """oracle Pr9 on the 8-way concatenated heat problem == reference"""
    g = np.load(os.path.join(golden_dir, "lockstep.npz"))
    N = int(g["N"])
    n = N * N
    f1 = pb.heat2d_rhs(N)
    y0 = np.concatenate([pb.heat2d_y0(N, seed=int(s)) for s in g["seeds"]])

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(8)])

    s = rk_oracle.Pr9(fun, 0.0, y0, float(g["t_end"]), rtol=1e-6, atol=1e-9,
                      nfev_stiff_detect=0)
    assert_allclose(s.h_abs, float(g["h0"]), rtol=1e-12)
    ts, errs = [], []
    while s.status == "running":
        s.step()
        ts.append(s.t)
        errs.append(s.error_norm_old)
    assert_allclose(ts, g["t"], rtol=1e-10)
    assert_allclose(errs, g["err"], rtol=1e-6)
    assert_allclose(s.y, g["y_end"], rtol=1e-9, atol=1e-12)
    assert s.nfev == int(g["nfev"])


# ------------------------------------------------------------------ CKdisc --
def _ckdisc_check(cls, case, gold, nfs_counter, t_rtol):
    from tools_cases import ckdisc_cases
    fun, t_span, y0, kw = ckdisc_cases()[case]
    res = solve_ivp(fun, t_span, y0, method=cls, dense_output=True, **kw)
    g = gold[case]
    noisy = case == "sawtooth"      # 80 rejected steps around discontinuities
    assert res.status == g["status"]
    if noisy:
        assert abs(res.nfev - g["nfev"]) <= 0.02 * g["nfev"]
    else:
        assert res.nfev == g["nfev"] and int(nfs_counter[()]) == g["nfs"]
        assert_allclose(res.t, g["t"], rtol=t_rtol)
    y_end = np.array(g["y_end_re"]) + 1j * np.array(g["y_end_im"])
    tol = kw.get("rtol", 1e-3)
    assert_allclose(res.y[:, -1], y_end if np.iscomplexobj(res.y) else y_end.real,
                    rtol=(50 if noisy else 1e-3) * tol, atol=1e-9)
    dense = np.array(g["dense"]["re"]) + 1j * np.array(g["dense"]["im"])
    got = res.sol(np.array(g["dense"]["tc"]))
    assert_allclose(got, dense if np.iscomplexobj(got) else dense.real,
                    rtol=(50 if noisy else 1e-2) * tol, atol=1e-8)
    s = cls(fun, t_span[0], y0, t_span[1], **kw)
    orders = []
    while s.status == "running":
        s.step()
        orders.append(int(s.order_accepted))
    if noisy:
        assert set(orders) == set(g["orders"])       # 5th order + both fall-backs
    else:
        assert orders == g["orders"]


@pytest.mark.parametrize("case", ["readme", "duffing", "rational_bwd", "complex",
                                  "sawtooth", "kink", "bruss1d"])
def test_ckdisc_trajectory(golden_dir, case):
    with open(os.path.join(golden_dir, "ckdisc_traces.json")) as fh:
        gold = json.load(fh)
    _ckdisc_check(rk_oracle.CKdisc, case, gold, rk_oracle.NFS, 1e-9)


# --------------------------------------------- PDE workloads (real reference) --
PDE_CASES = {"bruss8": ("bruss2d", 8), "heat8": ("heat2d", 8),
             "bruss130": ("bruss2d", 130), "heat130": ("heat2d", 130)}


def pde_problem(case):
    kind, N = PDE_CASES[case]
    return (getattr(pb, kind + "_rhs")(N), getattr(pb, kind + "_y0")(N),
            getattr(pb, kind + "_rho")(N), N)


@pytest.mark.parametrize("name", ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "CK5", "Me4",
                                  "CFMR7osc"])
@pytest.mark.parametrize("case", list(PDE_CASES))
def test_pde_steps_reference(golden_dir, name, case):
    """three fixed steps on the benchmark's PDE workloads: oracle == reference"""
    g = np.load(os.path.join(golden_dir, "pde_steps.npz"))
    fun, y0, rho, N = pde_problem(case)
    key = f"{case}/{name}"
    h = float(g[key + "/h"])
    s = rk_oracle.METHODS[name](fun, 0.0, y0, 1.0, first_step=h, max_step=h,
                                rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
    for _ in range(3):
        assert s.step() is None
    small = y0.size <= 512
    assert s.t == float(g[key + "/t"]) and s.nfev == int(g[key + "/nfev"])
    y = s.y if small else s.y[::97]
    K = s.K[:s.n_stages + s.FSAL]
    K = K if small else K[:, ::97]
    assert_allclose(y, g[key + "/y"], rtol=1e-13, atol=1e-15)
    assert_allclose(K, g[key + "/K"], rtol=0, atol=1e-11 * np.abs(g[key + "/K"]).max())
    assert_allclose(s.error_norm_old, g[key + "/err"][-1], rtol=1e-6)


@pytest.mark.parametrize("case", ["heat8", "heat130", "diff12"])
def test_pde_steps_reference_rkc(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "pde_steps.npz"))
    if case == "diff12":
        fun, y0, rho = pb.diff3d_rhs(12), pb.diff3d_y0(12), 12.0 * 13 ** 2
    else:
        fun, y0, rho, _N = pde_problem(case)
    s = rkc_oracle.SSV2stab(fun, 0.0, y0, 1.0, rtol=1e-4, atol=1e-7,
                            first_step=40.0 / rho, rho_jac=lambda t, y: rho,
                            const_jac=True)
    for _ in range(2):
        assert s.step() is None
    key = f"{case}/SSV2stab"
    assert s.nfev == int(g[key + "/nfev"])
    assert int(rkc_oracle.maxm[()]) == int(g[key + "/maxm"])
    assert_allclose(s.t, float(g[key + "/t"]), rtol=1e-12)
    y = s.y if y0.size <= 2048 else s.y[::97]
    assert_allclose(y, g[key + "/y"], rtol=1e-11, atol=1e-14)
