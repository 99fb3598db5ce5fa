#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (build container only).

Imports extensisq v0.6.0 from /root/reference (read-only; it never travels to
the GPU box) and records inputs + outputs of the hot path as plain numbers in
tests/golden/.  The problems themselves come from oracle/problems.py, so a
fixture is (seeded input, reference output) and nothing else.

Files written
  erk_single_step.npz   G2: one attempted step from (t0, y0, h) per method and
                        problem: K, y_new, error_norm, next h_abs, nfev
  erk_traces.json       G3: full solve_ivp runs (README example, Duffing,
                        rational both directions, complex decay): t, nfev, NFS,
                        y_end, per-attempt error norms
  rkc_stages.npz        G4a: SSV2stab._stages outputs, m in {2,3,10,100,132}
  rkc_traces.json       G4b: SSV2stab runs on the tanh heat problem
                        (published integer table) and with the power iteration
  lockstep.npz          G5: Pr9 on 8 concatenated heat problems (lock-step ref)
  pde_steps.npz         G7: three fixed steps of every ERK method (two of SSV2stab)
                        on the 2-D Brusselator / heat (3-D diffusion) workloads

Usage:  OPENBLAS_NUM_THREADS=1 python tools/gen_golden.py
"""
import json
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np  # noqa: E402

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from scipy.integrate import solve_ivp  # noqa: E402
import extensisq as ref  # noqa: E402
from extensisq.sommeijer import maxm, nfesig  # noqa: E402
from oracle import problems as pb  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_cases import bruss1d, single_step_cases  # noqa: E402  (seeded inputs)

GOLD = os.path.join(ROOT, "tests", "golden")
ERK = ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "CK5", "Me4", "CFMR7osc"]


def gen_single_step():
    out = {}
    for name in ERK:
        cls = getattr(ref, name)
        for pname, (fun, t0, y0, h) in single_step_cases().items():
            for sign in (+1, -1):
                tb = t0 + sign * 10.0
                s = cls(fun, t0, y0, tb, first_step=abs(h), rtol=1e-6,
                        atol=1e-9, nfev_stiff_detect=0)
                f0 = s.f.copy()
                msg = s.step()
                key = f"{name}/{pname}/{'fwd' if sign > 0 else 'bwd'}"
                out[key + "/t0"] = t0
                out[key + "/y0"] = y0
                out[key + "/f0"] = f0
                out[key + "/h"] = s.h_previous
                out[key + "/K"] = s.K.copy()
                out[key + "/y_new"] = s.y.copy()
                out[key + "/t_new"] = s.t
                out[key + "/error_norm"] = s.error_norm_old
                out[key + "/h_abs_next"] = s.h_abs
                out[key + "/nfev"] = s.nfev
                out[key + "/nfs"] = int(ref.NFS[()])
                assert msg is None
    np.savez_compressed(os.path.join(GOLD, "erk_single_step.npz"), **out)
    print("erk_single_step:", len(out), "arrays")


def run_trace(cls, fun, t_span, y0, **kw):
    """drive the reference step by step, recording attempts"""
    res = solve_ivp(fun, t_span, y0, method=cls, **kw)
    nfs = int(ref.NFS[()])
    return {
        "t": [float(v) for v in res.t],
        "y_end_re": [float(np.real(v)) for v in res.y[:, -1]],
        "y_end_im": [float(np.imag(v)) for v in res.y[:, -1]],
        "nfev": int(res.nfev), "nfs": nfs, "status": int(res.status),
    }


def gen_traces():
    out = {}
    for name in ERK:
        cls = getattr(ref, name)
        d = {}
        d["readme"] = run_trace(cls, lambda t, y: -0.5 * y, [0, 10], [2, 4, 8])
        d["duffing"] = run_trace(cls, pb.duffing_rhs, [0.0, 20.0], [0.0, 0.0])
        d["duffing_tight"] = run_trace(cls, pb.duffing_rhs, [0.0, 20.0],
                                       [0.0, 0.0], rtol=1e-9, atol=1e-12)
        d["rational_fwd"] = run_trace(cls, pb.rational_rhs, [5, 9],
                                      [1 / 3, 2 / 9], rtol=1e-3, atol=1e-6)
        d["rational_bwd"] = run_trace(cls, pb.rational_rhs, [5, 1],
                                      [1 / 3, 2 / 9], rtol=1e-3, atol=1e-6)
        d["complex"] = run_trace(cls, lambda t, y: -y, [0, 1], [0.5 + 1j],
                                 rtol=1e-3, atol=1e-6)
        fb, yb = bruss1d()
        d["bruss1d"] = run_trace(cls, fb, [0, 0.5], yb, rtol=1e-6, atol=1e-9)
        # dense-output samples on the rational problem (all interpolants)
        r = solve_ivp(pb.rational_rhs, [5, 9], [1 / 3, 2 / 9], method=cls,
                      dense_output=True)
        tc = np.linspace(5, 9, 11)
        d["rational_dense"] = {"tc": tc.tolist(), "yc": r.sol(tc).tolist()}
        out[name] = d
    for interp in ("free", "low", "best"):
        r = solve_ivp(pb.rational_rhs, [5, 9], [1 / 3, 2 / 9], method=ref.BS5,
                      dense_output=True, interpolant=interp)
        tc = np.linspace(5, 9, 11)
        out["BS5"]["dense_" + interp] = {"tc": tc.tolist(),
                                         "yc": r.sol(tc).tolist(),
                                         "nfev": int(r.nfev)}
    with open(os.path.join(GOLD, "erk_traces.json"), "w") as fh:
        json.dump(out, fh)
    print("erk_traces: nfev duffing",
          {k: v["duffing"]["nfev"] for k, v in out.items()})


def gen_ckdisc():
    """G7: CKdisc (variable order, no error_norm_old): trajectories only, incl.
    a non-smooth RHS that exercises the order-3 and order-2 fall-backs"""
    from tools_cases import ckdisc_cases
    out = {}
    for cname, (fun, t_span, y0, kw) in ckdisc_cases().items():
        d = run_trace(ref.CKdisc, fun, t_span, y0, **kw)
        r = solve_ivp(fun, t_span, y0, method=ref.CKdisc, dense_output=True, **kw)
        tc = np.linspace(t_span[0], t_span[1], 9)
        yc = r.sol(tc)
        d["dense"] = {"tc": tc.tolist(), "re": np.real(yc).tolist(),
                      "im": np.imag(yc).tolist()}
        # which orders were accepted (5th-order step = 4, fall-backs = 2, 1)
        s = ref.CKdisc(fun, t_span[0], y0, t_span[1], **kw)
        orders = []
        while s.status == "running":
            s.step()
            orders.append(int(s.order_accepted))
        d["orders"] = orders
        out[cname] = d
    with open(os.path.join(GOLD, "ckdisc_traces.json"), "w") as fh:
        json.dump(out, fh)
    print("ckdisc:", {k: (v["nfev"], v["nfs"], sorted(set(v["orders"])))
                      for k, v in out.items()})


def gen_rkc():
    out = {}
    rng = np.random.default_rng(64)
    n = 64
    lam = -rng.random(n) * 50.0
    yn = rng.standard_normal(n)
    fun = lambda t, y: lam * y + np.sin(t)  # noqa: E731
    for m in (2, 3, 10, 100, 132):
        s = ref.SSV2stab(fun, 0.0, yn, 1.0, first_step=1e-3,
                         rho_jac=lambda t, y: 50.0)
        y = np.empty(n)
        w1 = np.empty(n)
        w2 = np.empty(n)
        fn = fun(0.0, yn)
        h = 0.6 * m * m / 50.0 / 2
        s._stages(0.0, yn.copy(), fn, h, m, y, w1, w2)
        out[f"m{m}/yn"] = yn
        out[f"m{m}/fn"] = fn
        out[f"m{m}/lam"] = lam
        out[f"m{m}/h"] = h
        out[f"m{m}/y"] = y.copy()
    np.savez_compressed(os.path.join(GOLD, "rkc_stages.npz"), **out)

    traces = {}
    fun3, y03, rho3 = pb.tanh3d_problem(39)
    for tol in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):   # Demo_SSV2stab.ipynb:350-356
        res = solve_ivp(fun3, (0, 0.7), y03, method=ref.SSV2stab, rtol=tol,
                        atol=tol, const_jac=True, rho_jac=rho3)
        nfs = int(ref.NFS[()])
        traces[f"tanh3d_tol{tol:.0e}"] = {
            "steps": int(res.t.size - 1 + nfs), "nfs": nfs,
            "nfev": int(res.nfev), "maxm": int(maxm[()]),
            "t": [float(v) for v in res.t],
            "y_probe": [float(v) for v in res.y[::5000, -1]],
        }
    # combustion table (Demo_SSV2stab.ipynb:207-211): rho_jac=None, so the
    # spectral radius comes from the power iteration (f-sigma column)
    func, y0c = pb.combustion3d_problem(40)
    for tol in (1e-4, 1e-5, 1e-6, 1e-7):
        res = solve_ivp(func, (0, 0.3), y0c, method=ref.SSV2stab, rtol=tol,
                        atol=tol)
        nfs = int(ref.NFS[()])
        traces[f"combustion_tol{tol:.0e}"] = {
            "steps": int(res.t.size - 1 + nfs), "nfs": nfs,
            "nfev": int(res.nfev), "maxm": int(maxm[()]),
            "nfesig": int(nfesig[()]),
            "t": [float(v) for v in res.t],
            "y_probe": [float(v) for v in res.y[::4001, -1]],
        }
    # power-iteration branch (rho_jac=None) on a small 2-D heat problem
    N = 24
    f2 = pb.heat2d_rhs(N)
    y2 = pb.heat2d_y0(N, seed=1234)
    res = solve_ivp(f2, (0, 0.01), y2, method=ref.SSV2stab, rtol=1e-4,
                    atol=1e-6)
    nfs = int(ref.NFS[()])
    traces["heat2d_rho_power"] = {
        "steps": int(res.t.size - 1 + nfs), "nfs": nfs, "nfev": int(res.nfev),
        "maxm": int(maxm[()]), "nfesig": int(nfesig[()]),
        "t": [float(v) for v in res.t],
        "y_end": [float(v) for v in res.y[:, -1]],
    }
    with open(os.path.join(GOLD, "rkc_traces.json"), "w") as fh:
        json.dump(traces, fh)
    print("rkc:", {k: (v["steps"], v["nfs"], v["nfev"], v["maxm"],
                       v.get("nfesig")) for k, v in traces.items()})


def gen_lockstep():
    """reference Pr9 on the concatenation of 8 heat problems (N=24 each):
    this is what the 8-GPU lock-step mode must reproduce."""
    N = 24
    f1 = pb.heat2d_rhs(N)
    n = N * N
    y0 = np.concatenate([pb.heat2d_y0(N, seed=1234 + g) for g in range(8)])

    def fun(t, y):
        return np.concatenate([f1(t, y[g * n:(g + 1) * n]) for g in range(8)])

    s = ref.Pr9(fun, 0.0, y0, 2e-3, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    h0 = s.h_abs
    ts, errs = [], []
    while s.status == "running":
        s.step()
        ts.append(s.t)
        errs.append(s.error_norm_old)
    np.savez_compressed(os.path.join(GOLD, "lockstep.npz"),
                        N=N, seeds=np.arange(1234, 1242), t_end=2e-3,
                        h0=h0, t=np.array(ts), err=np.array(errs),
                        y_end=s.y, nfev=s.nfev, nfs=int(ref.NFS[()]))
    print("lockstep: steps", len(ts), "nfev", s.nfev)


def gen_pde():
    """G7: the REAL reference on the 2-D / 3-D workloads of the benchmark at small
    (even) grid sizes, with the NumPy right-hand sides of oracle/problems.py:
    three fixed-size steps per method.  The device classes are compared with
    these numbers directly -- device RHS plugins, fused sweeps -- not only
    through the oracle."""
    out = {}
    cases = {"bruss8": (pb.bruss2d_rhs(8), pb.bruss2d_y0(8), pb.bruss2d_rho(8)),
             "heat8": (pb.heat2d_rhs(8), pb.heat2d_y0(8), pb.heat2d_rho(8)),
             "bruss130": (pb.bruss2d_rhs(130), pb.bruss2d_y0(130), pb.bruss2d_rho(130)),
             "heat130": (pb.heat2d_rhs(130), pb.heat2d_y0(130), pb.heat2d_rho(130))}
    for cname, (fun, y0, rho) in cases.items():
        h = 0.5 / rho
        small = y0.size <= 512
        for name in ERK:
            s = getattr(ref, name)(fun, 0.0, y0, 1.0, first_step=h, max_step=h,
                                   rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
            errs = []
            for _ in range(3):
                assert s.step() is None
                errs.append(float(s.error_norm_old))
            key = f"{cname}/{name}"
            out[key + "/h"] = h
            out[key + "/t"] = s.t
            out[key + "/err"] = np.array(errs)
            out[key + "/nfev"] = s.nfev
            out[key + "/nfs"] = int(ref.NFS[()])
            out[key + "/y"] = s.y.copy() if small else s.y[::97].copy()
            K = s.K[:s.n_stages + s.FSAL]
            out[key + "/K"] = K.copy() if small else K[:, ::97].copy()
    # SSV2stab on the 2-D heat and 3-D diffusion workloads, two steps
    for cname, fun, y0, rho in (("heat8", pb.heat2d_rhs(8), pb.heat2d_y0(8), pb.heat2d_rho(8)),
                                ("heat130", pb.heat2d_rhs(130), pb.heat2d_y0(130), pb.heat2d_rho(130)),
                                ("diff12", pb.diff3d_rhs(12), pb.diff3d_y0(12), 12.0 * 13 ** 2)):
        s = ref.SSV2stab(fun, 0.0, y0, 1.0, rtol=1e-4, atol=1e-7, first_step=40.0 / rho,
                         rho_jac=lambda t, y, rho=rho: rho, const_jac=True)
        for _ in range(2):
            assert s.step() is None
        key = f"{cname}/SSV2stab"
        out[key + "/t"] = s.t
        out[key + "/nfev"] = s.nfev
        out[key + "/maxm"] = int(maxm[()])
        out[key + "/y"] = s.y.copy() if y0.size <= 2048 else s.y[::97].copy()
    np.savez_compressed(os.path.join(GOLD, "pde_steps.npz"), **out)
    print("pde:", len(out), "arrays")


def gen_stiffness():
    """G6: the reference's stiffness diagnosis (stiff_a, common.py:824-1103) on
    stiff / oscillatory / mild problems: every call's inputs and verdict, and
    the warnings the run emitted."""
    import warnings
    import extensisq.common as rc
    from stiffness_cases import stiffness_cases
    out = {}
    orig = rc.stiff_a
    for cname, (fun, t_span, y0, kw) in stiffness_cases().items():
        for name in ("BS5", "Ts5", "Pr8"):
            calls = []

            def spy(f, x, y, hnow, havg, xend, maxfcn, wt, fxy, v0, cost):
                res = orig(f, x, y, hnow, havg, xend, maxfcn, wt, fxy, v0, cost)
                stif, rootre, root = res
                calls.append({"t": float(x), "hnow": float(hnow),
                              "havg": float(havg),
                              "stif": None if stif is None else bool(stif),
                              "rootre": None if rootre is None else bool(rootre),
                              "roots": None if root is None else
                              [list(map(float, root[0])), list(map(float, root[1])),
                               float(root[2])]})
                return res
            rc.stiff_a = spy
            try:
                with warnings.catch_warnings(record=True) as wlist:
                    warnings.simplefilter("always")
                    res = solve_ivp(fun, t_span, y0, method=getattr(ref, name), **kw)
            finally:
                rc.stiff_a = orig
            out[f"{cname}/{name}"] = {
                "calls": calls, "nfev": int(res.nfev), "nfs": int(ref.NFS[()]),
                "steps": int(res.t.size - 1),
                "warnings": sorted({str(w.message)[:60] for w in wlist}),
            }
    with open(os.path.join(GOLD, "stiffness.json"), "w") as fh:
        json.dump(out, fh)
    print("stiffness:", {k: (len(v["calls"]), v["warnings"][:1], v["nfev"])
                         for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    gen_single_step()
    gen_traces()
    gen_rkc()
    gen_lockstep()
    gen_pde()
    gen_stiffness()
    gen_ckdisc()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
