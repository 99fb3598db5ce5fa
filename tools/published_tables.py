#!/usr/bin/env python3
"""The only timings the reference ever published -- SSV2stab on its two 3-D demo
problems, docs/Demo_SSV2stab.ipynb:207-211 (combustion, 2 x 40^3) and :350-356 (heat
equation with a travelling tanh front, 39^3) -- like for like: every tolerance row
through plain `solve_ivp(method=SSV2stab)`,

  * with the right-hand side on the device (user plugins on csrc/esq_stencil3d.hpp,
    examples/ssv2stab_demo_plugins.hip), wall seconds of the whole call;
  * with the NumPy oracle on this box's host cores (the reference's algorithm and
    right-hand sides restated: oracle/rkc_oracle.py, oracle/problems.py);
  * beside the notebook's own seconds (the author's machine, unknown).

The integer columns (steps, failed, f-evals, f-sigma, s-max) are asserted against
the published ones for the device run.  --no-oracle skips the CPU side (minutes).

    python tools/published_tables.py [--no-oracle] [--json out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from scipy.integrate import solve_ivp  # noqa: E402

HEAT = [(1e-1, (6, 1, 402, 132), 2.8), (1e-2, (15, 4, 729, 85), 4.7),
        (1e-3, (27, 2, 786, 40), 5.0), (1e-4, (57, 0, 1087, 26), 6.9),
        (1e-5, (129, 1, 1682, 20), 10.9), (1e-6, (262, 0, 2445, 12), 24.3)]
COMBUSTION = [(1e-4, (51, 1, 525, 21, 36), 4.2), (1e-5, (124, 0, 781, 27, 29), 6.0),
              (1e-6, (270, 0, 1270, 39, 20), 8.8), (1e-7, (581, 0, 2147, 65, 14), 13.8)]


def main():
    import extensisq_amd as esq
    from extensisq_amd import sommeijer as dev_rkc
    import examples.ssv2stab_demo_plugins as demo
    from oracle import problems as pb
    from oracle import rkc_oracle
    with_oracle = "--no-oracle" not in sys.argv
    rows = []
    demo.build()
    # (first use of the device: context, plugin library, clocks)
    rhs, y0, rho = demo.tanh_heat(39)
    solve_ivp(rhs, (0, 0.05), y0, method=esq.SSV2stab, rtol=1e-2, atol=1e-2, const_jac=True,
              rho_jac=rho)

    def run(problem, tol, expect, published):
        if problem == "heat":
            rhs, y0, rho = demo.tanh_heat(39)
            kw = dict(const_jac=True, rho_jac=rho)
            span = (0, 0.7)
            cpu = pb.tanh3d_problem(39)
            cpu_fun, cpu_kw = cpu[0], dict(const_jac=True, rho_jac=cpu[2])
        else:
            rhs, y0 = demo.combustion(40)
            kw, span = {}, (0, 0.3)
            cpu_fun, cpu_kw = pb.combustion3d_problem(40)[0], {}
        t0 = time.perf_counter()
        res = solve_ivp(rhs, span, y0, method=esq.SSV2stab, rtol=tol, atol=tol, **kw)
        dev_s = time.perf_counter() - t0
        nfs = int(dev_rkc.nrejct[()])
        got = (int(res.t.size - 1 + nfs), nfs, int(res.nfev))
        got += ((int(dev_rkc.nfesig[()]),) if problem == "combustion" else ()) + (
            int(dev_rkc.maxm[()]),)
        assert got == expect, (problem, tol, got, expect)
        cpu_s = None
        if with_oracle:
            t0 = time.perf_counter()
            ref = solve_ivp(cpu_fun, span, y0, method=rkc_oracle.SSV2stab, rtol=tol, atol=tol,
                            **cpu_kw)
            cpu_s = time.perf_counter() - t0
            assert ref.nfev == res.nfev
            err = float(np.abs(ref.y[:, -1] - res.y[:, -1]).max())
        else:
            err = None
        rows.append(dict(problem=problem, tol=tol, integers=list(got), device_s=dev_s,
                         oracle_s=cpu_s, notebook_s=published, max_abs_diff_vs_oracle=err))
        print(f"{problem:10s} tol {tol:7.0e}  {str(got):28s} device {dev_s:7.3f} s   "
              f"oracle {'-' if cpu_s is None else format(cpu_s, '7.2f')} s   notebook "
              f"{published:5.1f} s" + ("" if err is None else f"   |dy| {err:.1e}"), flush=True)

    for tol, expect, pub in HEAT:
        run("heat", tol, expect, pub)
    for tol, expect, pub in COMBUSTION:
        run("combustion", tol, expect, pub)
    out = dict(rows=rows, host_cpus=os.cpu_count(),
               note="device: MI355X, right-hand side as a user plugin, plain solve_ivp; "
                    "oracle: NumPy restatement on this box's host; notebook: the "
                    "reference author's machine (docs/Demo_SSV2stab.ipynb)")
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
