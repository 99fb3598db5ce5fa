#!/usr/bin/env python3
"""Time the TRUE reference (extensisq v0.6.0 imported from /root/reference) on the
bench.py workloads, with the same inputs and the same harness shape as
bench.py's `cpu_baseline` leg (1 warm-up step, k timed accepted steps, fixed
step size so that every step is accepted).  Build container only: the reference
never travels to the GPU box.  BASELINE.md §4.2.

    python tools/time_reference.py [--steps K] [--configs pr8,ts5,pr9,rkc]

Prints one JSON line per config and a Markdown table for BASELINE.md.  The
oracle (the port bench.py times on the GPU box) is timed beside it so that the
two CPU figures can be related."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import extensisq as ref  # noqa: E402
from oracle import problems as pb  # noqa: E402
from oracle import rk_oracle, rkc_oracle  # noqa: E402


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        return max(int(p["num_threads"]) for p in pools) if pools else None
    except Exception:
        return None


def workload(name):
    """(label, reference class, oracle class, rhs, y0, kwargs) -- the inputs of
    bench.py's make_workload, built from oracle/problems.py only"""
    if name == "pr8":
        N = 2236
        h = 1.0 / pb.bruss2d_rho(N)
        return ("Pr8, 2-D Brusselator N=2236", ref.Pr8, rk_oracle.Pr8,
                pb.bruss2d_rhs(N), pb.bruss2d_y0(N),
                dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
                     nfev_stiff_detect=0))
    if name == "ts5":
        N = 1000
        h = 1.0 / pb.heat2d_rho(N)
        return ("Ts5, 2-D heat N=1000", ref.Ts5, rk_oracle.Ts5, pb.heat2d_rhs(N),
                pb.heat2d_y0(N, seed=1234),
                dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
                     nfev_stiff_detect=0))
    if name == "pr9":
        N = 2236
        h = 1.0 / pb.heat2d_rho(N)
        return ("Pr9, 2-D heat N=2236", ref.Pr9, rk_oracle.Pr9, pb.heat2d_rhs(N),
                pb.heat2d_y0(N, seed=1234),
                dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9,
                     nfev_stiff_detect=0))
    N = 159
    rho = 12.0 * (N + 1) ** 2
    m = 100
    h = ((m - 1) ** 2 - 1 + 0.5 * (2 * m - 1)) / (1.54 * rho)
    return ("SSV2stab (m~100), 3-D diffusion N=159", ref.SSV2stab,
            rkc_oracle.SSV2stab, pb.diff3d_rhs(N), pb.diff3d_y0(N),
            dict(first_step=h, max_step=h, rtol=1e-3, atol=1e-3, const_jac=True,
                 rho_jac=lambda t, y: rho))


def time_steps(cls, fun, y0, kw, steps):
    s = cls(fun, 0.0, y0, 1.0e9, **kw)
    s.step()
    nfev0 = s.nfev
    t0 = time.perf_counter()
    for _ in range(steps):
        msg = s.step()
        assert msg is None and s.status == "running", msg
    dt = (time.perf_counter() - t0) / steps
    return dt, (s.nfev - nfev0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--configs", default="pr8,ts5,pr9,rkc")
    args = ap.parse_args()
    rows = []
    for name in args.configs.split(","):
        label, rcls, ocls, fun, y0, kw = workload(name)
        # the RHS alone, to split a step into RHS and RK-framework time
        t0 = time.perf_counter()
        for _ in range(3):
            fun(0.0, y0)
        t_rhs = (time.perf_counter() - t0) / 3
        dt_ref, nfev = time_steps(rcls, fun, y0, kw, args.steps)
        dt_ora, _ = time_steps(ocls, fun, y0, kw, args.steps)
        row = {"config": name, "workload": label, "n": int(y0.size),
               "steps_timed": args.steps, "host_cpus": os.cpu_count(),
               "blas_threads": blas_threads(),
               "reference_s_per_step": dt_ref, "rhs_evals_per_step": nfev,
               "rhs_s_per_step": t_rhs * nfev,
               "framework_s_per_step": dt_ref - t_rhs * nfev,
               "reference_value": y0.size / dt_ref,
               "oracle_s_per_step": dt_ora, "oracle_value": y0.size / dt_ora}
        rows.append(row)
        print(json.dumps(row), flush=True)
    print()
    print("| config | n | reference s/step | RHS part | RK-framework part | "
          "state-dim x steps/s (reference) | oracle s/step | oracle / reference |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['workload']} | {r['n']} | {r['reference_s_per_step']:.3f} | "
              f"{r['rhs_s_per_step']:.3f} | {r['framework_s_per_step']:.3f} | "
              f"{r['reference_value']:.3g} | {r['oracle_s_per_step']:.3f} | "
              f"{r['oracle_s_per_step'] / r['reference_s_per_step']:.2f} |")


if __name__ == "__main__":
    main()
