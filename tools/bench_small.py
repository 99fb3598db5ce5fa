#!/usr/bin/env python3
"""Where does the device path pay?  Wall time per accepted Pr8 step on the 2-D
Brusselator for growing state sizes: device RHS (state in HBM), Python RHS on the
device classes (host-RHS mode: pinned host slab up to 8192 doubles, device slab
beyond) and the NumPy oracle.  Run on the GPU box:  python tools/bench_small.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)

import extensisq_amd as esq  # noqa: E402
from oracle import problems as pb  # noqa: E402
from oracle import rk_oracle  # noqa: E402


def per_step(make, budget_s=1.0, min_steps=10, max_steps=400):
    s = make()
    for _ in range(5):
        assert s.step() is None
    t0 = time.perf_counter()
    k = 0
    while k < min_steps or (time.perf_counter() - t0 < budget_s and k < max_steps):
        assert s.step() is None
        k += 1
    return (time.perf_counter() - t0) / k


def device_time(make, steps=50):
    """summed kernel device time per step and launches per step (HIP events on
    every launch) -- what a launch-free replay of the same kernels would cost at
    the very least, before the ~1.5 us dependent-kernel boundaries"""
    s = make()
    for _ in range(5):
        assert s.step() is None
    dev = s._dev
    dev.profile_reset()
    dev.profile_enable([0, 1, 2, 3], every=1)
    for _ in range(steps):
        assert s.step() is None
    dev.profile_enable(None)
    rows = dev.profile_kernels()
    launches = sum(r[2] for r in rows)
    ms = sum(r[3] for r in rows)
    return 1e3 * ms / steps, launches / steps + 1        # + the final sum


def main():
    rows = []
    for N in (10, 14, 32, 64, 100, 316, 1000, 2236):
        n = 2 * N * N
        y0 = pb.bruss2d_y0(N)
        h = 1.0 / pb.bruss2d_rho(N)
        kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
        cpu_rhs = pb.bruss2d_rhs(N)
        row = {"N": N, "n": n,
               "device_rhs_us": 1e6 * per_step(
                   lambda: esq.Pr8(esq.Brusselator2D(N), 0.0, y0, 1e9, **kw)),
               "host_rhs_us": 1e6 * per_step(
                   lambda: esq.Pr8(cpu_rhs, 0.0, y0, 1e9, **kw)),
               "oracle_us": 1e6 * per_step(
                   lambda: rk_oracle.Pr8(cpu_rhs, 0.0, y0, 1e9, **kw))}
        row["device_kernels_us"], row["launches"] = device_time(
            lambda: esq.Pr8(esq.Brusselator2D(N), 0.0, y0, 1e9, **kw))
        rows.append(row)
        print(json.dumps(row), flush=True)
    print("\n| n | device RHS (us/step) | launches | summed kernel time (us) | Python RHS on "
          "the device classes | NumPy oracle |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['n']} | {r['device_rhs_us']:.0f} | {r['launches']:.0f} | "
              f"{r['device_kernels_us']:.0f} | {r['host_rhs_us']:.0f} | {r['oracle_us']:.0f} |")


if __name__ == "__main__":
    main()
