#!/usr/bin/env python3
"""Where do the marching chain sweeps start to pay?  Wall time per accepted step
(device RHS, 2-D Brusselator) for growing grids, with the plugin's own choice and
with forced tile heights (ESQ_CHAIN_ROWS lifts the small-grid rule).
Run on the GPU box:  python tools/small_chain_sweep.py [bruss|heat] [Pr8|Ts5 ...]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)

import extensisq_amd as esq  # noqa: E402
from oracle import problems as pb  # noqa: E402


PLUGIN = "bruss"


def per_step(cls, N, steps=300):
    if PLUGIN == "heat":
        y0, h, rhs = pb.heat2d_y0(N), 0.5 / pb.heat2d_rho(N), esq.Heat2D(N)
    else:
        y0, h, rhs = pb.bruss2d_y0(N), 0.5 / pb.bruss2d_rho(N), esq.Brusselator2D(N)
    s = cls(rhs, 0.0, y0, 1.0e9, first_step=h, max_step=h, rtol=1e-6,
            atol=1e-9, nfev_stiff_detect=0)
    for _ in range(10):
        assert s.step() is None
    s._dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert s.step() is None
    s._dev.synchronize()
    return 1e6 * (time.perf_counter() - t0) / steps


def main():
    global PLUGIN
    args = sys.argv[1:]
    if args and args[0] in ("heat", "bruss"):
        PLUGIN = args.pop(0)
    names = args or ["Pr8", "Ts5"]
    settings = [("default", {}), ("off", {"ESQ_CHAIN_DEPTH": "1"})]
    for rows in (6, 10, 16, 24, 36, 48):
        settings.append((f"R={rows}", {"ESQ_CHAIN_ROWS": str(rows)}))
    for name in names:
        cls = getattr(esq, name)
        print(name, " ".join("%9s" % lab for lab, _ in settings), "  us/step")
        grids = ((32, 64, 100, 160, 224, 316, 500, 708, 1000, 1416) if PLUGIN == "bruss"
                 else (64, 224, 448, 708, 1000, 1416, 2000, 2236, 3162))
        for N in grids:
            out = []
            for lab, env in settings:
                for k, v in env.items():
                    os.environ[k] = v
                try:
                    out.append(per_step(cls, N, steps=200 if N < 800 else 60))
                finally:
                    for k in env:
                        del os.environ[k]
            print("N=%5d n=%8d" % (N, (2 if PLUGIN == "bruss" else 1) * N * N),
                  " ".join("%9.1f" % v for v in out))


if __name__ == "__main__":
    main()
