#!/usr/bin/env python3
"""ms per accepted step of every explicit pair on the two 2-D plugins at the
BASELINE grid (N = 2236) -- and on the 3-D plugin at N = 159 -- with fixed steps
h = 1/rho:  python tools/method_sweep.py [steps]   (ESQ_* knobs from the environment)"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import extensisq_amd as esq  # noqa: E402
from extensisq_amd import workloads as wl  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cases = [("bruss2d", lambda: esq.Brusselator2D(2236), lambda: wl.bruss2d_y0(2236)),
         ("heat2d", lambda: esq.Heat2D(2236), lambda: wl.heat2d_y0(2236)),
         ("diff3d", lambda: esq.Diffusion3D(159), lambda: wl.diff3d_y0(159))]
for pname, mk, y0f in cases:
    y0 = y0f()
    for name in ("BS5", "Ts5", "CK5", "Me4", "Pr7", "Pr8", "Pr9", "CFMR7osc"):
        rhs = mk()
        h = 1.0 / rhs.spectral_radius()
        s = getattr(esq, name)(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=1e-3,
                               atol=1e-6, nfev_stiff_detect=0)
        for _ in range(8):
            assert s.step() is None
        s._dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            assert s.step() is None
        s._dev.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("%-8s %-9s n=%-9d %8.4f ms/step  %6.2f us/stage" % (
            pname, name, y0.size, dt * 1e3, dt * 1e6 / getattr(esq, name).n_stages), flush=True)
        del s
