#!/usr/bin/env python3
"""per-kernel device times (HIP events on every launch) and wall ms/step of one explicit
pair on a built-in plugin at fixed steps -- for A/B runs of library builds (ESQ_LIB=...,
what-if builds of tools/whatif_build.sh included: tolerances are wide open so that a build
with wrong results still takes its steps):
    python tools/kernel_times.py [Pr8] [bruss|heat] [N] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import extensisq_amd as esq  # noqa: E402
from extensisq_amd import workloads as wl  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Pr8"
plug = sys.argv[2] if len(sys.argv) > 2 else "bruss"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2236
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rhs = esq.Brusselator2D(N) if plug == "bruss" else esq.Heat2D(N)
y0 = wl.bruss2d_y0(N) if plug == "bruss" else wl.heat2d_y0(N)
h = 1.0 / rhs.spectral_radius()
s = getattr(esq, name)(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=0.1, atol=1e3,
                       nfev_stiff_detect=0)
for _ in range(8):
    assert s.step() is None
s._dev.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    assert s.step() is None
s._dev.synchronize()
wall = (time.perf_counter() - t0) / steps
s._dev.profile_enable([0, 1, 2])
for _ in range(steps):
    assert s.step() is None
rows = s._dev.profile_kernels()
print("%s %s N=%d lib=%s: %.4f ms/step  " % (name, plug, N, os.path.basename(os.environ.get("ESQ_LIB", "product")), 1e3 * wall)
      + "  ".join("%s=%.1f" % (r[0], 1e3 * r[3] / r[2]) for r in rows), flush=True)
