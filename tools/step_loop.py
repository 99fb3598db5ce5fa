#!/usr/bin/env python3
"""plain solver.step() loop of one method on a built-in plugin at fixed steps (for kernel
traces: `rocprofv3 --kernel-trace -- python3 tools/step_loop.py Pr8 bruss 2236 40`)
    python tools/step_loop.py [method] [bruss|heat|diff3d] [N] [steps]"""
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
if plug == "bruss":
    rhs, y0 = esq.Brusselator2D(N), wl.bruss2d_y0(N)
elif plug == "heat":
    rhs, y0 = esq.Heat2D(N), wl.heat2d_y0(N)
else:
    rhs, y0 = esq.Diffusion3D(N), wl.diff3d_y0(N)
rho = rhs.spectral_radius()
if name == "SSV2stab":
    h = 6490.0 / rho
    s = esq.SSV2stab(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=1e-3, atol=1e-3,
                     rho_jac=lambda t, y: rho, const_jac=True)
else:
    h = 1.0 / rho
    s = getattr(esq, name)(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=0.1, atol=1e3,
                           nfev_stiff_detect=0)
for _ in range(8):
    assert s.step() is None
s._dev.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    assert s.step() is None
s._dev.synchronize()
print("%s %s N=%d: %.4f ms/step" % (name, plug, N, 1e3 * (time.perf_counter() - t0) / steps), flush=True)
