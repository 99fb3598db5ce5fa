#!/usr/bin/env python3
"""where the HOST's share of a small step goes: cProfile over accepted steps of one pair
(default: config 2, Ts5 on the heat plugin at N = 1000):
    python tools/host_profile.py [Ts5|SSV2stab|...] [heat|bruss|diff3d] [N] [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import extensisq_amd as esq  # noqa: E402
from extensisq_amd import workloads as wl  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Ts5"
plug = sys.argv[2] if len(sys.argv) > 2 else "heat"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
if plug == "diff3d":
    rhs, y0 = esq.Diffusion3D(N), wl.diff3d_y0(N)
else:
    rhs = esq.Brusselator2D(N) if plug == "bruss" else esq.Heat2D(N)
    y0 = wl.bruss2d_y0(N) if plug == "bruss" else wl.heat2d_y0(N)
rho = rhs.spectral_radius()
if name == "SSV2stab":                 # about a hundred stages per step, as config 4
    h = 6490.0 / rho
    s = esq.SSV2stab(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=1e-3, atol=1e-3,
                     rho_jac=lambda t, y: rho, const_jac=True)
else:
    h = 1.0 / rho
    s = getattr(esq, name)(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=1e-3, atol=1e-6,
                           nfev_stiff_detect=0)
for _ in range(50):
    s.step()
s._dev.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    s.step()
s._dev.synchronize()
print("plain: %.2f us/step" % (1e6 * (time.perf_counter() - t0) / steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    s.step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
