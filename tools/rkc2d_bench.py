#!/usr/bin/env python3
"""SSV2stab on the 2-D heat plugin, m = 100 stages per step (tolerances loose enough
that the stability limit, not the error, sets the step):
    ESQ_RKC_DEPTH=d python tools/rkc2d_bench.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import extensisq_amd as esq  # noqa: E402
from extensisq_amd import workloads as wl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
rhs = esq.Heat2D(N)
rho = rhs.spectral_radius()
h0 = (100 ** 2 - 1) / (1.54 * rho) * 0.999
s = esq.SSV2stab(rhs, 0.0, wl.heat2d_y0(N), 1e9, rtol=1.0, atol=1.0, const_jac=True,
                 first_step=h0, max_step=h0, rho_jac=lambda t, y: rho)
for _ in range(3):
    assert s.step() is None
s._dev.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    assert s.step() is None
s._dev.synchronize()
dt = (time.perf_counter() - t0) / 10
s._dev.profile_reset()
s._dev.profile_enable([2, 3])
assert s.step() is None
s._dev.profile_enable(None)
print("heat2d N=%d depth %s: %.3f ms/step, %d stages" % (
    N, os.environ.get("ESQ_RKC_DEPTH", "default"), dt * 1e3, s.nfev // 14),
    [(k[0], k[2], round(1e3 * k[3] / k[2], 1)) for k in s._dev.profile_kernels()])
