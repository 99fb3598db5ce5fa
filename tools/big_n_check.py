import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import extensisq_amd as esq
from extensisq_amd import workloads as wl
N = 7070
rhs, y0, h = wl.pr8_brusselator(N)
print("n =", y0.size, "slab GB ~", 25 * y0.size * 8 / 1e9)
kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
a = esq.Pr8(rhs, 0.0, y0, 1e9, **kw)
os.environ["ESQ_CHAIN"] = "0"
b = esq.Pr8(esq.Brusselator2D(N), 0.0, y0, 1e9, **kw)
del os.environ["ESQ_CHAIN"]
b._prelaunch = False
for _ in range(3):
    assert a.step() is None and b.step() is None
assert a.t == b.t
print("err norms", a.error_norm_old, b.error_norm_old)
ya, yb = a.y, b.y
print("y identical:", np.array_equal(ya, yb))
for row in (1, 6, 12, 13):
    print("K row", row, "identical:", np.array_equal(a._dev.download_last_K(row), b._dev.download_last_K(row)))
a._dev.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    a.step()
a._dev.synchronize()
dt = (time.perf_counter() - t0) / 20
print("fused  ms/step %.3f  -> %.3e state-dim*steps/s, %.2f TB/s designed" % (dt * 1e3, y0.size / dt, 97 * 8 * y0.size / dt / 1e12))
b._dev.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    b.step()
b._dev.synchronize()
dt = (time.perf_counter() - t0) / 10
print("unfused ms/step %.3f" % (dt * 1e3))
