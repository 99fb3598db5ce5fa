import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import extensisq_amd as esq
from extensisq_amd import workloads as wl
# kernel table of one method:  python tools/method_profile.py <method> [bruss2d|heat2d] [N]
name = sys.argv[1]
plugin = sys.argv[2] if len(sys.argv) > 2 else "bruss2d"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2236
if plugin == "heat2d":
    rhs = esq.Heat2D(N); y0 = wl.heat2d_y0(N)
else:
    rhs = esq.Brusselator2D(N); y0 = wl.bruss2d_y0(N)
h = 1.0 / rhs.spectral_radius()
s = getattr(esq, name)(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=1e-3, atol=1e-6, nfev_stiff_detect=0)
for _ in range(8): assert s.step() is None
nfs0 = int(esq.NFS[()]); nf0 = s.nfev
s._dev.profile_reset(); s._dev.profile_enable([0,1,2])
s._dev.synchronize(); t0 = time.perf_counter()
for _ in range(40): assert s.step() is None
s._dev.synchronize(); dt = (time.perf_counter()-t0)/40
s._dev.profile_enable(None)
print(name, "ms/step %.4f" % (dt*1e3), "rejected", int(esq.NFS[()])-nfs0, "nfev/step", (s.nfev-nf0)/40, "h", s.h_abs/h)
for r in s._dev.profile_kernels(): print("   %-22s x%-4d %8.1f us" % (r[0], r[2], 1e3*r[3]/r[2]))
