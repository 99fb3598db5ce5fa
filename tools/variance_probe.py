#!/usr/bin/env python3
"""where does the run-to-run spread of the Pr8 chain sweeps come from?  (`chain4<4>` reads
158 or 175 us per PROCESS with one binary on one box: profiles/r06_experiments.md section 9)
    python tools/variance_probe.py [instances] [steps]
Several solver instances in ONE process, the library's cached device memory released in
between and a block of varying size allocated first (another placement of the slab); per
instance the per-kernel device times and the shader / memory clocks the driver reports."""
import glob
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import extensisq_amd as esq  # noqa: E402
from extensisq_amd import _lib, workloads as wl  # noqa: E402

inst = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = 2236


def clocks():
    out = []
    for kind in ("sclk", "mclk", "fclk"):
        cur = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_%s" % kind)):
            try:
                cur += [ln.split(":")[1].strip().rstrip(" *") for ln in open(f) if ln.strip().endswith("*")]
            except OSError:
                pass
        out.append("%s=%s" % (kind, "/".join(cur) or "?"))
    return " ".join(out)


rhs = esq.Brusselator2D(N)
y0 = wl.bruss2d_y0(N)
h = 1.0 / rhs.spectral_radius()
spacers = []
for k in range(inst):
    s = esq.Pr8(rhs, 0.0, y0, 1e9, first_step=h, max_step=h, rtol=0.1, atol=1e3, nfev_stiff_detect=0)
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
    print("pid %d instance %d: %.4f ms/step  " % (os.getpid(), k, 1e3 * wall)
          + "  ".join("%s=%.1f" % (r[0], 1e3 * r[3] / r[2]) for r in rows), flush=True)
    dev = s._dev
    del s
    dev.close()
    _lib.release_cached_memory()
    # shift the next slab: a block of another size stays allocated
    spacers.append(esq.DeviceContext(1 + (k + 1) * 3_000_017, 1))
