#!/usr/bin/env python3
"""process exit with solvers alive and work in flight (no explicit close): must not fault
    python tools/exit_probe.py [ssv|pr8|bs5] [raise]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import extensisq_amd as esq  # noqa: E402
from extensisq_amd import workloads as wl  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "ssv"
keep = []
if kind == "ssv":
    for N in (57, 64):
        rhs = esq.Diffusion3D(N)
        rho = rhs.spectral_radius()
        h = 400.0 / rho
        s = esq.SSV2stab(rhs, 0.0, wl.diff3d_y0(N), 1.0, first_step=h, max_step=h, rtol=1e-2,
                         atol=1e-2, rho_jac=lambda t, y: rho, const_jac=True)
        for _ in range(5):
            s.step()
        keep.append(s)
else:
    rhs = esq.Brusselator2D(500)
    h = 1.0 / rhs.spectral_radius()
    cls = esq.Pr8 if kind == "pr8" else esq.BS5
    s = cls(rhs, 0.0, wl.bruss2d_y0(500), 1e9, first_step=h, max_step=h, rtol=1e-3, atol=1e-6,
            nfev_stiff_detect=0)
    for _ in range(5):
        s.step()
    keep.append(s)
print("alive:", len(keep), flush=True)
if len(sys.argv) > 2:
    raise RuntimeError("leaving with an exception")
