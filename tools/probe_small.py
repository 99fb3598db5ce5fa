#!/usr/bin/env python3
"""Host-side time breakdown of a device-RHS Pr8 step at small sizes: how long the
enqueue call (esq_rk_stages), the wait for the error norm (esq_rk_solution_error)
and the accept call take.  python tools/probe_small.py [N ...]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)

import extensisq_amd as esq  # noqa: E402
from oracle import problems as pb  # noqa: E402


def probe(N, steps=200):
    y0 = pb.bruss2d_y0(N)
    h = 1.0 / pb.bruss2d_rho(N)
    kw = dict(first_step=h, max_step=h, rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    s = esq.Pr8(esq.Brusselator2D(N), 0.0, y0, 1e9, **kw)
    acc = {"stages": 0.0, "solerr": 0.0, "finish": 0.0}
    orig = (s._run_stages, s._solution_and_error, s._finish_step)

    def wrap(name, fn):
        def inner(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            acc[name] += time.perf_counter() - t0
            return out
        return inner

    s._run_stages = wrap("stages", orig[0])
    s._solution_and_error = wrap("solerr", orig[1])
    s._finish_step = wrap("finish", orig[2])
    for _ in range(10):
        assert s.step() is None
    for k in acc:
        acc[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        assert s.step() is None
    total = time.perf_counter() - t0
    print(f"N={N} n={2*N*N} depth={os.environ.get('ESQ_CHAIN_DEPTH', 'default')}: "
          f"{1e6*total/steps:.1f} us/step = enqueue {1e6*acc['stages']/steps:.1f} + "
          f"error-norm wait {1e6*acc['solerr']/steps:.1f} + accept {1e6*acc['finish']/steps:.1f} "
          f"+ python {1e6*(total - sum(acc.values()))/steps:.1f}")


if __name__ == "__main__":
    for N in [int(a) for a in sys.argv[1:]] or [100, 316]:
        probe(N)
