#!/usr/bin/env python3
"""Where the host spends a Pr8 step (bench workload): time inside each C call and in
Python between them.  Run on the GPU box:  python tools/host_share.py [steps]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Timed:
    def __init__(self, lib):
        self._lib = lib
        self.t = {}
        self.n = {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not callable(fn):
            return fn

        def wrap(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            dt = time.perf_counter() - t0
            self.t[name] = self.t.get(name, 0.0) + dt
            self.n[name] = self.n.get(name, 0) + 1
            return r
        setattr(self, name, wrap)
        return wrap


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else None
    w = bench.make_workload("pr8", grid, 0)
    s = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=0, **w["kw"])
    for _ in range(20):
        assert s.step() is None
    tl = Timed(s._lib)
    s._lib = tl
    s._dev.lib = tl
    s._dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert s.step() is None
    s._dev.synchronize()
    total = time.perf_counter() - t0
    print("wall %.1f us/step" % (1e6 * total / steps))
    inside = 0.0
    for k in sorted(tl.t, key=lambda k: -tl.t[k]):
        print("  %-28s %7.2f us/step  (%d calls/step)" % (k, 1e6 * tl.t[k] / steps, tl.n[k] / steps))
        inside += tl.t[k]
    print("  python between the calls     %7.2f us/step" % (1e6 * (total - inside) / steps))


if __name__ == "__main__":
    main()
