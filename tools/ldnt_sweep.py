#!/usr/bin/env python3
"""Cache policy of the chain sweeps' loads (ESQ_CHAIN_LDNT=first,middle,last; bit 0 the
input, 1 y, 2 the K rows) on the Pr8 bench workload: ms/step for every middle x last
combination.  Run on the GPU box:  python tools/ldnt_sweep.py [steps]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(mask, steps):
    os.environ["ESQ_CHAIN_LDNT"] = mask
    w = bench.make_workload("pr8", None, 0)
    s = w["cls"](w["rhs"], 0.0, w["y0"], 1.0e9, device=0, **w["kw"])
    for _ in range(8):
        assert s.step() is None
    s._dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert s.step() is None
    s._dev.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = sys.argv[2] if len(sys.argv) > 2 else "4"
    run("4,4,4", 60)                                    # warm the box
    print("first=%s   last: " % first + " ".join("%6d" % l for l in range(8)))
    for m in range(8):
        row = [run("%s,%d,%d" % (first, m, l), steps) for l in range(8)]
        print("middle=%d        " % m + " ".join("%6.4f" % v for v in row), flush=True)


if __name__ == "__main__":
    main()
