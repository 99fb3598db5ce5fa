#!/usr/bin/env python3
"""bench.py under a watchdog: python tools/bench_watchdog.py <seconds> [bench.py args ...]
dumps the Python stack of every thread to stderr and exits if the run takes longer
than <seconds> (a hung run on the GPU box otherwise costs the whole gpurun limit)."""
import faulthandler
import os
import runpy
import sys

limit = float(sys.argv[1])
faulthandler.enable()
faulthandler.dump_traceback_later(limit, exit=True)
root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
sys.path.insert(0, root)
runpy.run_path(sys.argv[0], run_name="__main__")
