#!/usr/bin/env python3
"""one plain solve_ivp (every state kept) on the metric workload: python tools/ivp_one.py [steps]"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

w = bench.make_workload("pr8", None, 0)
s = bench._solve_ivp_run(w, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 24, {})
print("median %.2f mean %.2f ms/step" % (s["ms_per_step"], s["ms_per_step_mean"]))
