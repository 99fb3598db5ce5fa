"""config.solve_ivp of the bench line by itself: plain solve_ivp (every state kept)
and solve_ivp(t_eval=[t_end]) on the metric workload; ESQ_LAZY_Y=0 for the A side.
    python tools/solve_ivp_probe.py [steps]"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
w = bench.make_workload("pr8", None, 0)
for rep in range(3):
    print(json.dumps(bench.solve_ivp_figure(w, 0, steps)))
