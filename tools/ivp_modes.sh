#!/bin/bash
# tools/ivp_modes.sh [reps]: plain solve_ivp (every state kept) <reps> times in one process,
# per download mode (auto / engine / kernel); prints ms/step and the lane's record
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
REPS=${1:-6}
for MODE in auto engine kernel; do
ESQ_D2H_MODE=$MODE timeout 300 python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
import bench
w = bench.make_workload("pr8", None, 0)
for rep in range($REPS):
    s = bench.solve_ivp_figure(w, 0, 24)
    d = s.get("download_stream") or {}
    print("$MODE rep %d: median %.2f mean %.2f t_eval %.3f | best probe %.1f GB/s, kernel ref %.1f, last %.1f, engine copies %d, kernel copies %d" % (
        rep, s["ms_per_step"], s["ms_per_step_mean"], s["t_eval_end"]["ms_per_step"], d.get("best_probe_gbs", 0), d.get("kernel_ref_gbs", 0), d.get("last_gbs", 0),
        d.get("engine_copies", 0), d.get("kernel_copies", 0)), flush=True)
PY
done
