#!/bin/bash
# six solve_ivp figures in one process; slow copies probe the streams (ESQ_SNAPSHOT_DEBUG=2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
ESQ_SNAPSHOT_DEBUG=2 timeout 300 python3 - > $OUT/ivp_debug2.out 2> $OUT/ivp_debug2.err <<PY
import sys, json
sys.path.insert(0, "$ROOT")
import bench
w = bench.make_workload("pr8", None, 0)
for rep in range(6):
    s = bench.solve_ivp_figure(w, 0, 24)
    print("rep %d: median %.2f mean %.2f" % (rep, s["ms_per_step"], s["ms_per_step_mean"]), s.get("download_stream"), flush=True)
    sys.stderr.write("== end of rep %d\n" % rep); sys.stderr.flush()
PY
cat $OUT/ivp_debug2.out
grep -c "copy 1\.[34]" $OUT/ivp_debug2.err
grep "SLOW\|== end" $OUT/ivp_debug2.err | cut -c1-330 | head -24
