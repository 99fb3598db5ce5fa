#!/bin/bash
# SSV2stab bench config (3-D diffusion) at every chain depth, same box, interleaved:
#   tools/rkc_depth_sweep.sh [grid] [rounds] [extra bench flags...]
# prints ms/step and the per-kernel table of each run (gpurun_out/rkc_d<depth>_<round>.json)
GRID=${1:-159}; ROUNDS=${2:-2}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for r in $(seq 1 $ROUNDS); do
  for d in ${ESQ_SWEEP_DEPTHS:-1 2 3 4}; do
    ESQ_RKC_DEPTH=$d python3 $ROOT/bench.py --config rkc --grid $GRID --steps ${ESQ_SWEEP_STEPS:-20} --warmup 5 \
        --no-cpu-baseline --no-solve-ivp --no-extras "$@" > $ROOT/gpurun_out/rkc_d${d}_$r.json 2> $ROOT/gpurun_out/rkc_d${d}_$r.err
    python3 - <<PY
import json
try:
    b = json.loads(open("$ROOT/gpurun_out/rkc_d${d}_$r.json").read().strip().splitlines()[-1])
    print("depth $d round $r: %.4f ms/step" % b["ms_per_step"])
    for k, v in b["roofline"]["kernels"].items():
        print("    %-18s x%-5d %8.1f us  %6.0f GB/s designed  %7.1f MB" % (
            k, v["launches"], v["avg_us"], v["gbs"] or 0, v["moved_bytes_per_launch"] / 1e6))
except Exception as exc:
    print("depth $d round $r failed:", exc)
    print(open("$ROOT/gpurun_out/rkc_d${d}_$r.err").read()[-2000:])
PY
  done
done
