#!/bin/bash
# SSV2stab bench config over (depth, JT, NW) shapes of the 3-D chain sweep:
#   tools/rkc_shape_sweep.sh <grid> "<depth>:<JT>,<NW> ..."
GRID=${1:-159}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for spec in $1; do
    d=${spec%%:*}; cfg=${spec#*:}
    ESQ_RKC_MAXDEPTH=$d ESQ_RKC_DEPTH=$d ESQ_RKC_CFG=$cfg python3 $ROOT/bench.py --config rkc --grid $GRID \
        --steps ${ESQ_SWEEP_STEPS:-20} --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras \
        > $ROOT/gpurun_out/rkc_shape.json 2> $ROOT/gpurun_out/rkc_shape.err
    python3 - <<PY
import json
try:
    b = json.loads(open("$ROOT/gpurun_out/rkc_shape.json").read().strip().splitlines()[-1])
    ks = b["roofline"]["kernels"]
    main = max((k for k in ks if k.startswith("rkc_chain") or k == "rhs_rkc"), key=lambda k: ks[k]["launches"])
    print("depth $d cfg $cfg: %.4f ms/step   %s x%d %.1f us (%.1f MB)" % (
        b["ms_per_step"], main, ks[main]["launches"], ks[main]["avg_us"], ks[main]["moved_bytes_per_launch"] / 1e6))
except Exception as exc:
    print("depth $d cfg $cfg failed:", exc, open("$ROOT/gpurun_out/rkc_shape.err").read()[-600:])
PY
done
