#!/bin/bash
# same-box interleaved A/B of the marching chain sweeps on a bench config:
#   tools/ab_chain.sh <config> <depth> [<depth> ...]     (depth 1 = one sweep per stage)
# extra environment (ESQ_BLOCK_ACC, ESQ_CHAIN_ROWS ...) is passed through
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out
CFG=${1:-pr8}; shift
python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > /dev/null 2>&1
for round in 1 2 3; do
  for D in "$@"; do
    ESQ_CHAIN_DEPTH=$D python3 $ROOT/bench.py --config $CFG --steps 60 --warmup 10 --no-cpu-baseline --no-solve-ivp --no-extras \
      > $OUT/ab_chain_${CFG}_${D}_${round}.json 2>> $OUT/ab_chain.err
    python3 - <<PY
import json
d=json.load(open("$OUT/ab_chain_${CFG}_${D}_${round}.json"))
ks=d["roofline"]["kernels"]
print("$CFG depth=$D round=$round ms/step=%.4f  sum_kernels=%.4f"%(d["ms_per_step"], sum(v["avg_us"]*v["launches"] for v in ks.values())/d["steps"]/1e3))
if $round==1:
    for k,v in sorted(ks.items()): print("    %-18s %3d x %7.1f us  %6.0f GB/s"%(k, v["launches"]//d["steps"], v["avg_us"], v["gbs"] or 0))
PY
  done
done
