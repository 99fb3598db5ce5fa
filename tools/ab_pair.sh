#!/bin/bash
# same-box interleaved A/B of the two-stage marching sweeps (ESQ_PAIR) on the
# bench configs:  tools/ab_pair.sh [config ...]   (default pr8)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-solve-ivp > /dev/null 2>&1
for CFG in "${@:-pr8}"; do
  for round in 1 2 3; do
    for P in 0 1; do
      ESQ_PAIR=$P python3 $ROOT/bench.py --config $CFG --steps 60 --warmup 10 --no-cpu-baseline --no-solve-ivp \
        > $OUT/ab_pair_${CFG}_${P}_${round}.json 2>> $OUT/ab_pair.err
      python3 - <<PY
import json
d=json.load(open("$OUT/ab_pair_${CFG}_${P}_${round}.json"))
ks=d["roofline"]["kernels"]
print("$CFG pair=$P round=$round ms/step=%.4f  sum_kernels=%.4f"%(d["ms_per_step"], sum(v["avg_us"]*v["launches"] for v in ks.values())/d["steps"]/1e3))
if $round==1:
    for k,v in sorted(ks.items()): print("    %-18s %3d x %7.1f us  %6.0f GB/s"%(k, v["launches"]//d["steps"], v["avg_us"], v["gbs"] or 0))
PY
    done
  done
done
