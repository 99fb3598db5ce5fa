#!/bin/bash
# same-box interleaved A/B of environment settings on a bench config:
#   tools/ab_env.sh <config> "VAR=a VAR2=b" "VAR=c" ...     ("" = defaults)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-pr8}; shift
python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > /dev/null 2>&1
for round in 1 2 3; do
  for SET in "$@"; do
    env $SET python3 $ROOT/bench.py --config $CFG --steps 60 --warmup 10 --no-cpu-baseline --no-solve-ivp --no-extras > /tmp/ab_env.json 2>/dev/null
    python3 - <<PY
import json
d=json.load(open("/tmp/ab_env.json"))
ks=d["roofline"]["kernels"]
print("$CFG [%s] round=$round ms/step=%.4f sum=%.4f  "%("$SET", d["ms_per_step"], sum(v["avg_us"]*v["launches"] for v in ks.values())/d["steps"]/1e3) + " ".join("%s=%.0f"%(k,v["avg_us"]) for k,v in sorted(ks.items())))
PY
  done
done
