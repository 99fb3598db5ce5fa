#!/bin/bash
# tile height of the marching chain sweeps: per-kernel us of bench config $1 for ESQ_CHAIN_ROWS in $2..
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-pr8}; shift
python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > /dev/null 2>&1
for round in 1 2; do
for R in "$@"; do
  ESQ_CHAIN_ROWS=$R python3 $ROOT/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --no-solve-ivp --no-extras > /tmp/sw.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("/tmp/sw.json"))
ks=d["roofline"]["kernels"]
print("$CFG R=$R ms/step=%.4f sum_kernels=%.4f  "%(d["ms_per_step"], sum(v["avg_us"]*v["launches"] for v in ks.values())/d["steps"]/1e3) + " ".join("%s=%.0f"%(k,v["avg_us"]) for k,v in sorted(ks.items()) if k.startswith("chain")))
PY
done; done
