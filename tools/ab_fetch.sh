#!/bin/bash
# HBM-side read bytes per kernel (FETCH_SIZE, one --pmc pass) for environment settings:
#   tools/ab_fetch.sh <config> "VAR=a" "VAR=b" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out
CFG=${1:-pr8}; shift
cd /tmp && export TMPDIR=/tmp
k=0
for SET in "$@"; do
  k=$((k+1))
  for v in $SET; do export $v; done
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ab_fetch_$k -o p -- \
      python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-solve-ivp --no-extras > $OUT/ab_fetch_$k.log 2>&1
  for v in $SET; do unset ${v%%=*}; done
  python3 - <<PY
import csv, collections, re
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/ab_fetch_$k/p_counter_collection.csv")):
    c=re.search(r"k_chain2d<\d+, (?:true|false), (\d+), (\d+), (\d+)", r["Kernel_Name"])
    if c: agg[f"chain{c.group(1)}{'+solerr' if c.group(3)=='3' else ''}<{c.group(2)}>"].append(float(r["Counter_Value"]))
print("[$SET]", "  ".join("%s fetch %.1f MB" % (k2, 2*1024*sum(v)/len(v)/1e6) for k2, v in sorted(agg.items())))
PY
done
