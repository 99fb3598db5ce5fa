#!/bin/bash
# A/B of one environment knob on ONE box, interleaved rounds.
#   tools/ab_env.sh <config> <VAR> "<v1> <v2> ..." [rounds] [steps]
CFG=$1; VAR=$2; VALS=$3; ROUNDS=${4:-2}; STEPS=${5:-100}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_${VAR}_$CFG.jsonl
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python3 $ROOT/bench.py --config $CFG --steps $STEPS --warmup 10 \
        --no-cpu-baseline --no-solve-ivp 2>/dev/null | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'v':'$v','ms':d['ms_per_step'],'frac':d['roofline']['frac'],'k':{k:round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()}}))" >> $OUT
  done
done
python3 - <<PY
import json,collections
agg=collections.defaultdict(list)
for l in open("$OUT"):
    d=json.loads(l); agg[d['v']].append(d)
for k,v in agg.items():
    print('$CFG $VAR=%-6s ms/step %s   %s'%(k,' '.join('%.4f'%x['ms'] for x in v), v[-1]['k']))
PY
