#!/bin/bash
# A/B of one environment knob on ONE box, interleaved rounds.
#   tools/ab_env.sh <config> <VAR> "<v1> <v2> ..." [rounds] [steps]
# prints per value: wall ms/step of every round, and the summed device time of
# all kernels per step (less sensitive to host jitter than the wall clock)
CFG=$1; VAR=$2; VALS=$3; ROUNDS=${4:-2}; STEPS=${5:-100}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_${VAR}_$CFG.jsonl
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python3 $ROOT/bench.py --config $CFG --steps $STEPS --warmup 10 \
        --no-cpu-baseline --no-solve-ivp 2>/dev/null | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print(json.dumps({'v':'$v','ms':d['ms_per_step'],'frac':d['roofline']['frac'],'dev_ms':sum(x['avg_us']*x['launches'] for x in k.values())/d['steps']/1e3,'k':{n:round(x['avg_us'],1) for n,x in k.items()}}))" >> $OUT
  done
done
python3 - <<PY
import json,collections
agg=collections.defaultdict(list)
for l in open("$OUT"):
    d=json.loads(l); agg[d['v']].append(d)
for k,v in agg.items():
    print('$CFG $VAR=%-9s wall %s | kernels %s | mean %.4f'%(k,' '.join('%.4f'%x['ms'] for x in v),' '.join('%.4f'%x['dev_ms'] for x in v), sum(x['dev_ms'] for x in v)/len(v)))
PY
