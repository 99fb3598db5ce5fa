#!/bin/bash
# A/B of the fused epilogues on ONE box (interleaved rounds): which epilogue
# kinds the library may request (ESQ_FUSE) and the accept-time first stage
# (ESQ_PRELAUNCH).  Usage: tools/ab_fuse.sh <config> [rounds]
CFG=${1:-pr8}
ROUNDS=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/ab_fuse_$CFG.jsonl
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "stage:0" "stage,block:0" "stage,solerr,errnorm:0" "stage,block,solerr,errnorm:0" "stage,block,solerr,errnorm:1"; do
    F=${v%%:*}; P=${v##*:}
    ESQ_FUSE=$F ESQ_PRELAUNCH=$P python3 $ROOT/bench.py --config $CFG --steps 100 --warmup 10 \
        --no-cpu-baseline --no-solve-ivp 2>/dev/null | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'fuse':'$F','prelaunch':$P,'ms':d['ms_per_step'],'frac':d['roofline']['frac'],'busy':d['roofline']['device_busy_frac_replay'],'k':{k:round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()}}))" >> $OUT
  done
done
cat $OUT | python3 -c "
import sys,json,collections
agg=collections.defaultdict(list)
for l in sys.stdin:
    d=json.loads(l); agg[(d['fuse'],d['prelaunch'])].append(d['ms'])
for k,v in agg.items(): print('%-32s prelaunch=%d  ms/step %s'%(k[0],k[1],' '.join('%.4f'%x for x in v)))
"
