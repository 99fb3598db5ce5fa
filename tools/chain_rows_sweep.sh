#!/bin/bash
# tile height of the chain sweeps (ESQ_CHAIN_ROWS) on one bench config, same box:
#   tools/chain_rows_sweep.sh <config> <rows> [rows ...]      (0 = the planner's choice)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cfg=$1; shift
for r in "$@"; do
  if [ "$r" = 0 ]; then unset ESQ_CHAIN_ROWS; else export ESQ_CHAIN_ROWS=$r; fi
  python3 $ROOT/bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras \
      > $ROOT/gpurun_out/rows_${cfg}_$r.json 2> $ROOT/gpurun_out/rows.err
  python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/rows_${cfg}_$r.json').read().strip().splitlines()[-1])
print('$cfg rows $r: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
done
