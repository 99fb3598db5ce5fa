#!/bin/bash
# explicit pairs on the 3-D plugin, ms per step by chain depth:
#   tools/plugin3d_bench.sh [grid]          (default N = 159)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
G=${1:-159}
for cfg in pr8 ts5 pr9; do
  for depth in 1 2 3 4; do
    ESQ_CHAIN_DEPTH=$depth python3 $ROOT/bench.py --config $cfg --plugin diff3d --grid $G --steps 40 --warmup 5 \
        --no-cpu-baseline --no-solve-ivp --no-extras > $ROOT/gpurun_out/p3d_${cfg}_${G}_d$depth.json 2> $ROOT/gpurun_out/p3d.err
    python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/p3d_${cfg}_${G}_d$depth.json').read().strip().splitlines()[-1])
print('$cfg N=$G depth $depth: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s x%d %.1f' % (k, v['launches'], v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
  done
done
