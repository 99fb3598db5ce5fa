#!/bin/bash
# tools/p3d_quick.sh <grid> <depth> [config ...]: one bench line per config on the 3-D plugin
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
G=$1; D=$2; shift; shift
for cfg in "$@"; do
    ESQ_CHAIN_DEPTH=$D python3 $ROOT/bench.py --config $cfg --plugin diff3d --grid $G --steps 40 --warmup 5 \
        --no-cpu-baseline --no-solve-ivp --no-extras > $ROOT/gpurun_out/p3dq.json 2> $ROOT/gpurun_out/p3dq.err
    python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/p3dq.json').read().strip().splitlines()[-1])
print('$cfg N=$G depth $D: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s x%d %.1f' % (k, v['launches'], v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
done
