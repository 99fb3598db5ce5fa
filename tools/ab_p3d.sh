#!/bin/bash
# same-box interleaved A/B of library builds on the 3-D plugin:
#   tools/ab_p3d.sh <variant[,variant...]> <rounds> <config> <grid> [steps]
VARS=$1; ROUNDS=$2; CFG=$3; G=$4; STEPS=${5:-40}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for r in $(seq 1 $ROUNDS); do
  for side in A ${VARS//,/ }; do
    if [ $side = A ]; then unset ESQ_LIB; else export ESQ_LIB=$ROOT/extensisq_amd/libextensisq_amd_$side.so; fi
    python3 $ROOT/bench.py --config $CFG --plugin diff3d --grid $G --steps $STEPS --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras \
        > $ROOT/gpurun_out/ab_p3d.json 2> $ROOT/gpurun_out/ab_p3d.err
    python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/ab_p3d.json').read().strip().splitlines()[-1])
print('$CFG@$G $side round $r: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
  done
done
