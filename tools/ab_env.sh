#!/bin/bash
# same-box interleaved A/B of one environment switch of the library:
#   tools/ab_env.sh <VAR>=<value-B> <rounds> <config> [config ...]     (A = unset)
KV=$1; ROUNDS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for cfg in "$@"; do
  for r in $(seq 1 $ROUNDS); do
    for side in A B; do
      if [ $side = B ]; then export "$KV"; else unset ${KV%%=*}; fi
      python3 $ROOT/bench.py --config $cfg --steps ${ESQ_AB_STEPS:-60} --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras \
          > $ROOT/gpurun_out/ab_env_${cfg}_${side}_$r.json 2> $ROOT/gpurun_out/ab_env.err
      python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/ab_env_${cfg}_${side}_$r.json').read().strip().splitlines()[-1])
print('$cfg $side round $r: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
    done
  done
done
