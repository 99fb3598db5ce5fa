#!/bin/bash
# one bench config under a list of environment settings, same box:
#   tools/env_sweep.sh <config> "VAR=a" "VAR=b VAR2=c" ...      ("" = defaults)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cfg=$1; shift
i=0
for kv in "$@"; do
  i=$((i+1))
  env $kv python3 $ROOT/bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras \
      > $ROOT/gpurun_out/envsweep_${cfg}_$i.json 2> $ROOT/gpurun_out/envsweep.err
  python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/envsweep_${cfg}_$i.json').read().strip().splitlines()[-1])
print('$cfg [$kv]: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
done
