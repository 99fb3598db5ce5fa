#!/bin/bash
# round 6: SSV2stab's opening sweep of the next step launched ahead (esq_rkc_guess_next)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
out=$OUT/r06_rkc_ahead.log
: > $out
python -m pytest tests/test_gpu_rkc.py -q -x 2>&1 | tail -3 >> $out
for rep in 1 2 3; do
for la in 1 0; do
  ESQ_LAUNCH_AHEAD=$la python bench.py --config rkc --no-cpu-baseline --no-extras --no-solve-ivp 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('launch_ahead=$la %.4f ms/step'%d['ms_per_step'])" >> $out
  ESQ_LAUNCH_AHEAD=$la python tools/step_loop.py SSV2stab diff3d 159 40 >> $out 2>&1
done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_ssv_159c -o t -- python3 $ROOT/tools/step_loop.py SSV2stab diff3d 159 20 > $OUT/gaps_ssv_159c.log 2>&1
tail -1 $OUT/gaps_ssv_159c.log >> $out
python3 $ROOT/tools/gap_report.py $OUT/gaps_ssv_159c/t_kernel_trace.csv >> $out
cat $out
