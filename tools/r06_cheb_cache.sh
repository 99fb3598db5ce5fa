#!/bin/bash
# round 6: SSV2stab with the Chebyshev coefficients' m-dependent part cached (host side)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
out=$OUT/r06_cheb_cache.log
: > $out
python -m pytest tests/test_gpu_rkc.py -q -x 2>&1 | tail -1 >> $out
for rep in 1 2 3; do
  for cfg in "--config rkc" "--config rkc --grid 400 --steps 6"; do
    python bench.py $cfg --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; c=d['config']
print('[$cfg]', '%.4f ms/step'%d['ms_per_step'], 'solve_ivp %.4f'%c['solve_ivp']['ms_per_step'], 't_eval %.4f'%c['solve_ivp']['t_eval_end']['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_ssv_159b -o t -- python3 $ROOT/tools/step_loop.py SSV2stab diff3d 159 20 > $OUT/gaps_ssv_159b.log 2>&1
tail -1 $OUT/gaps_ssv_159b.log >> $out
python3 $ROOT/tools/gap_report.py $OUT/gaps_ssv_159b/t_kernel_trace.csv >> $out
cat $out
