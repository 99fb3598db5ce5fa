ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for c in 1 0; do for cfg in pr8 ts5; do ESQ_CHAIN=$c python3 bench.py --config $cfg --plugin diff3d --steps 40 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > gpurun_out/d3_${cfg}_$c.json 2>gpurun_out/d3.err; python3 -c "
import json
b=json.loads(open('gpurun_out/d3_${cfg}_$c.json').read().strip().splitlines()[-1])
print('ESQ_CHAIN=$c $cfg: %.4f ms/step' % b['ms_per_step'])
for k,v in b['roofline']['kernels'].items(): print('   %-20s x%-4d %7.1f us %6.0f GB/s' % (k, v['launches'], v['avg_us'], v['gbs'] or 0))"; done; done
