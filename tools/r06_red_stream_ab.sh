#!/bin/bash
# round 6: the final sum of a step's error norm on a stream of its own, beside the next step's
# first sweep (RED_STREAM=1, the default) against in line with the sweeps (RED_STREAM=0)
mkdir -p gpurun_out
out=gpurun_out/r06_red_stream_ab.log
: > $out
for rep in 1 2 3; do
for rs in 1 0; do
  for cfg in "--config ts5" "" "--config pr9" "--config rkc"; do
    ESQ_RED_STREAM=$rs python bench.py $cfg --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('RED_STREAM=$rs', '[$cfg]', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
for rs in 1 0; do
  for N in 16 128 316 500 1000; do
    ESQ_RED_STREAM=$rs python tools/kernel_times.py Pr8 bruss $N 200 2>&1 | sed "s/^/RED_STREAM=$rs /" | cut -c1-140 >> $out
  done
  ESQ_RED_STREAM=$rs python tools/kernel_times.py BS5 heat 1000 200 2>&1 | sed "s/^/RED_STREAM=$rs /" | cut -c1-160 >> $out
done
cat $out
