#!/bin/bash
# round 6: is the 5.5 us between config 2's sweep and its final sum the write-back of the sweep's
# dirty lines?  The sweep's stores streamed (EPI_NT bit 0) against cached
mkdir -p gpurun_out
out=gpurun_out/r06_ts5_nt.log
: > $out
for rep in 1 2; do
for nt in "" 0 1 3 15 31; do
  ESQ_EPI_NT=$nt python bench.py --config ts5 --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('EPI_NT=$nt', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
done; done
cat $out
