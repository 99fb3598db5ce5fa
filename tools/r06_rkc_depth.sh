#!/bin/bash
# round 6: is four stages per Chebyshev chain sweep still the best depth at N = 159 (config 4)?
mkdir -p gpurun_out
out=gpurun_out/r06_rkc_depth.log
: > $out
for rep in 1 2; do
for d in 4 5 6; do
  ESQ_RKC_MAXDEPTH=$d python bench.py --config rkc --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('depth $d', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
done; done
cat $out
