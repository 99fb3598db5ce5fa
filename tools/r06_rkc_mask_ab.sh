#!/bin/bash
# round 6: the Dirichlet masks of the 3-D Chebyshev chain sweeps as a factor of the stage's
# hmus instead of a select of the result (prev = the library before the change)
mkdir -p gpurun_out
out=gpurun_out/r06_rkc_mask_ab.log
: > $out
python -m pytest tests/test_gpu_rkc.py -q -x 2>&1 | tail -2 >> $out
for rep in 1 2 3; do
for lib in product prev; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for cfg in "--config rkc" "--config rkc --grid 400 --steps 6"; do
    python bench.py $cfg --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', '[$cfg]', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
cat $out
