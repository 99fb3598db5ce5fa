#!/bin/bash
# round 6: the chain that ends a Chebyshev step at depth 5 (N >= 178) on eight waves of THREE
# rows (no spills; experiments/r06_rkc_end_3x8.patch, results right) against four rows (product)
mkdir -p gpurun_out
out=gpurun_out/r06_rkc_end_ab.log
: > $out
for rep in 1 2; do
for lib in product end38; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for g in 400 256; do
    python bench.py --config rkc --grid $g --steps 6 --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', 'N=$g', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_end38.so python -m pytest tests/test_gpu_rkc.py -q -x 2>&1 | tail -1 >> $out
cat $out
