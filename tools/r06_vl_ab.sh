#!/bin/bash
# round 6: do stored row segments of whole 64-byte blocks pay in the 3-D sweeps?  N = 160 (rows
# start on cache lines without any padding): tiles of 54 points (product) against tiles of 56
# (experiments/r06_vl_aligned.patch); N = 159 (rows start anywhere) as the control
mkdir -p gpurun_out
out=gpurun_out/r06_vl_ab.log
: > $out
for rep in 1 2; do
for lib in product vl8; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for cfg in "--config pr8 --plugin diff3d --grid 160" "--config pr8 --plugin diff3d --grid 159" "--config rkc --grid 160" "--config rkc --grid 159"; do
    python bench.py $cfg --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', '$cfg', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
cat $out
