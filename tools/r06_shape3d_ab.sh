#!/bin/bash
# round 6: shape of the 3-D chain sweeps of the explicit pairs (rows per thread x waves per
# workgroup): 2 x 16 (round 5; the 1024-thread workgroup caps the kernel at 128 VGPRs: the wide
# kernels spill) against 4 x 8 and 3 x 8 (variant libraries)
mkdir -p gpurun_out
out=gpurun_out/r06_shape3d_ab.log
: > $out
for rep in 1 2; do
for lib in product s48 s38; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for cfg in "--config pr8 --plugin diff3d" "--config pr8 --plugin diff3d --grid 400 --steps 10"; do
    python bench.py $cfg --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', '$cfg', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
cat $out
