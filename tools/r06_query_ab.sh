#!/bin/bash
# round 6: the stream query of the host's wait for a reduction only after 2 ms (product) against
# every 4096 spins (prev = the library before the change)
mkdir -p gpurun_out
out=gpurun_out/r06_query_ab.log
: > $out
for rep in 1 2 3; do
for lib in product prev; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for cfg in "--config ts5" "" "--config pr9" "--config rkc"; do
    python bench.py $cfg --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', '[$cfg]', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
for lib in product prev; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for N in 16 128 316 500 1000; do
    python tools/kernel_times.py Pr8 bruss $N 200 2>&1 | cut -c1-140 >> $out
  done
  python tools/kernel_times.py BS5 heat 1000 200 2>&1 | cut -c1-160 >> $out
  python tools/kernel_times.py BS5 bruss 2236 40 2>&1 | cut -c1-160 >> $out
done
cat $out
