#!/bin/bash
# round 6: what would a predicate-free steady-state marching loop save?  Upper bound: the
# steady-state predicates in EVERY iteration (experiments/r06_steady_everywhere.patch; results WRONG)
mkdir -p gpurun_out
out=gpurun_out/r06_steady_ab.log
: > $out
for rep in 1 2; do
for lib in product steady; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  for cfg in "" "--config pr9" "--config ts5"; do
    python bench.py $cfg --no-cpu-baseline --no-solve-ivp --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$lib', '[$cfg]', '%.4f ms/step'%d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))" >> $out
  done
done; done
cat $out
