#!/bin/bash
# round 6: spread of the Pr8 chain sweeps between processes and between allocations
mkdir -p gpurun_out
out=gpurun_out/r06_variance.log
: > $out
for rep in 1 2 3 4 5 6; do
  ESQ_PLAN_DEBUG=1 python tools/variance_probe.py 1 30 2>&1 | grep "slab at\|instance" >> $out
done
ESQ_PLAN_DEBUG=1 python tools/variance_probe.py 10 30 2>&1 | grep "slab at\|instance" >> $out
cat $out
