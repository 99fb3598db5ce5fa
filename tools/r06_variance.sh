#!/bin/bash
# round 6: spread of the Pr8 chain sweeps between processes and between allocations
mkdir -p gpurun_out
out=gpurun_out/r06_variance.log
: > $out
for rep in 1 2 3 4 5 6; do
  python tools/variance_probe.py 1 30 >> $out 2>&1
done
python tools/variance_probe.py 6 30 >> $out 2>&1
cat $out
