#!/bin/bash
out=${1:-gpurun_out/sweep_bench3.jsonl}
: > $out
for var in 11 2 11 2; do
   echo "# ESQ_RHS_VARIANT=$var" >> $out
   ESQ_RHS_VARIANT=$var python bench.py --steps 40 --warmup 3 --no-cpu-baseline >> $out 2>&1
done
