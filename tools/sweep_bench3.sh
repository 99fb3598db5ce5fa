#!/bin/bash
out=${1:-gpurun_out/sweep_bench3.jsonl}
: > $out
for var in 1 2 3 4 8; do
   echo "# ESQ_RHS_VARIANT=$var" >> $out
   ESQ_RHS_VARIANT=$var python bench.py --steps 30 --warmup 3 --no-cpu-baseline >> $out 2>&1
done
