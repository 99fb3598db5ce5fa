#!/bin/bash
out=${1:-gpurun_out/sweep_bench2.jsonl}
: > $out
for var in 1 2 4 8; do
 for bpc in 1 2 3 4; do
   echo "# ESQ_RHS_VARIANT=$var ESQ_BLOCKS_PER_CU=$bpc" >> $out
   ESQ_RHS_VARIANT=$var ESQ_BLOCKS_PER_CU=$bpc python bench.py --steps 30 --warmup 3 --no-cpu-baseline >> $out 2>&1
 done
done
for pol in 10 20; do
   echo "# ESQ_STAGE_POLICY=$pol (variant 4, bpc 2)" >> $out
   ESQ_STAGE_POLICY=$pol python bench.py --steps 30 --warmup 3 --no-cpu-baseline >> $out 2>&1
done
