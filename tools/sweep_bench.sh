#!/bin/bash
# full-step sweep of the tuning knobs (run on the GPU box); one JSON line each
out=${1:-gpurun_out/sweep_bench.jsonl}
: > $out
for pol in 0 1 10 11; do
 for bpc in 2 8 32; do
  for rnt in 0 1; do
   echo "# ESQ_STAGE_POLICY=$pol ESQ_BLOCKS_PER_CU=$bpc ESQ_RHS_STORE_NT=$rnt" >> $out
   ESQ_STAGE_POLICY=$pol ESQ_BLOCKS_PER_CU=$bpc ESQ_RHS_STORE_NT=$rnt \
     python bench.py --steps 30 --warmup 3 --no-cpu-baseline >> $out 2>&1
  done
 done
done
