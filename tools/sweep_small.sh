#!/bin/bash
out=${1:-gpurun_out/sweep_small.jsonl}
: > $out
for pol in 0 10 1; do
 for bpc in 2 4 8 16; do
   echo "# ts5 ESQ_STAGE_POLICY=$pol ESQ_BLOCKS_PER_CU=$bpc" >> $out
   ESQ_STAGE_POLICY=$pol ESQ_BLOCKS_PER_CU=$bpc python bench.py --config ts5 --steps 200 --warmup 10 --no-cpu-baseline >> $out 2>&1
 done
done
for bpc in 2 4 8; do
   echo "# rkc ESQ_BLOCKS_PER_CU=$bpc" >> $out
   ESQ_BLOCKS_PER_CU=$bpc python bench.py --config rkc --steps 10 --warmup 2 --no-cpu-baseline >> $out 2>&1
done
echo "# pr8 default (sampled events)" >> $out
python bench.py --steps 60 --no-cpu-baseline >> $out 2>&1
echo "# pr8 rccl-1" >> $out
python bench.py --steps 60 --no-cpu-baseline --force-lockstep >> $out 2>/dev/null
