#!/bin/bash
out=${1:-gpurun_out/sweep_fuse.jsonl}
: > $out
for fz in 1 0 1; do
   echo "# ESQ_FUSE_STAGE=$fz" >> $out
   ESQ_FUSE_STAGE=$fz python bench.py --steps 60 --warmup 5 --no-cpu-baseline >> $out 2>&1
done
