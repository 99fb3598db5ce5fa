#!/bin/bash
# A/B sweep of the library's environment knobs on the full benchmark step (run on
# the GPU box from the repo root; one JSON line per setting, '#'-lines label them).
#   tools/sweep_bench.sh [out.jsonl] [config]
out=${1:-gpurun_out/sweep_bench.jsonl}
cfg=${2:-pr8}
: > $out
run() { echo "# $*" >> $out; env "$@" python bench.py --config $cfg --steps 40 --warmup 3 --no-cpu-baseline >> $out 2>&1; }
run ESQ_NONE=1
run ESQ_BLOCK_ACC=0
run ESQ_CHAIN=0
run ESQ_BLOCK_ACC=0 ESQ_CHAIN=0
run ESQ_BLOCK_FOLD=0
for pol in 0 1 10 11 20; do run ESQ_STAGE_POLICY=$pol; done
for bpc in 1 2 4 8; do run ESQ_BLOCKS_PER_CU=$bpc; done
for bpc in 4 8 16 32; do run ESQ_BLOCK_BPC=$bpc; done
for var in 1 2 4; do run ESQ_RHS_VARIANT=$var; done
