#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command
#   2. HBM traffic counters, one --pmc pass each (FETCH_SIZE, WRITE_SIZE)
# Outputs land in gpurun_out/prof_*; summaries are copied into profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- \
    python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o bench -- \
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o bench -- \
    python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1
ls -R $OUT/prof_stats | head -20
