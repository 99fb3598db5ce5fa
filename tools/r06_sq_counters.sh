#!/bin/bash
# round 6: where do the waves of the chain sweeps spend their cycles?  SQ counters (one --pmc
# pass, kernel trace only): parked at s_waitcnt / barrier, issue-stalled, issuing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/r06_counters_list.txt 2>&1
FLAGS="--no-cpu-baseline --no-solve-ivp --no-extras"
for cfg in ts5 pr8 rkc; do
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
      --kernel-trace --output-format csv -d $OUT/prof_${cfg}_sq -o bench -- \
      python3 $ROOT/bench.py --config $cfg --steps 3 --warmup 1 $FLAGS > $OUT/prof_${cfg}_sq.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES \
      --kernel-trace --output-format csv -d $OUT/prof_${cfg}_sq2 -o bench -- \
      python3 $ROOT/bench.py --config $cfg --steps 3 --warmup 1 $FLAGS > $OUT/prof_${cfg}_sq2.log 2>&1
done
ls $OUT | grep "_sq"
