#!/bin/bash
# round 6: tile heights under diverging pairs (ESQ_CHAIN_ROWS; "-" = the library's rule)
mkdir -p gpurun_out
out=gpurun_out/r06_rows_sweep.log
: > $out
sweep() { name=$1; plug=$2; N=$3; steps=$4; shift 4
  for r in "$@"; do
    if [ "$r" = "-" ]; then unset ESQ_CHAIN_ROWS; else export ESQ_CHAIN_ROWS=$r; fi
    echo -n "rows=$r " >> $out; python tools/kernel_times.py $name $plug $N $steps >> $out 2>&1
  done; unset ESQ_CHAIN_ROWS; }
sweep Ts5 heat 1000 200 - 3 4 5 6 7 8 10
sweep Pr8 bruss 1000 100 - 6 7 8 9 10 12 14 18
sweep Pr8 bruss 2236 40 - 22 30 36 40 44 48 56
sweep Pr9 heat 2236 40 - 15 18 22 26 30 44
sweep Pr8 bruss 500 200 - 4 5 6 7 8 10
cat $out
