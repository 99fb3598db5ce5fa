#!/bin/bash
# round 6: where do chain sweeps start to pay now (diverging pairs, no mask tests)?  Pr8 / Ts5 on
# small grids: the library's rule (no chains below N * tiles-per-row = 2048) against forced tiles
mkdir -p gpurun_out
out=gpurun_out/r06_small_grids.log
: > $out
for N in 96 128 200 256 316 384 448; do
  for r in - 4 5 6 8; do
    if [ "$r" = "-" ]; then unset ESQ_CHAIN_ROWS; else export ESQ_CHAIN_ROWS=$r; fi
    echo -n "rows=$r " >> $out; python tools/kernel_times.py Pr8 bruss $N 300 2>&1 | cut -c1-60 >> $out
  done
done
for N in 128 256 384 512; do
  for r in - 4 5 6; do
    if [ "$r" = "-" ]; then unset ESQ_CHAIN_ROWS; else export ESQ_CHAIN_ROWS=$r; fi
    echo -n "rows=$r " >> $out; python tools/kernel_times.py Ts5 heat $N 300 2>&1 | cut -c1-60 >> $out
  done
done
cat $out
