#!/bin/bash
# round 6, VERDICT r05 item 3a: what would it buy if the halo ROWS of a chain tile cost
# nothing?  Upper bounds from two what-if builds (wrong results; experiments/):
#   norowhalo  rows outside the tile are not LOADED (an LDS hand-over of the shared rows
#              between vertical neighbours could save at most this)
#   norunin    ... and not even walked: no run-in, no run-out iterations (what a fully
#              cooperative exchange of boundary VALUES between vertical neighbours could
#              save at most)
mkdir -p gpurun_out
out=gpurun_out/r06_whatif_row_halo.log
: > $out
for rep in 1 2 3; do
  for lib in product norowhalo norunin; do
    if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
    python tools/kernel_times.py Pr8 bruss 2236 40 >> $out 2>&1
  done
done
for lib in product norowhalo norunin; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  python tools/kernel_times.py Pr9 heat 2236 40 >> $out 2>&1
  python tools/kernel_times.py Ts5 heat 1000 200 >> $out 2>&1
done
cat $out
