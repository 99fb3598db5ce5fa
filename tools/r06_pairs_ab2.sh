#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r06_pairs_ab2.log
: > $out
run() { for lib in pairs indep; do
    if [ $lib = pairs ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_indep.so; fi
    python tools/kernel_times.py "$@" >> $out 2>&1; done; unset ESQ_LIB; }
for rep in 1 2; do
  run Pr9 heat 2236 40
  run Pr9 bruss 2236 30
done
run Pr9 heat 1000 100
run Pr9 bruss 1000 100
cat $out
