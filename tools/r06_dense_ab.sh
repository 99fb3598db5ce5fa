#!/bin/bash
# round 6: participation masks (one s_bitcmp + a real branch per FMA pair) against weights +0.0
# (libextensisq_amd_dense.so, experiments/r06_dense_weights.patch), interleaved
mkdir -p gpurun_out
out=gpurun_out/r06_dense_ab.log
: > $out
run() { for lib in product dense; do
    if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
    python tools/kernel_times.py "$@" >> $out 2>&1; done; unset ESQ_LIB; }
for rep in 1 2; do
  run Pr8 bruss 2236 40
  run Ts5 heat 1000 200
done
run Pr9 heat 2236 40
run Pr9 bruss 2236 30
run BS5 bruss 2236 40
run CFMR7osc bruss 2236 40
run Pr7 bruss 2236 40
run Ts5 bruss 2236 40
run Pr8 bruss 1000 100
run Pr8 heat 2236 40
cat $out
