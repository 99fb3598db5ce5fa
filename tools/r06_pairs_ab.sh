#!/bin/bash
# round 6: diverging tile pairs (esq_chain.hpp) against the independent tiles of round 5
# (libextensisq_amd_indep.so = the commit before), interleaved, per-kernel times
mkdir -p gpurun_out
out=gpurun_out/r06_pairs_ab.log
: > $out
run() { for lib in pairs indep; do
    if [ $lib = pairs ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_indep.so; fi
    python tools/kernel_times.py "$@" >> $out 2>&1; done; unset ESQ_LIB; }
for rep in 1 2; do
  run Pr8 bruss 2236 40
  run Ts5 heat 1000 200
  run Pr9 heat 2236 40
done
run Pr8 bruss 1000 100
run Pr8 bruss 500 200
run Pr8 heat 1000 100
run Ts5 bruss 2236 40
run BS5 bruss 2236 40
run BS5 heat 2236 40
run CFMR7osc bruss 2236 40
run Pr7 bruss 2236 40
run Ts5 heat 2236 60
run Pr9 bruss 2236 30
cat $out
