#!/bin/bash
# round 6: Pr9 on the Brusselator read 1.024 ms/step in one session and 1.132 in a later one:
# per-kernel times of HEAD against the library of commit 4d2de7e (before the small-grid rules)
mkdir -p gpurun_out
out=gpurun_out/r06_pr9_bruss_ab.log
: > $out
for rep in 1 2; do
for lib in product presmall; do
  if [ $lib = product ]; then unset ESQ_LIB; else export ESQ_LIB=$PWD/extensisq_amd/libextensisq_amd_$lib.so; fi
  python tools/kernel_times.py Pr9 bruss 2236 30 >> $out 2>&1
  python tools/kernel_times.py Pr8 bruss 2236 30 >> $out 2>&1
done; done
cat $out
