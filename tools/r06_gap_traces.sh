#!/bin/bash
# kernel-trace timelines of plain step loops: where does the GPU wait between kernels?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # tag method plugin N steps
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_$1 -o t -- \
      python3 $ROOT/tools/step_loop.py $2 $3 $4 $5 > $OUT/gaps_$1.log 2>&1
  echo "== $1: $(tail -1 $OUT/gaps_$1.log)"
  python3 $ROOT/tools/gap_report.py $OUT/gaps_$1/t_kernel_trace.csv
}
run bs5_heat1000 BS5 heat 1000 200
run bs5_bruss BS5 bruss 2236 40
run pr9_heat Pr9 heat 2236 40
run pr8_128 Pr8 bruss 128 300
run ssv_159 SSV2stab diff3d 159 20
run pr8_diff3d Pr8 diff3d 159 60
run cfmr_heat CFMR7osc heat 2236 40
