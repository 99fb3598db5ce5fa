#!/bin/bash
# more kernel-trace timelines: free controller, solve_ivp loops
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # tag method plugin N steps mode
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_$1 -o t -- \
      python3 $ROOT/tools/step_loop.py $2 $3 $4 $5 $6 > $OUT/gaps_$1.log 2>&1
  echo "== $1: $(grep 'ms/step' $OUT/gaps_$1.log | tail -1)"
  python3 $ROOT/tools/gap_report.py $OUT/gaps_$1/t_kernel_trace.csv 0.3
}
run pr8_adaptive Pr8 bruss 2236 30 adaptive
run pr8_ivp Pr8 bruss 2236 24 ivp
run pr8_ivp_dense Pr8 bruss 2236 16 ivp_dense
run bs5_adaptive BS5 heat 2236 40 adaptive
run ts5_adaptive Ts5 heat 1000 200 adaptive
