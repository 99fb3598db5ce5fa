#!/bin/bash
# tools/hang_hunt.sh <runs> [limit_s]: the driver's bench command <runs> times under the
# watchdog (tools/bench_watchdog.py), in the environment of tools/profile_bench.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
RUNS=${1:-10}; LIMIT=${2:-120}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > /dev/null 2>&1
for r in $(seq 1 $RUNS); do
  t0=$(date +%s.%N)
  timeout $((LIMIT + 60)) python3 $ROOT/tools/bench_watchdog.py $LIMIT --gpus 1 --steps 20 --warmup 5 \
      > $OUT/hunt_$r.json 2> $OUT/hunt_$r.err
  rc=$?
  t1=$(date +%s.%N)
  echo "run $r rc=$rc $(echo "$t1 - $t0" | bc) s, json bytes $(stat -c %s $OUT/hunt_$r.json)"
  if [ $rc -ne 0 ]; then tail -80 $OUT/hunt_$r.err; fi
done
