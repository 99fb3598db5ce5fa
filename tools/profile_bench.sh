#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   tools/profile_bench.sh [config[:plugin][@grid] ...]  (default: pr8; pr8@7070 = n 1e8;
#   pr8:diff3d = Pr8 on the 3-D plugin, pr8:diff3d@400 the same at N = 400)
# per config:
#   1. kernel trace + stats of the driver-style bench command
#   2. HBM traffic counters, one --pmc pass each (FETCH_SIZE, WRITE_SIZE) --
#      never combined with a trace domain other than --kernel-trace
#   3. the bench JSON of an un-profiled run of the same command
# Outputs land in gpurun_out/prof_<config>_*; tools/summarize_profiles.py
# condenses them into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
# a fresh box runs its first seconds of GPU work measurably slower: warm it up
timeout 900 python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > /dev/null 2>&1
# Three passes over the configs, so that no un-profiled number is taken behind a
# counter-collection run (rocprofv3 --pmc leaves the device in its profiling power
# state for a while: the driver's command read 0.533 ms/step behind one, 0.484 on the
# same box otherwise):  1. the bench JSONs (the driver's own command first),
# 2. kernel trace + stats,  3. the PMC passes.
spec() {       # -> CFG GRID TAG STEPS (GRID also carries --plugin)
    local base=${1%@*}
    CFG=${base%%:*}; GRID=""; TAG=$CFG; STEPS=20
    if [ "$base" != "$CFG" ]; then GRID="--plugin ${base#*:}"; TAG=${CFG}_${base#*:}; fi
    if [ "$1" != "$base" ]; then GRID="$GRID --grid ${1#*@}"; TAG=${TAG}_${1#*@}; STEPS=${ESQ_PROF_STEPS:-10}; fi
}
FLAGS="--no-cpu-baseline --no-solve-ivp --no-extras"
for SPEC in "${@:-pr8}"; do
    spec $SPEC
    if [ "$CFG" = driver ]; then
        # the driver's own command line (its flags), extras and CPU baseline included
        timeout 900 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 \
            > $OUT/prof_driver_bench.json 2> $OUT/prof_driver_bench.err
        continue
    fi
    timeout 900 python3 $ROOT/bench.py --config $CFG $GRID --steps $STEPS --warmup 5 $FLAGS \
        > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_bench.err
done
for SPEC in "${@:-pr8}"; do
    spec $SPEC
    if [ "$CFG" = driver ]; then
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_driver_stats -o bench -- \
            python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/prof_driver_stats.log 2>&1
        continue
    fi
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -o bench -- \
        python3 $ROOT/bench.py --config $CFG $GRID --steps $STEPS --warmup 5 $FLAGS \
        > $OUT/prof_${TAG}_stats.log 2>&1
done
for SPEC in "${@:-pr8}"; do
    spec $SPEC
    [ "$CFG" = driver ] && continue
    timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -o bench -- \
        python3 $ROOT/bench.py --config $CFG $GRID --steps 3 --warmup 1 $FLAGS \
        > $OUT/prof_${TAG}_fetch.log 2>&1
    timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -o bench -- \
        python3 $ROOT/bench.py --config $CFG $GRID --steps 3 --warmup 1 $FLAGS \
        > $OUT/prof_${TAG}_write.log 2>&1
done
ls $OUT | grep prof_ | head -40
