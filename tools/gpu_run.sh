#!/bin/bash
# usage (on the GPU box, through gpurun): tools/gpu_run.sh <log name> <pytest args...>
# runs pytest with the arguments given and keeps the log under gpurun_out/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
log="$ROOT/gpurun_out/$1"; shift
cd "$ROOT" && python -m pytest "$@" > "$log" 2>&1
rc=$?
tail -n 15 "$log"
exit $rc
