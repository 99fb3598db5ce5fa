#!/bin/bash
# tools/numa_session.sh: tools/numa_probe.py, then the driver's bench command; prints the
# probe's table, the affinity record and the solve_ivp figure of the same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 300 python3 $ROOT/tools/numa_probe.py > $OUT/numa_probe.txt 2>&1
cat $OUT/numa_probe.txt | grep -v "^$" | tail -24
timeout 300 python3 $ROOT/tools/solve_ivp_probe.py 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: s = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print('solve_ivp_probe: median %.2f mean %.2f  t_eval %.3f' % (s['ms_per_step'], s['ms_per_step_mean'], s['t_eval_end']['ms_per_step']))
"
for VAR in default; do
  unset ESQ_BENCH_NO_PIN ESQ_WARM_BUFFERS
  if [ $VAR = nopin ]; then export ESQ_BENCH_NO_PIN=1; fi
  if [ $VAR = nowarm ]; then export ESQ_WARM_BUFFERS=0; fi
  timeout 600 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/numa_bench.json 2> $OUT/numa_bench.err
  python3 - <<PY
import json
b=json.loads(open("$OUT/numa_bench.json").read().strip().splitlines()[-1])
s=b["config"]["solve_ivp"]
print("$VAR affinity", b["config"].get("cpu_affinity"))
print("$VAR step %.4f  solve_ivp median %.2f mean %.2f assembly %.0f  t_eval %.3f" % (b["ms_per_step"], s["ms_per_step"], s["ms_per_step_mean"], s["assembly_ms"], s["t_eval_end"]["ms_per_step"]))
PY
done
