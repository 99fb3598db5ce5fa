#!/bin/bash
# three solve_ivp figures in one process with the copy log on; prints the gaps and the copies
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
python3 -c "
import sys; sys.path.insert(0, '$ROOT')
from extensisq_amd import _lib
b = _lib.device_pci_bus_id(0)
print('GPU', b, 'node', open('/sys/bus/pci/devices/%s/numa_node' % b).read().strip())
"
ESQ_SNAPSHOT_DEBUG=1 ESQ_BENCH_DEBUG=1 timeout 300 python3 $ROOT/tools/solve_ivp_probe.py > $OUT/ivp_debug.out 2> $OUT/ivp_debug.err
grep -v "lazy flags" $OUT/ivp_debug.err | grep -v "t_eval" | cut -c1-400 | awk '/solve_ivp gaps/{print; n=0; next} {n++; if (n<=4) print}' | head -120
python3 -c "
import sys; sys.path.insert(0, '$ROOT')
import json
for ln in open('$OUT/ivp_debug.out'):
    try: s = json.loads(ln)
    except Exception: continue
    print('solve_ivp_probe: median %.2f mean %.2f t_eval %.3f' % (s['ms_per_step'], s['ms_per_step_mean'], s['t_eval_end']['ms_per_step']), s.get('download_stream'))
"
