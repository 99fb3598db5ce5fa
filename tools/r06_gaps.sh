#!/bin/bash
# kernel-trace timeline of a few steps: which kernel starts how long after its predecessor ended
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in ts5 pr8; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_${cfg}_gaps -o bench -- \
      python3 $ROOT/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > $OUT/prof_${cfg}_gaps.log 2>&1
done
python3 - <<PY
import csv, re
for cfg in ['ts5','pr8']:
    rows=list(csv.DictReader(open('$OUT/prof_%s_gaps/bench_kernel_trace.csv'%cfg)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    ks=[(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    mid=len(ks)*2//3
    print(cfg)
    for i in range(mid, mid+10):
        n,s,e=ks[i]; pn,ps,pe=ks[i-1]
        m=re.search(r'k_chain2d<(\d), \w+, (\d), (\d), (\d)|k_\w+', n)
        print("  %-36s dur %8.2f us  gap after prev %7.2f us" % (m.group(0)[:34], (e-s)/1000, (s-pe)/1000))
PY
