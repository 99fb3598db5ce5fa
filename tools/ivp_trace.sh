#!/bin/bash
# kernel trace of one plain solve_ivp with the downloads made by the copy kernel: do the
# copy kernel and the step's sweeps overlap?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ESQ_D2H_MODE=${1:-kernel}
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ivp_trace -o t -- python3 $ROOT/tools/ivp_one.py 24 > $OUT/ivp_trace.log 2>&1
tail -1 $OUT/ivp_trace.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/ivp_trace/**/t_kernel_trace.csv", recursive=True) + glob.glob("$OUT/ivp_trace/t_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows = sorted(set(rows))
t0 = rows[0][0]
copies = [r for r in rows if "k_d2h" in r[2]]
print(len(rows), "kernels,", len(copies), "copy kernels")
if len(copies) > 8:
    a, b = copies[6][0], copies[9][1]
    for r in rows:
        if r[1] >= a and r[0] <= b:
            print("%9.3f .. %9.3f ms  %-40s queue %s stream %s" % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, r[2], r[3], r[4]))
PY
