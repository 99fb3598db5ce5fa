#!/bin/bash
# PMC groups (memory side, L2, TA, SQ) of the Chebyshev chain sweeps, one rocprofv3 pass each:
#   [GRIDARG="--grid 400"] tools/rkc_pmc_probe.sh       (GRIDARG: extra bench.py arguments,
#   default: the BASELINE grid N = 159); tables per kernel label on stdout
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  k=$((k+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmcx_$k -o p -- \
      python3 $ROOT/bench.py --config rkc $GRIDARG --steps 3 --warmup 1 --no-cpu-baseline --no-solve-ivp --no-extras > $OUT/pmcx_$k.log 2>&1
done
python3 - <<PY
import csv, collections, glob, re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmcx_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"]
        k3=re.search(r"k_rkc3d_chain<(\d+), (\d+), (\d+)", name)
        if not k3: continue
        lab = f"rkc_chain{k3.group(1)}[JT={k3.group(2)},NW={k3.group(3)}]"
        agg[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
for lab in sorted(agg):
    a={c:sum(v)/len(v) for c,v in agg[lab].items()}
    print(lab)
    for c,v in sorted(a.items()):
        print("    %-36s %12.5g"%(c, v))
PY
