#!/bin/bash
# Diagnostic PMC passes on the sweeps of one bench config (one --pmc group per
# pass, --kernel-trace only).  Output: gpurun_out/pmc_probe_<config>_<k>/
CFG=${1:-pr8}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  k=$((k+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_probe_${CFG}_$k -o p -- \
      python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-solve-ivp --no-extras > $OUT/pmc_probe_${CFG}_$k.log 2>&1
done
python3 - <<PY
import csv, collections, glob, re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_probe_${CFG}_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"]
        c=re.search(r"k_chain2d<\d+, (?:true|false), (\d+), (\d+), (\d+)", name)
        m=re.search(r"k_(\w+)_sweep<.*Epi(\w+?)<(\d+)", name)
        if c: lab = f"chain{c.group(1)}{'+solerr' if c.group(3)=='3' else ''}<{c.group(2)}>"
        else: lab = f"{m.group(2)}<{m.group(3)}>" + ("/src" if "SrcAxpy" in name else "") if m else re.sub(r"\(.*","",name)[-30:]
        agg[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
for lab in sorted(agg):
    print(lab)
    for c,v in sorted(agg[lab].items()):
        print("    %-44s %.4g"%(c, sum(v)/len(v)))
PY
