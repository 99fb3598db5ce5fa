#!/bin/bash
# Where the waves of each sweep spend their cycles (one --pmc group per pass,
# --kernel-trace only): tools/sq_probe.sh <config>[:plugin][@grid].  Output on stdout and in
# gpurun_out/sq_probe_<config>_<k>/
SPEC=${1:-pr8}
CP=${SPEC%@*}
GRID=""
if [ "$SPEC" != "$CP" ]; then GRID="--grid ${SPEC#*@}"; fi
CFG=${CP%:*}
if [ "$CP" != "$CFG" ]; then GRID="$GRID --plugin ${CP#*:}"; fi
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"; do
  k=$((k+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/sq_probe_${CFG}_$k -o p -- \
      python3 $ROOT/bench.py --config $CFG $GRID --steps 3 --warmup 1 --no-cpu-baseline --no-solve-ivp --no-extras > $OUT/sq_probe_${CFG}_$k.log 2>&1
done
python3 - <<PY
import csv, collections, glob, re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq_probe_${CFG}_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"]
        c=re.search(r"k_chain2d<\d+, (?:true|false), (\d+), (\d+), (\d+)", name)
        m=re.search(r"k_(\w+)_sweep<.*Epi(\w+?)<(\d+)", name)
        k3=re.search(r"k_rkc3d_chain<(\d+), (\d+), (\d+)", name)
        c3=re.search(r"k_chain3d<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
        if c3: lab = f"chain3d{c3.group(1)}{'+solerr' if c3.group(5)=='3' else ''}<{c3.group(2)}>"
        elif k3: lab = f"rkc_chain{k3.group(1)}[JT={k3.group(2)},NW={k3.group(3)}]"
        elif c: lab = f"chain{c.group(1)}{'+solerr' if c.group(3)=='3' else ''}<{c.group(2)}>"
        else: lab = f"{m.group(2)}<{m.group(3)}>" + ("/src" if "SrcAxpy" in name else "") if m else re.sub(r"\(.*","",name)[-30:]
        agg[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
for lab in sorted(agg):
    a={c:sum(v)/len(v) for c,v in agg[lab].items()}
    w=a.get("SQ_WAVES",0) or 1
    print(lab, "waves=%d"%w)
    for c,v in sorted(a.items()):
        print("    %-28s %12.4g   per wave %10.4g"%(c, v, v/w))
PY
