#!/bin/bash
# round 6: does the pitch of the slab's vectors decide where the 158 / 175 us of `chain4<4>`
# fall?  Six allocations per pitch (tools/variance_probe.py), ESQ_ROW_STRIDE = align,offset
mkdir -p gpurun_out
out=gpurun_out/r06_stride.log
: > $out
for rs in "" "65536" "2097152" "2097152,4096" "2097152,65536" "2097152,262144" "2097152,1048576" "1073741824"; do
  echo "== ROW_STRIDE=$rs" >> $out
  ESQ_ROW_STRIDE=$rs ESQ_PLAN_DEBUG=1 python tools/variance_probe.py 6 20 2>&1 | grep "slab at\|instance\|Error\|error" >> $out
done
cat $out
