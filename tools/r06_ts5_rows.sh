#!/bin/bash
# round 6: tile height of Ts5's whole-step chain (config 2, N = 1000): ms/step by ESQ_CHAIN_ROWS,
# twice each, interleaved; "-" = the library's own rule
mkdir -p gpurun_out
out=gpurun_out/r06_ts5_rows.log
: > $out
for rep in 1 2; do
for r in - 3 4 5 6 7 8 10 12; do
  if [ "$r" = "-" ]; then unset ESQ_CHAIN_ROWS; else export ESQ_CHAIN_ROWS=$r; fi
  ms=$(python bench.py --config ts5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))")
  echo "rows=$r through $ms" >> $out
  ms=$(ESQ_CHAIN_ERRNORM=0 python bench.py --config ts5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(d['ms_per_step'], ' '.join('%s=%.1f'%(n,v['avg_us']) for n,v in k.items()))")
  echo "rows=$r pair    $ms" >> $out
done; done
cat $out
