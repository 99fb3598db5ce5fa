ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
for side in A noldsw; do
  for d in 4 5 6; do
    if [ $side = A ]; then unset ESQ_LIB; else export ESQ_LIB=$ROOT/extensisq_amd/libextensisq_amd_$side.so; fi
    ESQ_CHAIN_DEPTH=$d python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-solve-ivp --no-extras > $ROOT/gpurun_out/abd.json 2> $ROOT/gpurun_out/abd.err
    python3 -c "
import json
b=json.loads(open('$ROOT/gpurun_out/abd.json').read().strip().splitlines()[-1])
print('$side depth $d round $r: %.4f ms/step  ' % b['ms_per_step'] + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in b['roofline']['kernels'].items()))"
  done
done
done
