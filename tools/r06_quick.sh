#!/bin/bash
# round 6, first GPU contact: the new whole-step tests, the chain tests that cover
# BS5 / Ts5 / CFMR7osc, then the method sweep with the round-5 sequence beside it
mkdir -p gpurun_out
python -m pytest tests/test_gpu_whole_step.py -x -q 2>&1 | tail -15 > gpurun_out/r06_t1.log
python -m pytest tests/test_gpu_parity.py -x -q -k "chained_stage_sweeps or trajectory_golden or first_launch_ahead or lazy or end_point" 2>&1 | tail -8 > gpurun_out/r06_t2.log
python tools/method_sweep.py 60 > gpurun_out/r06_sweep_new.log 2>&1
ESQ_PRE_WHOLE=0 ESQ_CHAIN_ERRNORM=0 python tools/method_sweep.py 60 > gpurun_out/r06_sweep_old.log 2>&1
python bench.py --config ts5 > gpurun_out/r06_ts5_new.json 2> gpurun_out/r06_ts5_new.err
ESQ_CHAIN_ERRNORM=0 python bench.py --config ts5 > gpurun_out/r06_ts5_old.json 2> gpurun_out/r06_ts5_old.err
cat gpurun_out/r06_t1.log gpurun_out/r06_t2.log gpurun_out/r06_sweep_new.log gpurun_out/r06_sweep_old.log
