#!/bin/bash
mkdir -p gpurun_out/r6
python scripts/r6/c5_wordemb_diag.py > gpurun_out/r6/call02_diag.log 2>&1
python -m pytest tests/test_reward_scst_gpu.py -x -q -k "scst_step_matches_oracle" > gpurun_out/r6/call02_temp.log 2>&1
tail -n 30 gpurun_out/r6/call02_diag.log; tail -n 5 gpurun_out/r6/call02_temp.log
