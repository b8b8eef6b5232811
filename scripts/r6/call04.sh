#!/bin/bash
# round 6 call 4: cross-attention K/V from the Infinity Cache (prefetch hint on a side stream) -- micro experiment; the reward test that changed
mkdir -p gpurun_out/r6
python scripts/r6/cross_prefetch_micro.py > gpurun_out/r6/call04_prefetch.log 2>&1
python -m pytest tests/test_reward_scst_gpu.py -q -x > gpurun_out/r6/call04_tests.log 2>&1
cat gpurun_out/r6/call04_prefetch.log; tail -n 3 gpurun_out/r6/call04_tests.log
