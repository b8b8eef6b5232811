#!/bin/bash
# round 6 call 3: LayerNorm backward, software-pipelined loop vs the old loop, grid caps; tests of what changed; one bench run with the new SCST keys
mkdir -p gpurun_out/r6
for pf in 0 1; do for grid in 512 1024 2048; do
  echo "== CXR_LN_BWD_PF=$pf CXR_LN_BWD_GRID=$grid" >> gpurun_out/r6/call03_ln.log
  CXR_LN_BWD_PF=$pf CXR_LN_BWD_GRID=$grid python scripts/ln_bwd_micro.py >> gpurun_out/r6/call03_ln.log 2>&1
done; done
python -m pytest tests/test_kernels_gpu.py -q -k "layernorm or layer_norm or ln_" > gpurun_out/r6/call03_tests.log 2>&1
python -m pytest tests/test_reward_scst_gpu.py tests/test_model_gpu.py -q -x -k "reward or scst or tf_single_logits or c5_scst" >> gpurun_out/r6/call03_tests.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/r6/call03_bench.json 2> gpurun_out/r6/call03_bench.err
cat gpurun_out/r6/call03_ln.log; tail -n 4 gpurun_out/r6/call03_tests.log; tail -c 600 gpurun_out/r6/call03_bench.err
