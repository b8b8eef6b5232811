#!/bin/bash
# round 6 call 23: 64-token steps in the block weight-gradient kernels (CXR_TN_BR=32 = rounds 3-5): tests, alone, TF step A/B
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn or linear_bwd or deferred_weight" > gpurun_out/r6/call23_tests.log 2>&1; tail -n 3 gpurun_out/r6/call23_tests.log
python -m pytest tests/test_model_gpu.py -q -x -k "full_size_tf_gradients or tf_single_logits" >> gpurun_out/r6/call23_tests.log 2>&1; tail -n 2 gpurun_out/r6/call23_tests.log
for v in 32 64; do echo "== CXR_TN_BR=$v" >> gpurun_out/r6/call23_micro.log; CXR_TN_BR=$v python scripts/tn_micro.py 2>&1 | grep -E "R= 36928|R=  8192 I= 3072|R= 36864|R=147456 I=  768" >> gpurun_out/r6/call23_micro.log; done
cat gpurun_out/r6/call23_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call23_step.log; }
for rep in 1 2; do
  run CXR_TN_BR=32
  run CXR_TN_BR=64
done
