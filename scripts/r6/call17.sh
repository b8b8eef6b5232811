#!/bin/bash
# round 6 call 17: grouped q / k / v launches hand their 36928-row member to the row-strip kernel: tests + TF step A/B (CXR_GEMM_STRIP=0 is the round-5 path)
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "gemm or full_size_tf_gradients or full_depth_encoder or tf_single_logits" > gpurun_out/r6/call17_tests.log 2>&1; tail -n 3 gpurun_out/r6/call17_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call17_step.log; }
for rep in 1 2; do
  run CXR_GEMM_STRIP=0
  run CXR_GEMM_STRIP=1
done
