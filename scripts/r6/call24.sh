#!/bin/bash
# round 6 call 24: pipeline depth / step width of the tiled NT kernel in the TF step (tuning aids CXR_GEMM_STAGES / CXR_GEMM_BK), one box
mkdir -p gpurun_out/r6
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call24_step.log; }
for rep in 1 2; do
  run CXR_GEMM_STAGES=2
  run CXR_GEMM_STAGES=3
  run CXR_GEMM_BK=32 CXR_GEMM_STAGES=4
  run CXR_GEMM_BK=32 CXR_GEMM_STAGES=3
done
