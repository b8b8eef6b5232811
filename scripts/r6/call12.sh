#!/bin/bash
# round 6 call 12: row-strip GEMM -- bit-identity tests, alone (all strip heights / stage counts), in the TF step and in the forward
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "row_strip" > gpurun_out/r6/call12_tests.log 2>&1; tail -n 3 gpurun_out/r6/call12_tests.log
python scripts/r6/strip_micro.py > gpurun_out/r6/call12_micro.log 2>&1; grep -v amdgpu gpurun_out/r6/call12_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call12_step.log; }
for rep in 1 2; do
  run CXR_GEMM_STRIP=0
  run CXR_GEMM_STRIP=1
  run CXR_GEMM_STRIP=1 CXR_STRIP_STAGES=2
done
