#!/bin/bash
# round 6 call 16: row-strip loop with 64-deep steps (half the barriers): bit-identity, alone, in the TF step
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "row_strip" > gpurun_out/r6/call16_tests.log 2>&1; tail -n 3 gpurun_out/r6/call16_tests.log
STRIP_QUICK=1 python scripts/r6/strip_micro.py 2>&1 | grep -v amdgpu > gpurun_out/r6/call16_micro.log; cat gpurun_out/r6/call16_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call16_step.log; }
for rep in 1 2; do
  run CXR_STRIP_STAGES=0
  run CXR_STRIP_STAGES=64
done
