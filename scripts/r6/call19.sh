#!/bin/bash
# round 6 call 19: N = 192 row strips (CvT stage 2): bit-identity, alone, in the TF step (CXR_STRIP_MIN_M192 = rows from which they are used)
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "row_strip" > gpurun_out/r6/call19_tests.log 2>&1; tail -n 3 gpurun_out/r6/call19_tests.log
STRIP_N192=1 python scripts/r6/strip_micro.py 2>&1 | grep -v amdgpu > gpurun_out/r6/call19_micro.log; cat gpurun_out/r6/call19_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call19_step.log; }
for rep in 1 2; do
  run CXR_X=0
  run CXR_STRIP_MIN_M192=100000
  run CXR_STRIP_MIN_M192=100000 CXR_STRIP_MT=16
done
