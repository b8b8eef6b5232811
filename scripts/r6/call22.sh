#!/bin/bash
# round 6 call 22: grouped row-strip launch for the q / k / v groups: bit-identity + TF step A/B (CXR_STRIP_GROUP=0 = grouped tiled kernel)
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_nt_group or row_strip" > gpurun_out/r6/call22_tests.log 2>&1; tail -n 3 gpurun_out/r6/call22_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call22_step.log; }
for rep in 1 2; do
  run CXR_STRIP_GROUP=0
  run CXR_STRIP_GROUP=1
done
