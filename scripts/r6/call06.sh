#!/bin/bash
# round 6 call 6: 384 x 192 blocks in the step -- stage count and workgroup targets (one box, alternating)
mkdir -p gpurun_out/r6
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call06_step.log; }
for rep in 1 2; do
  run CXR_TN5=0
  run CXR_TN5=1
  run CXR_TN5=1 CXR_TN5_STAGES=2
  run CXR_TN5=1 CXR_TN2_WGS=64
  run CXR_TN5=1 CXR_TN2_WGS=128
  run CXR_TN5=0 CXR_TN2_WGS=128
done
