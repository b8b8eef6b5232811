#!/bin/bash
# round 6 call 18: 384 x 192 blocks also for the 384 x 384 x 36928 gradients (CXR_TN5_MIN 2048: 2 blocks x 1154 steps) but not for the 9280-row ones
mkdir -p gpurun_out/r6
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call18_step.log; }
for rep in 1 2; do
  run CXR_TN5_MIN=4096
  run CXR_TN5_MIN=2048
  run CXR_TN5_MIN=2048 CXR_TN5_WGS=48
done
