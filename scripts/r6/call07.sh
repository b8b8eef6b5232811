#!/bin/bash
# round 6 call 7: workgroup targets of the 384 x 192-block launches (CXR_TN5_WGS) and of the 256-block launches (CXR_TN2_WGS) in the TF step, one box
mkdir -p gpurun_out/r6
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call07_step.log; }
for rep in 1 2; do
  run CXR_TN5=0
  run CXR_TN5=0 CXR_TN2_WGS=64
  run CXR_TN5=1 CXR_TN5_WGS=32
  run CXR_TN5=1 CXR_TN5_WGS=48
  run CXR_TN5=1 CXR_TN5_WGS=64
  run CXR_TN5=1 CXR_TN5_WGS=80
  run CXR_TN5=1 CXR_TN5_WGS=64 CXR_TN2_WGS=64
  run CXR_TN5=1 CXR_TN5_WGS=48 CXR_TN2_WGS=64 CXR_TN_WGS=128
done
