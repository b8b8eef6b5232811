#!/bin/bash
# round 6 call 37: workgroup targets of the 128-tile / 256-block weight-gradient kernels, re-swept now that the main stream runs one-workgroup-per-CU
# row-strip kernels (a 64-KB weight-gradient workgroup on a CU keeps a 140-KB strip workgroup off it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a $O/call37_step.log; }
for rep in 1 2; do
  run CXR_TN_WGS=176
  run CXR_TN_WGS=64
  run CXR_TN_WGS=96
  run CXR_TN_WGS=128
  run CXR_TN_WGS=224
  run CXR_TN2_WGS=64
  run CXR_TN2_WGS=80
  run CXR_TN2_WGS=128
done
run CXR_TN_WGS=176
