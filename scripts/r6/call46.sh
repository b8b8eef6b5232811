#!/bin/bash
# round 6 call 46: same-box alternation of three BUILDS of the library (CXR_LIB): the closing tree, commit 156e258 (before the GELU epilogue in the strip kernels),
# commit 4656cdc (before the column slices and the epilogue prefetch) -- python side = the closing tree in all three
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$* |" | tee -a $O/call46_step.log; }
for rep in 1 2 3; do
  run CXR_AB=closing
  run CXR_LIB=$R/cxrmate_amd/lib/ab_156e258.so
  run CXR_LIB=$R/cxrmate_amd/lib/ab_4656cdc.so
done
