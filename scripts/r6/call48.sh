#!/bin/bash
# round 6 call 48: build against build -- the in-tree library vs a build whose row-strip epilogue requests its operands one tile row ahead (no in-kernel switch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$* |" | tee -a $O/call48_step.log; }
for rep in 1 2 3; do
  run CXR_AB=in-tree
  run CXR_LIB=$R/cxrmate_amd/lib/ab_pf.so
done
