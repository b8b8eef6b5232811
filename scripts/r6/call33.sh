#!/bin/bash
# round 6 call 33: weight-gradient launches per fork event (CXR_WGRAD_BATCH; every fork is a barrier packet on the main queue, ~6 us of idle: 71 per step at 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a $O/call33_step.log; }
for rep in 1 2; do
  run CXR_WGRAD_BATCH=4
  run CXR_WGRAD_BATCH=6
  run CXR_WGRAD_BATCH=8
  run CXR_WGRAD_BATCH=12
  run CXR_WGRAD_BATCH=3
done
run CXR_WGRAD_BATCH=4
