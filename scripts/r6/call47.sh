#!/bin/bash
# round 6 call 47: after the reversal -- the in-tree build against the build of commit 4656cdc (same sources: must tie) and of 156e258, kernel tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > $O/call47_tests.log 2>&1; tail -n 2 $O/call47_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$* |" | tee -a $O/call47_step.log; }
for rep in 1 2; do
  run CXR_AB=in-tree
  run CXR_LIB=$R/cxrmate_amd/lib/ab_4656cdc.so
  run CXR_LIB=$R/cxrmate_amd/lib/ab_156e258.so
done
