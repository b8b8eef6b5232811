#!/bin/bash
# round 6 call 44: random-shape fuzz of the round-6 GEMM kernels (two seeds) + the round-3 fuzz script
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
for s in 0 1 2; do FUZZ_SEED=$s timeout 600 python scripts/r6/fuzz_r6.py 2>&1 | grep -v amdgpu.ids | tee -a $O/call44_fuzz.log; done
timeout 900 python scripts/fuzz_kernels.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O/call44_fuzz.log
