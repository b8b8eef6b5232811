#!/bin/bash
# round 6 call 43: GELU (+ saved pre-activation) epilogue in the row-strip kernels: bit-identity tests, FFN-up forward products alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -q -x -k "column_slices or row_strip or gemm_nt_group" > $O/call43_tests.log 2>&1; tail -n 2 $O/call43_tests.log
timeout 300 python scripts/r6/strip_gelu_micro.py 2>&1 | grep -v amdgpu.ids | tee $O/call43_micro.log
