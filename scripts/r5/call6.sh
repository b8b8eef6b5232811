#!/bin/bash
# GPU call 6 of round 5: the co-resident 4-wave weight-gradient kernel (gemm_tn4_kernel, CXR_TN4=1): parity, alone, and in the step (same-box alternation)
mkdir -p gpurun_out/r5
for st in 3 4; do
  CXR_TN4=1 CXR_TN4_STAGES=$st timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn or deferred_weight" > gpurun_out/r5/tn4_tests_st$st.log 2>&1; tail -3 gpurun_out/r5/tn4_tests_st$st.log
done
timeout 300 python scripts/tn_micro.py > gpurun_out/r5/tn_micro_base.txt 2>&1
CXR_TN4=1 timeout 300 python scripts/tn_micro.py > gpurun_out/r5/tn_micro_tn4_256.txt 2>&1
CXR_TN4=1 CXR_TN4_STAGES=4 timeout 300 python scripts/tn_micro.py > gpurun_out/r5/tn_micro_tn4_256_st4.txt 2>&1
CXR_TN4=1 CXR_TN4_WGS=128 timeout 300 python scripts/tn_micro.py > gpurun_out/r5/tn_micro_tn4_128.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r5/ab6_$name.json 2>/dev/null; }
for rep in 1 2; do
  run base_$rep CXR_X=0
  run tn4_256_$rep CXR_TN4=1
  run tn4_256_st4_$rep CXR_TN4=1 CXR_TN4_STAGES=4
  run tn4_128_$rep CXR_TN4=1 CXR_TN4_WGS=128
  run tn4_192_$rep CXR_TN4=1 CXR_TN4_WGS=192
done
for f in gpurun_out/r5/ab6_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['weight_grad_kernel']['achieved'],1))"; done
CXR_TEST_PHASES=1 timeout 600 python -m pytest tests/test_dp_gpu.py -q -x -s -k "dp8" 2>&1 | grep "dp8 rank\|passed\|failed" 
