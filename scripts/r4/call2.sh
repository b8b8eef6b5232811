#!/bin/bash
# GPU call 2 of round 4: new-kernel tests, attention micro-benchmark, same-box A/B of the round-4 switches on the TF step
mkdir -p gpurun_out/r4
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn or deferred or attention or beam or dwproj" > gpurun_out/r4/t2a.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t2a.log)
(timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -x -q -k "dp2 or accumulation or rccl or binding or fused_adamw or reference_caller or single_image or greedy_and_beam or autograd_bridges" > gpurun_out/r4/t2b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t2b.log)
timeout 300 python scripts/attn_micro.py > gpurun_out/r4/attn_micro.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
for rep in 1 2; do
  CXR_TN_DEFER=0 CXR_TN3=0 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab_base_$rep.json 2>/dev/null
  CXR_TN_DEFER=1 CXR_TN3=0 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab_defer_$rep.json 2>/dev/null
  CXR_TN_DEFER=1 CXR_TN3=1 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab_defer_tn3_$rep.json 2>/dev/null
  CXR_TN_DEFER=1 CXR_TN3=1 CXR_ATTN_DKDV=2 timeout 300 $B > gpurun_out/r4/ab_all_$rep.json 2>/dev/null
done
tail -3 gpurun_out/r4/t2a.log gpurun_out/r4/t2b.log
for f in gpurun_out/r4/ab_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"; done
cat gpurun_out/r4/attn_micro.txt | tail -12
