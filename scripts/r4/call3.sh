#!/bin/bash
# GPU call 3 of round 4: per-shape weight-gradient GEMM timing with / without the 384 x 128 blocks, deferred-reduce thresholds on the TF step
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn or deferred or beam" > gpurun_out/r4/t3a.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t3a.log)
(timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "binding or fused_adamw or reference_caller or single_image" > gpurun_out/r4/t3b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t3b.log)
CXR_TN3=0 timeout 300 python scripts/tn_micro.py > gpurun_out/r4/tn_micro_tn3_0.txt 2>&1
CXR_TN3=1 timeout 300 python scripts/tn_micro.py > gpurun_out/r4/tn_micro_tn3_1.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
for rep in 1 2; do
  CXR_TN_DEFER=0 CXR_TN3=0 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab3_base_$rep.json 2>/dev/null
  CXR_TN_DEFER=1 CXR_TN_DEFER_MB=32 CXR_TN3=0 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab3_defer32_$rep.json 2>/dev/null
  CXR_TN_DEFER=1 CXR_TN_DEFER_MB=64 CXR_TN3=0 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab3_defer64_$rep.json 2>/dev/null
  CXR_TN_DEFER=1 CXR_TN_DEFER_MB=128 CXR_TN3=0 CXR_ATTN_DKDV=1 timeout 300 $B > gpurun_out/r4/ab3_defer128_$rep.json 2>/dev/null
done
tail -n 3 gpurun_out/r4/t3a.log gpurun_out/r4/t3b.log
for f in gpurun_out/r4/ab3_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"; done
paste gpurun_out/r4/tn_micro_tn3_0.txt gpurun_out/r4/tn_micro_tn3_1.txt | cut -c1-160
