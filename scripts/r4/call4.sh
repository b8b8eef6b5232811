#!/bin/bash
# GPU call 4 of round 4: fixed tests, attention micro (5-wave dK/dV at Tk = 145), weight-gradient stream sweeps on the TF step
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn or deferred or beam or attention" > gpurun_out/r4/t4a.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t4a.log)
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -k "binding or fused_adamw or reference_caller or single_image or dp2 or accumulation or rccl" > gpurun_out/r4/t4b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t4b.log)
timeout 300 python scripts/attn_micro.py > gpurun_out/r4/attn_micro4.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab4_$name.json 2>/dev/null; }
for rep in 1 2; do
  run base_$rep CXR_TN_DEFER=0 CXR_TN3=0
  run defer_$rep CXR_TN_DEFER=1 CXR_TN3=0
  run tn3_w96_$rep CXR_TN_DEFER=1 CXR_TN3=1
  run tn3_w64_$rep CXR_TN_DEFER=1 CXR_TN3=1 CXR_TN2_WGS=64
  run tn3_w128_$rep CXR_TN_DEFER=1 CXR_TN3=1 CXR_TN2_WGS=128
  run batch8_$rep CXR_TN_DEFER=1 CXR_TN3=0 CXR_WGRAD_BATCH=8
done
tail -n 3 gpurun_out/r4/t4a.log gpurun_out/r4/t4b.log
for f in gpurun_out/r4/ab4_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
grep "stage 3\|per step" gpurun_out/r4/attn_micro4.txt
