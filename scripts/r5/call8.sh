#!/bin/bash
# GPU call 8 of round 5: weight-gradient launches forked in front of the attention-backward / depthwise-projection kernels (CXR_WGRAD_GATE=1), with the
# shipped and with the co-resident weight-gradient kernel; same-box alternation, two repetitions
mkdir -p gpurun_out/r5
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r5/ab8_$name.json 2>/dev/null; }
for rep in 1 2; do
  run base_$rep CXR_X=0
  run gate_$rep CXR_WGRAD_GATE=1
  run gate_tn4_256_$rep CXR_WGRAD_GATE=1 CXR_TN4=1
  run gate_tn4_192_$rep CXR_WGRAD_GATE=1 CXR_TN4=1 CXR_TN4_WGS=192
  run gate_tn2_160_$rep CXR_WGRAD_GATE=1 CXR_TN2_WGS=160
  run gate_max8_$rep CXR_WGRAD_GATE=1 CXR_WGRAD_GATE_MAX=8
  run gate_excl_$rep CXR_WGRAD_GATE=1 CXR_GEMM_EXCL_ALWAYS=1
  run gate_tn4_excl_$rep CXR_WGRAD_GATE=1 CXR_TN4=1 CXR_GEMM_EXCL_ALWAYS=1
done
for f in gpurun_out/r5/ab8_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['weight_grad_kernel']['achieved'],1))"; done
