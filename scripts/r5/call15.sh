#!/bin/bash
# GPU call 15 of round 5: non-temporal operand loads in the weight-gradient kernel (protect the main stream's L2 residency?)
mkdir -p gpurun_out/r5
CXR_TN_NT=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r5/ab15_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run base_$rep CXR_X=0
  run nt_$rep CXR_TN_NT=1
done
for f in gpurun_out/r5/ab15_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['weight_grad_kernel']['achieved'],1))"; done
