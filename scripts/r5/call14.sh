#!/bin/bash
# GPU call 14 of round 5: fewer bytes on the weight-gradient stream? atomics instead of partial tiles + reduce (non-deterministic: lab only)
mkdir -p gpurun_out/r5
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r5/ab14_$name.json 2>/dev/null; }
for rep in 1 2; do
  run base_$rep CXR_X=0
  run atomics_$rep CXR_TN_ATOMICS=1
  run wgs64_$rep CXR_TN2_WGS=64 CXR_TN_WGS=128
done
for f in gpurun_out/r5/ab14_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['weight_grad_kernel']['achieved'],1))"; done
