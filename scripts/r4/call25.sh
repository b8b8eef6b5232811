#!/bin/bash
# GPU call 25 of round 4: the pixel im2col matrix of the last weight-gradient GEMM queued at the start of the encoder backward instead of in the step's tail
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "step or train or grad or encoder" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab25_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run late_$rep CXR_EARLY_PATCH_COL=0
  run early_$rep CXR_EARLY_PATCH_COL=1
done
for f in gpurun_out/r4/ab25_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
