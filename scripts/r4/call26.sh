#!/bin/bash
# GPU call 26 of round 4: workgroup target of the first stage's weight gradients (the last launches of the step, mostly after the main stream has finished)
mkdir -p gpurun_out/r4
timeout 600 env CXR_TN_TAIL_WGS=352 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "step or train or grad" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab26_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run w0_$rep CXR_TN_TAIL_WGS=0
  run w352_$rep CXR_TN_TAIL_WGS=352
  run w704_$rep CXR_TN_TAIL_WGS=704
done
for f in gpurun_out/r4/ab26_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
