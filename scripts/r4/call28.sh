#!/bin/bash
# GPU call 28 of round 4: the last weight-gradient GEMM of the step (patch embedding) on the main stream, beside the first stage's other weight gradients instead of behind them
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "step or train or grad or encoder" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab28_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run late_$rep CXR_TAIL_ON_MAIN=0
  run early_$rep CXR_TAIL_ON_MAIN=1
done
for f in gpurun_out/r4/ab28_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
