#!/bin/bash
# GPU call 13 of round 4: dc pass of the fused q/k/v projection backward with c taken from the forward outputs (CXR_DW3_DC_FROM_Y) -- parity, then A/B of the step
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwproj" > gpurun_out/r4/t13.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t13.log
tail -5 gpurun_out/r4/t13.log
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab13_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run conv_$rep CXR_DW3_DC_FROM_Y=0
  run fromy_$rep CXR_DW3_DC_FROM_Y=1
done
for f in gpurun_out/r4/ab13_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "train or step or grad" > gpurun_out/r4/t13b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t13b.log
tail -5 gpurun_out/r4/t13b.log
