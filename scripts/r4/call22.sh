#!/bin/bash
# GPU call 22 of round 4: the encoder's last stage + head updated on the weight-gradient stream under the earlier stages' backward (CXR_EARLY_ENC_ADAMW)
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "adamw or step or train" > gpurun_out/r4/t22.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t22.log
tail -4 gpurun_out/r4/t22.log
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab22_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run late_$rep CXR_EARLY_ENC_ADAMW=0
  run early_$rep CXR_EARLY_ENC_ADAMW=1
done
for f in gpurun_out/r4/ab22_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
