#!/bin/bash
# GPU call 16 of round 4: several-problems-per-launch LoRA kernels -- parity, the re-scoring phase with and without them, the SCST step A/B
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lora" > gpurun_out/r4/t16.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t16.log
tail -5 gpurun_out/r4/t16.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reward_scst_gpu.py -x -q -m gpu -k "longitudinal or lora or scst or train" > gpurun_out/r4/t16b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t16b.log
tail -5 gpurun_out/r4/t16b.log
for v in 0 1; do
  CXR_LORA_MULTI=$v timeout 300 python scripts/r4/rescore_profile.py 2>/dev/null | tail -1
done
B="python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --no-dropin --scst-steps 10"
for rep in 1 2; do for v in 0 1; do
  CXR_LORA_MULTI=$v timeout 600 $B > gpurun_out/r4/ab16_lora${v}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4/ab16_lora${v}_$rep.json').read().strip().splitlines()[-1])
print('multi=$v rep=$rep scst ms', round(d['scst']['ms_per_step'],2), 'c5', round(d['scst_c5']['ms_per_step'],2), 'tf', round(d['ms_per_step'],2))
PY
done; done
