#!/bin/bash
# GPU call 29 of round 4: SCST step without the host synchronisation behind the decode (BOS stripping decided from the prompt)
timeout 900 python -m pytest tests/test_reward_scst_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "scst or sample or speculative or fused" 2>&1 | tail -2
B="python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --no-dropin --scst-steps 10"
for rep in 1 2 3; do for v in 0 2; do
  CXR_SCST_BOS_FROM_PROMPT=$v timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scst']; print('from_prompt=$v rep $rep scst ms', round(s['ms_per_step'],2), 'decode', round(s['roofline']['decode_ms_per_step'],2), 'strings', round(s['string_round_trip']['ms_per_step'],2))"
done; done
