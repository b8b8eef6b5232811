#!/bin/bash
# GPU call 35 of round 4: re-sweep of the weight-gradient launch parameters on the final tree (two alternations)
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
for rep in 1 2; do
  for cfg in "X=0" "CXR_TN2_MIN=2048" "CXR_TN2_MIN=8192" "CXR_TN2_WGS=80" "CXR_TN2_WGS=112" "CXR_TN_WGS=144" "CXR_TN_WGS=208" "CXR_LN_BWD_GRID=384" "CXR_LN_BWD_GRID=768" "CXR_WGRAD_BATCH=2" "CXR_WGRAD_BATCH=6"; do
    env $cfg timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg rep $rep', round(d['ms_per_step'],3))"
  done
done
