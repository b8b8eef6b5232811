#!/bin/bash
# GPU call 31 of round 4: band height of the fused q/k/v projection passes (CXR_DW3_BAND): workgroup count vs rounds of 512 resident workgroups
for b in 0 6 8 10 12; do echo "== CXR_DW3_BAND=$b"; CXR_DW3_BAND=$b timeout 200 python scripts/dwproj_micro.py 2>/dev/null | cut -c1-170; done
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
for rep in 1 2; do for b in 0 6 10; do
  CXR_DW3_BAND=$b timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band $b rep $rep', round(d['ms_per_step'],3))"
done; done
