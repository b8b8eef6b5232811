#!/bin/bash
# GPU call 33 of round 4: does the SCST key depend on what the process ran before it? (default order: TF, forward_only, tf_single, tf_dropin, then scst)
for rep in 1 2; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default order      : tf', round(d['ms_per_step'],2), 'scst', round(d['scst']['ms_per_step'],2), 'decode', round(d['scst']['roofline']['decode_ms_per_step'],2))"
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scst right after TF: tf', round(d['ms_per_step'],2), 'scst', round(d['scst']['ms_per_step'],2), 'decode', round(d['scst']['roofline']['decode_ms_per_step'],2))"
done
