#!/bin/bash
# GPU call 22 of round 5: one string worker per half -- parity, host/GPU timeline, then the SCST bench key with / without the child processes, alternating
python -m pytest tests/test_reward_scst_gpu.py -q -x 2>&1 | tail -2
python scripts/r5/scst_timeline.py worker 2>&1 | tail -8
python scripts/r5/scst_timeline.py 2>&1 | tail -8
for rep in 1 2 3; do for v in worker inproc; do
  if [ $v = inproc ]; then export CXR_STRING_WORKER=0; else unset CXR_STRING_WORKER; fi
  python bench.py --steps 5 --warmup 2 --no-extras --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scst']; print('$v', round(s['ms_per_step'],2), round(s['synthetic_ids']['ms_per_step'],2), round(s['string_round_trip']['vs_synthetic_ids_step'],4))"
done; done
