#!/bin/bash
# GPU call 18 of round 5: the reward's string work in a child process -- parity, then SCST string step with / without it, same box, alternating
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_reward_scst_gpu.py -q -x 2>&1 | tail -3
for rep in 1 2 3; do
for v in worker inproc; do
if [ $v = inproc ]; then export CXR_STRING_WORKER=0; else unset CXR_STRING_WORKER; fi
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-dropin --no-cpu-baseline > gpurun_out/r5/b18_${v}_$rep.json 2>/dev/null
python - <<PY
import json
d = json.loads(open('gpurun_out/r5/b18_${v}_$rep.json').read().strip().splitlines()[-1])
s = d['scst']; r = s['string_round_trip']
print('$v', 'tf', round(d['ms_per_step'], 2), 'scst string', round(s['ms_per_step'], 2), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'ratio', round(r['vs_synthetic_ids_step'], 4), r.get('string_worker'))
PY
done
done
