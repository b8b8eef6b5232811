#!/bin/bash
# GPU call 10 of round 5: SCST step with / without the host round trip between decode and re-scoring pass, same box, alternating
mkdir -p gpurun_out/r5
for rep in 1 2 3; do
for v in ahead sync; do
if [ $v = sync ]; then export CXR_SCST_SYNC_STRIP=1; else unset CXR_SCST_SYNC_STRIP; fi
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-dropin --no-cpu-baseline > gpurun_out/r5/bench_scst_${v}_$rep.json 2>/dev/null
python - <<PY
import json
d = json.loads(open('gpurun_out/r5/bench_scst_${v}_$rep.json').read().strip().splitlines()[-1])
s = d['scst']
print('$v', 'tf', round(d['ms_per_step'], 2), 'scst', s['headline_is'], round(s['ms_per_step'], 2), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'ratio', round(s['string_round_trip']['vs_synthetic_ids_step'], 4), 'decode ms', round(s['roofline']['decode_ms_per_step'], 2))
PY
done
done
