#!/bin/bash
# GPU call 9 of round 5: SCST step that queues its re-scoring pass behind the decode without a host round trip: parity suites + the two SCST numbers
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_reward_scst_gpu.py tests/test_fullsize_scst_gpu.py tests/test_precision16_gpu.py -q -x > gpurun_out/r5/scst_tests.log 2>&1; tail -3 gpurun_out/r5/scst_tests.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "scst or speculative or boundary or prompted" > gpurun_out/r5/scst_tests2.log 2>&1; tail -3 gpurun_out/r5/scst_tests2.log
for rep in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-dropin --no-cpu-baseline > gpurun_out/r5/bench_scst_$rep.json 2>/dev/null
python - <<PY
import json
d = json.loads(open('gpurun_out/r5/bench_scst_$rep.json').read().strip().splitlines()[-1])
s = d['scst']
print('tf', round(d['ms_per_step'], 2), 'scst', s['headline_is'], round(s['ms_per_step'], 2), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'ratio', round(s['string_round_trip']['vs_synthetic_ids_step'], 4), s['string_round_trip']['host_ms'], 'decode ms', round(s['roofline']['decode_ms_per_step'], 2))
PY
done
