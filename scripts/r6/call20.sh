#!/bin/bash
# round 6 call 20: full bench line on the current tree
mkdir -p gpurun_out/r6
python bench.py --steps 20 --warmup 5 > gpurun_out/r6/call20_bench.json 2> gpurun_out/r6/call20_bench.err
python - <<'PY'
import json
t = open('gpurun_out/r6/call20_bench.json').read(); d = json.loads(t[t.index('{"metric"'):])
s = d['scst']
print('tf', round(d['ms_per_step'], 2), round(d['value']), 'frac', round(d['roofline']['frac'], 4), 'wgrad', round(d['roofline']['weight_grad_kernel']['achieved'], 1))
print('scst', s['headline_is'], round(s['ms_per_step'], 2), round(s['value'], 3), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'r512', s['string_round_trip'].get('r512', {}).get('ms_per_step'), 'us/tok', round(s['roofline']['us_per_token_step'], 1))
for k in ('forward_only', 'tf_single', 'tf_dropin', 'scst_dropin', 'scst_c5', 'beam_generation', 'cpu_baseline'):
    v = d.get(k, {}); print(k, {kk: (round(v[kk], 3) if isinstance(v[kk], float) else v[kk]) for kk in ('value', 'ms_per_step', 'ms', 'frac', 'error', 'ms_per_batch', 'us_per_token_step') if kk in v})
print('c5 enc', d['scst_c5'].get('encoder_forward_ms'))
PY
