#!/bin/bash
# GPU call 20 of round 5: final tree -- full -m gpu suite, smoke, full bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_full_output.json 2> $O/bench.err; tail -c 200 $O/bench.err
python - <<PY
import json
d = json.loads(open('$O/r05_bench_full_output.json').read().strip().splitlines()[-1])
s = d['scst']
print('tf', round(d['ms_per_step'], 2), d['roofline']['frac'], 'scst', s['headline_is'], round(s['ms_per_step'], 2), s['value'], 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'ratio', s['string_round_trip'].get('vs_synthetic_ids_step'))
for k in ('forward_only', 'tf_single', 'tf_dropin', 'scst_dropin', 'scst_c5', 'beam_generation', 'cpu_baseline'):
    v = d.get(k, {}); print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'ms', 'frac', 'error', 'vs_fused_step', 'ms_per_batch') if kk in v})
PY
