#!/bin/bash
# Round-6 final pass on the frozen tree: the default bench line (-> profiles/r06_bench_full_output.json) and the evidence pass (scripts/r6/evidence.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_full_output.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r6e/r06_bench_full_output.json")
t = open(p).read(); d = json.loads(t[t.index('{"metric"'):])
s = d['scst']
print('tf', round(d['ms_per_step'], 2), round(d['value']), 'frac', round(d['roofline']['frac'], 4), 'wgrad', round(d['roofline']['weight_grad_kernel']['achieved'], 1))
print('scst', s['headline_is'], round(s['ms_per_step'], 2), round(s['value'], 3), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'r512', s['string_round_trip'].get('r512', {}).get('ms_per_step'), 'us/tok', round(s['roofline']['us_per_token_step'], 1))
for k in ('forward_only', 'tf_single', 'tf_dropin', 'scst_dropin', 'scst_c5', 'beam_generation', 'cpu_baseline'):
    v = d.get(k, {}); print(k, {kk: (round(v[kk], 3) if isinstance(v[kk], float) else v[kk]) for kk in ('value', 'ms_per_step', 'ms', 'frac', 'error', 'ms_per_batch', 'us_per_token_step') if kk in v})
PY
bash scripts/r6/evidence.sh
