#!/bin/bash
# round 6 call 36 (= call 28 on the tree with the step-tail change): the whole GPU suite + smoke, then the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -x -m gpu > $O/call36_suite.log 2>&1; echo "suite rc=$?"; tail -n 3 $O/call36_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/call36_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/call36_smoke.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/call36_bench.json 2> $O/call36_bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r6/call36_bench.json")
t = open(p).read(); d = json.loads(t[t.index('{"metric"'):])
s = d['scst']
print('tf', round(d['ms_per_step'], 2), round(d['value']), 'frac', round(d['roofline']['frac'], 4), 'traffic_source', d['roofline'].get('traffic_source'))
print('scst', s['headline_is'], round(s['ms_per_step'], 2), round(s['value'], 3), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'r512', s['string_round_trip'].get('r512', {}).get('ms_per_step'), 'us/tok', round(s['roofline']['us_per_token_step'], 1))
print('encoder', s.get('encoder_forward_ms'), {k: s['encoder_roofline'][k] for k in ('achieved', 'frac')})
for k in ('forward_only', 'tf_single', 'tf_dropin', 'scst_dropin', 'scst_c5', 'beam_generation', 'cpu_baseline'):
    v = d.get(k, {}); print(k, {kk: (round(v[kk], 3) if isinstance(v[kk], float) else v[kk]) for kk in ('value', 'ms_per_step', 'ms', 'frac', 'error', 'ms_per_batch', 'us_per_token_step') if kk in v})
PY
