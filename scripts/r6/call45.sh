#!/bin/bash
# round 6 call 45 (run three times, one box each): the default bench line of the closing tree -- box-to-box spread
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
T=$(date +%H%M%S)
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/call45_bench_$T.json 2> $O/call45_bench_$T.err; echo "bench rc=$?"
python - <<PY
import json
t = open("$O/call45_bench_$T.json").read(); d = json.loads(t[t.index('{"metric"'):]); s = d['scst']
print('box $T: tf', round(d['ms_per_step'], 2), 'fwd', round(d['forward_only']['ms'], 2), 'single', round(d['tf_single']['ms_per_step'], 2), 'scst strings', round(s['ms_per_step'], 2), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2),
      'r512', round(s['string_round_trip']['r512']['ms_per_step'], 2), 'us/tok', round(s['roofline']['us_per_token_step'], 1), 'c5', round(d['scst_c5']['ms_per_step'], 2), 'beam', round(d['beam_generation']['ms_per_batch'], 2),
      'dropin', round(d['tf_dropin']['ms_per_step'], 2), round(d['scst_dropin']['ms_per_step'], 2), 'cpu', round(d['cpu_baseline']['value'], 1))
PY
