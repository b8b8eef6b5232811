#!/bin/bash
# round 6 call 41: child processes of the SCST string round trip (CXR_STRING_WORKERS = 2 | 4 | 8: one | two | four per half)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_reward_scst_gpu.py -q -x -k "string or real_strings or worker" > $O/call41_tests.log 2>&1; tail -n 2 $O/call41_tests.log
run() { env "$@" python bench.py --no-extras --no-cpu-baseline --no-dropin --steps 2 --warmup 1 --scst-steps 10 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); d = json.loads(t[t.index('{\"metric\"'):]); s = d['scst']
print('$*', 'strings', round(s['string_round_trip']['ms_per_step'], 2), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'r512', round(s['string_round_trip']['r512']['ms_per_step'], 2), 'served', s['string_round_trip']['string_worker'])
" | tee -a $O/call41_scst.log; }
for rep in 1 2; do
  run CXR_STRING_WORKERS=2
  run CXR_STRING_WORKERS=4
  run CXR_STRING_WORKERS=8
done
