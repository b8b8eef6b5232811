#!/bin/bash
# GPU call 24 of round 4: the SCST step with the previous top-k threshold kernel (variant library) and the new one, alternated
L=$GRAFT_REPO_ROOT/cxrmate_amd/lib
B="python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --no-dropin --scst-steps 10"
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export CXR_LIB=$L/libcxrmate_hip_oldtopk.so; else unset CXR_LIB; fi
  timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scst']; print('$v', $rep, 'scst ms', round(s['ms_per_step'],2), 'decode', round(s['roofline']['decode_ms_per_step'],2), 'c5', round(d['scst_c5']['ms_per_step'],2) if 'scst_c5' in d else '')"
done; done
