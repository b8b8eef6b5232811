#!/bin/bash
# GPU call 11 of round 5: why is the pre-queued re-scoring pass slower? SCST ahead / sync x weight-gradient stream on / off; TF step with a host sync per step
mkdir -p gpurun_out/r5
show() { python - <<PY
import json
d = json.loads(open('$1').read().strip().splitlines()[-1])
s = d.get('scst') or {}
print('$2', 'tf', round(d['ms_per_step'], 2), 'scst synthetic', round((s.get('synthetic_ids') or {}).get('ms_per_step', 0), 2), 'string', round(s.get('ms_per_step', 0), 2))
PY
}
for rep in 1 2; do
  for v in "sync_2s:CXR_X=0" "ahead_2s:CXR_SCST_AHEAD=1" "sync_1s:CXR_WGRAD_OVERLAP=0" "ahead_1s:CXR_SCST_AHEAD=1 CXR_WGRAD_OVERLAP=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-dropin --no-cpu-baseline > gpurun_out/r5/b11_${name}_$rep.json 2>/dev/null
    show gpurun_out/r5/b11_${name}_$rep.json ${name}_$rep
  done
  CXR_BENCH_SYNC_EACH=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin > gpurun_out/r5/b11_tfsync_$rep.json 2>/dev/null; show gpurun_out/r5/b11_tfsync_$rep.json tf_sync_each_$rep
  timeout 300 python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin > gpurun_out/r5/b11_tfbase_$rep.json 2>/dev/null; show gpurun_out/r5/b11_tfbase_$rep.json tf_base_$rep
done
