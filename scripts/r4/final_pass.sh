#!/bin/bash
# Final pass of round 4 on the frozen tree: full GPU suite, smoke, the default bench line, kernel-trace summaries of the TF step and of the SCST decode
mkdir -p gpurun_out/r4f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r4f/gpu_suite.log
tail -3 gpurun_out/r4f/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4f/smoke.log 2>&1; tail -1 gpurun_out/r4f/smoke.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r4f/bench_full_output.json 2> gpurun_out/r4f/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench_full_output.json').read().strip().splitlines()[-1])
print('TF ms', round(d['ms_per_step'],2), 'tok/s', round(d['value']), 'frac', round(d['roofline']['frac'],4))
for k in ('forward_only','tf_single','tf_dropin','scst','scst_dropin','scst_c5','beam_generation','cpu_baseline'):
    v=d.get(k)
    if isinstance(v,dict): print(k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms_per_step','value','ms_per_batch','ratio_to_fused','steps_per_sec','us_per_token_step')})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4f/tf_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r4f/tf_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4f/decode_prof -- python3 $GRAFT_REPO_ROOT/scripts/scst_decode_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r4f/decode_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4f/rescore_prof -- python3 $GRAFT_REPO_ROOT/scripts/r4/rescore_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r4f/rescore_prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r4f/*/*/ | head -30
