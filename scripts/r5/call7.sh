#!/bin/bash
# GPU call 7 of round 5: one-pass top-k threshold (parity + time), faster 8-rank rehearsal tests, full bench line on the current tree, re-scoring kernel table
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "topk or softmax_ce or select or sample" > gpurun_out/r5/topk_tests.log 2>&1; tail -3 gpurun_out/r5/topk_tests.log
python - <<'PY' > gpurun_out/r5/topk_time.txt 2>&1
import torch, sys
sys.path.insert(0, '.')
from cxrmate_amd import ops
x = torch.randn(4080, 30000, device='cuda') * 3
for k in (50, 129):
    for _ in range(3): ops.topk_threshold(x, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.topk_threshold(x, k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"topk_threshold 4080 x 30000 fp32, k = {k} ({'one pass' if k <= 128 else 'radix select (round 4 kernel)'}): {us:.1f} us = {4080*30000*4/us/1e6:.2f} TB/s of row bytes")
PY
cat gpurun_out/r5/topk_time.txt
timeout 900 python -m pytest tests/test_dp_gpu.py -q -x -k "dp8 or bench_gpus_8" --durations=3 > gpurun_out/r5/dp8_tests.log 2>&1; tail -8 gpurun_out/r5/dp8_tests.log
timeout 1200 python bench.py > gpurun_out/r5/bench_full_a.json 2> gpurun_out/r5/bench_full_a.err; tail -c 600 gpurun_out/r5/bench_full_a.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5/bench_full_a.json').read().strip().splitlines()[-1])
print('tf', d['ms_per_step'], 'frac', d['roofline']['frac'])
s = d.get('scst', {})
print('scst value', s.get('value'), s.get('headline_is'), 'ms', s.get('ms_per_step'), 'synthetic', (s.get('synthetic_ids') or {}).get('ms_per_step'), 'strings', json.dumps(s.get('string_round_trip'))[:600])
for k in ('forward_only', 'tf_single', 'tf_dropin', 'scst_dropin', 'scst_c5', 'beam_generation', 'cpu_baseline'):
    v = d.get(k, {})
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'ms', 'error', 'vs_fused_step', 'vs_fused_string_round_trip_step') if kk in v})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/prof_rescore -o rescore -- python3 $GRAFT_REPO_ROOT/scripts/r5/rescore_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r5/prof_rescore.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r5/prof_rescore.log; ls $GRAFT_REPO_ROOT/gpurun_out/r5/prof_rescore | head
