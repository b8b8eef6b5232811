#!/bin/bash
# GPU call 10 of round 4: one-launch stage-1 patch embedding -- tests + same-box A/B
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "patch_embedding or implicit or im2col" > gpurun_out/r4/t10a.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t10a.log)
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_reward_scst_gpu.py -x -q > gpurun_out/r4/t10b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t10b.log)
B="python bench.py --steps 20 --warmup 5 --no-scst --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab10_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run old_$rep CXR_PATCH_EMBED_FUSED=0
  run fused_$rep CXR_PATCH_EMBED_FUSED=1
done
tail -n 4 gpurun_out/r4/t10a.log gpurun_out/r4/t10b.log
for f in gpurun_out/r4/ab10_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['forward_only']['ms'],3), round(d['tf_single']['ms_per_step'],3))"; done
