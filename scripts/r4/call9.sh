#!/bin/bash
# GPU call 9 of round 4: implicit-GEMM stage embeddings -- tests + same-box A/B
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "implicit or im2col or gemm_nt" > gpurun_out/r4/t9a.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t9a.log)
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -k "encoder or tf_ or train or fp8 or full" > gpurun_out/r4/t9b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t9b.log)
B="python bench.py --steps 20 --warmup 5 --no-scst --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab9_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run explicit_$rep CXR_IMPLICIT_EMBED=0
  run implicit_$rep CXR_IMPLICIT_EMBED=1
done
tail -n 3 gpurun_out/r4/t9a.log gpurun_out/r4/t9b.log
for f in gpurun_out/r4/ab9_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['forward_only']['ms'],3), round(d['tf_single']['ms_per_step'],3))"; done
