#!/bin/bash
# GPU call 11 of round 4: K/V-resident attention forward -- tests, micro-benchmark, same-box A/B
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > gpurun_out/r4/t11a.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t11a.log)
timeout 300 python scripts/attn_micro.py > gpurun_out/r4/attn_micro11.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-scst --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab11_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run fwd2_$rep CXR_ATT_KVRES=0
  run kvres_$rep CXR_ATT_KVRES=1
done
tail -n 4 gpurun_out/r4/t11a.log
grep "stage 3\|config" gpurun_out/r4/attn_micro11.txt
for f in gpurun_out/r4/ab11_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['forward_only']['ms'],3), round(d['tf_single']['ms_per_step'],3))"; done
