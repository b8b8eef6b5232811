#!/bin/bash
# GPU call 6 of round 4: what the weight-gradient stream costs the step (timing experiments) + a kernel trace of the current step
mkdir -p gpurun_out/r4
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab6_$name.json 2>/dev/null; }
for rep in 1 2; do
  run base_$rep CXR_X=0
  run skipwgrad_$rep CXR_WGRAD_SKIP=1
  run onestream_$rep CXR_WGRAD_OVERLAP=0
  run exclalways_skip_$rep CXR_WGRAD_SKIP=1 CXR_GEMM_EXCL_ALWAYS=1
done
for f in gpurun_out/r4/ab6_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4/prof6 -o tf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin > $GRAFT_REPO_ROOT/gpurun_out/r4/prof6.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r4/prof6 | head
