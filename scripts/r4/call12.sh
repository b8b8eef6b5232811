#!/bin/bash
# GPU call 12 of round 4: wave priority of the main stream's kernels against the weight-gradient stream's (build-time switch, two extra libraries)
mkdir -p gpurun_out/r4
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
L=$GRAFT_REPO_ROOT/cxrmate_amd/lib
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r4/ab12_$name.json 2>/dev/null; }
for rep in 1 2 3; do
  run prio0_$rep CXR_X=0
  run prio1_$rep CXR_LIB=$L/libcxrmate_hip_prio1.so
  run prio3_$rep CXR_LIB=$L/libcxrmate_hip_prio3.so
done
for f in gpurun_out/r4/ab12_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
