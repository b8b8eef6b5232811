#!/bin/bash
# GPU call 32 of round 4: at equal staging cost the taller band (variant library) vs the shorter one
L=$GRAFT_REPO_ROOT/cxrmate_amd/lib/libcxrmate_hip_band.so
CXR_LIB=$L timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k dwproj 2>&1 | tail -2
CXR_LIB=$L timeout 200 python scripts/dwproj_micro.py 2>/dev/null | cut -c1-170
B="python bench.py --steps 20 --warmup 5 --no-scst --no-extras --no-cpu-baseline --no-dropin"
for rep in 1 2 3; do for v in old new; do
  if [ $v = new ]; then export CXR_LIB=$L; else unset CXR_LIB; fi
  timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep $rep', round(d['ms_per_step'],3))"
done; done
