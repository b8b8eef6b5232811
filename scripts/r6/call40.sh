#!/bin/bash
# round 6 call 40: row-strip epilogue: second-operand / row-factor loads of tile row mt + 1 requested while tile row mt goes through LDS (CXR_STRIP_DEBUG=8: as before)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -q -x -k "column_slices or row_strip or gemm_nt_group" > $O/call40_tests.log 2>&1; tail -n 2 $O/call40_tests.log
for d in 8 0; do echo "== CXR_STRIP_DEBUG=$d" | tee -a $O/call40_micro.log; CXR_STRIP_DEBUG=$d timeout 300 python scripts/r6/strip_wide_micro.py 2>&1 | grep -v amdgpu.ids | tee -a $O/call40_micro.log
  STRIP_QUICK=1 CXR_STRIP_DEBUG=$d timeout 300 python scripts/r6/strip_micro.py 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $O/call40_micro.log; done
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a $O/call40_step.log; }
for rep in 1 2 3; do
  run CXR_STRIP_DEBUG=8
  run CXR_STRIP_DEBUG=0
done
