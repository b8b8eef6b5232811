#!/bin/bash
# round 6 call 15: staggered row-strip loop -- bit-identity, alone, ablation, in the TF step
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "row_strip" > gpurun_out/r6/call15_tests.log 2>&1; tail -n 3 gpurun_out/r6/call15_tests.log
for d in 0 1 2 4 7; do echo "== CXR_STRIP_DEBUG=$d" >> gpurun_out/r6/call15_micro.log; CXR_STRIP_DEBUG=$d STRIP_QUICK=1 python scripts/r6/strip_micro.py 2>&1 | grep -v amdgpu | grep "plain" >> gpurun_out/r6/call15_micro.log; done
cat gpurun_out/r6/call15_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call15_step.log; }
for rep in 1 2; do
  run CXR_STRIP_STAGGER=0
  run CXR_STRIP_STAGGER=1
  run CXR_STRIP_STAGGER=1 CXR_STRIP_STAGES=4
done
