#!/bin/bash
# round 6 call 8: 384 x 192 blocks for the short-reduction 384 x 384 gradients too (R = 9280: key / value projections of CvT stage 3)? alone + in the step
mkdir -p gpurun_out/r6
for v in 4096 512; do echo "== CXR_TN5_MIN=$v" >> gpurun_out/r6/call08_micro.log; CXR_TN5_MIN=$v python scripts/tn_micro.py 2>&1 | grep -E "R=  9280|R= 36928 I=  384 J=  384" >> gpurun_out/r6/call08_micro.log; done
cat gpurun_out/r6/call08_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call08_step.log; }
for rep in 1 2; do
  run CXR_TN5_MIN=4096
  run CXR_TN5_MIN=512
  run CXR_TN5_MIN=512 CXR_TN5_WGS=56
  run CXR_TN5_MIN=4096 CXR_TN5_WGS=56
  run CXR_TN5_MIN=4096 CXR_TN5_WGS=72
done
