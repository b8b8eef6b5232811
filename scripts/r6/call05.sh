#!/bin/bash
# round 6 call 5: 384 x 192 weight-gradient blocks (gemm_tn5_kernel): tests, alone (tn_micro), in the TF step (two alternations CXR_TN5=0 / 1)
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn or linear_bwd or attention_bwd or cross" > gpurun_out/r6/call05_tests.log 2>&1
tail -n 3 gpurun_out/r6/call05_tests.log
for v in 0 1; do echo "== CXR_TN5=$v" >> gpurun_out/r6/call05_micro.log; CXR_TN5=$v python scripts/tn_micro.py 2>&1 | grep -E "I=  384 J=  384|I= 1536 J=  384|I=  384 J= 1536|I=  384 J= 1728" >> gpurun_out/r6/call05_micro.log; done
cat gpurun_out/r6/call05_micro.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
for rep in 1 2; do for v in 0 1; do
  CXR_TN5=$v python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/rep $rep CXR_TN5=$v /" | tee -a gpurun_out/r6/call05_step.log
done; done
