#!/bin/bash
# round 6 call 26: W-stationary K = 384 kernel beside the weight-gradient stream with a partition cut to the CUs that are left (CXR_WS_SHARED_WGS):
# the FFN-up input-gradient GEMM (36928 x 1536 x 384 + GELU') is 16 x 147 us of the main stream on the tiled kernel
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "softmax or reinforce" > gpurun_out/r6/call26_tests.log 2>&1; tail -n 2 gpurun_out/r6/call26_tests.log
CXR_WS_SHARED_WGS=160 python -m pytest tests/test_model_gpu.py -q -x -k "full_size_tf_gradients" >> gpurun_out/r6/call26_tests.log 2>&1; tail -n 2 gpurun_out/r6/call26_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call26_step.log; }
for rep in 1 2; do
  run CXR_WS_SHARED_WGS=0
  run CXR_WS_SHARED_WGS=128
  run CXR_WS_SHARED_WGS=160
  run CXR_WS_SHARED_WGS=192
  run CXR_WS_SHARED_WGS=256
done
run CXR_GEMM_EXCL_ALWAYS=1
run CXR_WS_SHARED_WGS=0
