#!/bin/bash
# round 6 call 38: column-sliced row-strip kernel for the FFN-up input gradient (36928 x 1536 x 384 + GELU') beside the weight-gradient stream (CXR_STRIP_WIDE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -q -x -k "column_slices or row_strip" > $O/call38_tests.log 2>&1; tail -n 3 $O/call38_tests.log
python -m pytest tests/test_model_gpu.py -q -x -k "full_size_tf_gradients or tf_single_logits" >> $O/call38_tests.log 2>&1; tail -n 2 $O/call38_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a $O/call38_step.log; }
for rep in 1 2 3; do
  run CXR_STRIP_WIDE=0
  run CXR_STRIP_WIDE=1
done
python scripts/gemm_profile.py 2>/dev/null | grep -E "^ +36928 +1536 +384|^ +36864 +768 +384|total" | tee $O/call38_shapes.txt
