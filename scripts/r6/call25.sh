#!/bin/bash
# round 6 call 25: the drawn token takes the k-th entry's place (exactly top_k finite entries per processed row) -- kernel, model and full-size SCST tests;
# class-token gradient row zeroed alone instead of a 28-MB fill: TF step A/B is not possible by switch, so the step is timed before / after by stash (two runs each)
mkdir -p gpurun_out/r6
python -m pytest tests/test_kernels_gpu.py -q -x -k "softmax or reinforce or topk or top_k or ce_" > gpurun_out/r6/call25_tests.log 2>&1; tail -n 3 gpurun_out/r6/call25_tests.log
python -m pytest tests/test_fullsize_scst_gpu.py tests/test_reward_scst_gpu.py -q -x -s >> gpurun_out/r6/call25_tests.log 2>&1; tail -n 3 gpurun_out/r6/call25_tests.log
grep -a "finite entries per processed row" gpurun_out/r6/call25_tests.log
python -m pytest tests/test_model_gpu.py -q -x -k "scst or sampl or boundary or full_size_tf_gradients or encoder" >> gpurun_out/r6/call25_tests.log 2>&1; tail -n 3 gpurun_out/r6/call25_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a gpurun_out/r6/call25_step.log; }
for rep in 1 2 3; do
  run CXR_ENC_FULLFILL=0
  run CXR_ENC_FULLFILL=1
done
