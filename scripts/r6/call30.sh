#!/bin/bash
# round 6 call 30: the step's last weight gradient (patch embedding) on the idle main stream instead of behind the side stream's backlog (CXR_TAIL_ON_MAIN)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_model_gpu.py -q -x -k "full_size_tf_gradients or tf_single or train_mode or graph" > $O/call30_tests.log 2>&1; tail -n 2 $O/call30_tests.log
python -m pytest tests/test_fullsize_gpu.py tests/test_dp_gpu.py -q -x >> $O/call30_tests.log 2>&1; tail -n 2 $O/call30_tests.log
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 30 --warmup 5"
run() { env "$@" python $CMD 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$* /" | tee -a $O/call30_step.log; }
for rep in 1 2 3; do
  run CXR_TAIL_ON_MAIN=0
  run CXR_TAIL_ON_MAIN=1
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tailprof -- python3 $R/bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 8 --warmup 3 > $O/call30_prof.log 2>&1; echo trace $?
f=$(ls /tmp/tailprof/*/*kernel_trace.csv | head -1)
python3 $R/scripts/r6/tail_analysis.py $f > $O/call30_tail.txt 2>&1; head -16 $O/call30_tail.txt; tail -n 2 $O/call30_tail.txt
