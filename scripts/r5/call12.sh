#!/bin/bash
# GPU call 12 of round 5: evidence pass on the final tree -- full -m gpu suite, full bench line, kernel trace of the TF step, per-shape in-step GEMM table,
# SQ counters of the weight-gradient / NT kernels with the shipped and the co-resident weight-gradient kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; tail -14 $O/gpu_suite.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_full_output.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 300 python scripts/gemm_profile.py > $O/r05_gemm_shapes_instep.txt 2>/dev/null; echo shapes $?
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 20 --warmup 5"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/tf_trace -o tf -- python3 $R/$CMD > $O/tf_trace.log 2>&1; echo trace $?
python3 $R/scripts/rocprof_summary.py $(ls $O/tf_trace/*.db | head -1) > $O/r05_bench_tf_step_2image_kernel_stats.csv 2>/dev/null; head -3 $O/r05_bench_tf_step_2image_kernel_stats.csv
CMD3="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 2 --warmup 1"
for v in base tn4; do
  if [ $v = tn4 ]; then export CXR_TN4=1; else unset CXR_TN4; fi
  CXR_BENCH_PREWARM=2 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/sq_$v -- python3 $R/$CMD3 > $O/sq_$v.log 2>&1; echo sq $v $?
  python3 $R/scripts/pmc_table.py $O/sq_$v gemm_ > $O/r05_pmc_sq_gemm_$v.txt 2>&1
  rm -rf $O/sq_$v
done
unset CXR_TN4
rm -rf $O/tf_trace
head -30 $O/r05_pmc_sq_gemm_tn4.txt
