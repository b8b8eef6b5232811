#!/bin/bash
# GPU call 13 of round 5: WHERE the weight-gradient stream costs the main stream its 4.9 ms -- per-kernel durations of the TF step with and without the
# weight-gradient GEMMs (CXR_WGRAD_SKIP=1: timing experiment, wrong gradients), two kernel traces of the same command on the same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 20 --warmup 5"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/with -o tf -- python3 $R/$CMD > $O/with.log 2>&1; echo with $?
export CXR_WGRAD_SKIP=1 CXR_DEBUG_TIMING=1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/skip -o tf -- python3 $R/$CMD > $O/skip.log 2>&1; echo skip $?
unset CXR_WGRAD_SKIP CXR_DEBUG_TIMING
python3 $R/scripts/rocprof_summary.py $(ls $O/with/*.db | head -1) > $O/with.csv
python3 $R/scripts/rocprof_summary.py $(ls $O/skip/*.db | head -1) > $O/skip.csv
grep ms_per_step $O/with.log | head -1 | cut -c1-200; grep -o '"ms_per_step": [0-9.]*' $O/with.log | head -1; grep -o '"ms_per_step": [0-9.]*' $O/skip.log | head -1
rm -rf $O/with $O/skip
