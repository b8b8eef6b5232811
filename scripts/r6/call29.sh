#!/bin/bash
# round 6 call 29: the end of a TF step on the two queues (exposed weight-gradient tail?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tailprof -- python3 $R/bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 8 --warmup 3 > $O/call29_prof.log 2>&1; echo trace $?
f=$(ls /tmp/tailprof/*/*kernel_trace.csv | head -1); echo $f; head -1 $f
python3 $R/scripts/r6/tail_analysis.py $f > $O/call29_tail.txt 2>&1; cat $O/call29_tail.txt
