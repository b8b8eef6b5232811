#!/bin/bash
# round 6 call 35: the end of an SCST step's backward on the two queues (as call 29 for the TF step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/idleprof -- python3 $R/bench.py --steps 1 --warmup 0 --scst-steps 6 --no-extras --no-cpu-baseline --no-dropin > $O/call35_prof.log 2>&1; echo trace $?
f=$(ls /tmp/idleprof/*/*kernel_trace.csv | head -1)
python3 $R/scripts/r6/tail_analysis.py $f 95 140 > $O/call35_tail.txt 2>&1; head -60 $O/call35_tail.txt; tail -n 2 $O/call35_tail.txt
