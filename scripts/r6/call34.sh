#!/bin/bash
# round 6 call 34: GPU idle time inside the SCST steps (synthetic ids and string round trip), from a kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/idleprof -- python3 $R/bench.py --steps 1 --warmup 0 --scst-steps 6 --no-extras --no-cpu-baseline --no-dropin > $O/call34_prof.log 2>&1; echo trace $?
f=$(ls /tmp/idleprof/*/*kernel_trace.csv | head -1)
python3 $R/scripts/r6/idle_analysis.py $f adamw 95 140 > $O/call34_idle.txt 2>&1; cat $O/call34_idle.txt
tail -c 1500 $O/call34_prof.log | tr ',' '\n' | grep -E "ms_per_step|\"value\"" | head -12
