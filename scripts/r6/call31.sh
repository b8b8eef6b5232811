#!/bin/bash
# round 6 call 31: idle time of the main queue inside a TF step (gaps between kernels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gapprof -- python3 $R/bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 8 --warmup 3 > $O/call31_prof.log 2>&1; echo trace $?
f=$(ls /tmp/gapprof/*/*kernel_trace.csv | head -1)
python3 $R/scripts/r6/gap_analysis.py $f > $O/call31_gaps.txt 2>&1; cat $O/call31_gaps.txt
