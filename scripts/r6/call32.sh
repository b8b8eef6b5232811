#!/bin/bash
# round 6 call 32: host time of the eager TF step against its wall time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout 600 python scripts/r6/host_ahead.py > $O/call32_host.log 2>&1; head -c 9000 $O/call32_host.log
