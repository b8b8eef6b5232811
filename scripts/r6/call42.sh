#!/bin/bash
# round 6 call 42: the whole GPU suite + smoke on the closing tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -x -m gpu > $O/call42_suite.log 2>&1; echo "suite rc=$?"; tail -n 3 $O/call42_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/call42_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/call42_smoke.log
