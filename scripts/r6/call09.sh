#!/bin/bash
# round 6 call 9: the whole -m gpu suite + smoke on the tree after the removals / the 384 x 192 blocks / the switch tests
mkdir -p gpurun_out/r6
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/r6/call09_suite.log 2>&1
tail -n 15 gpurun_out/r6/call09_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
