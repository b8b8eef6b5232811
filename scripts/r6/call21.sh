#!/bin/bash
# round 6 call 21: the whole -m gpu suite + smoke on the tree with the row-strip GEMMs / 384 x 192 weight-gradient blocks
mkdir -p gpurun_out/r6
( time python -m pytest tests -q -m gpu ) > gpurun_out/r6/call21_suite.log 2>&1
tail -n 8 gpurun_out/r6/call21_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
