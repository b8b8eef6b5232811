#!/bin/bash
# the full GPU suite twice more (different random order of nothing: same order, fresh processes) -- flakiness check before the round ends
mkdir -p gpurun_out/r4
for i in 1 2; do
  timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/soak_$i.log 2>&1; echo "run $i rc=$?"; tail -2 gpurun_out/r4/soak_$i.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
