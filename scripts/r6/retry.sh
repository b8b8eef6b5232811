#!/bin/bash
# retries a gpurun call while the pod's GPU slots are busy (exit code 3 = nothing charged)
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
