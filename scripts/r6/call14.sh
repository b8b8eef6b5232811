#!/bin/bash
mkdir -p gpurun_out/r6
for d in 0 1 2 4 3 5 6 7; do echo "== CXR_STRIP_DEBUG=$d" >> gpurun_out/r6/call14_micro.log; CXR_STRIP_DEBUG=$d STRIP_QUICK=1 python scripts/r6/strip_micro.py 2>&1 | grep -v amdgpu | grep "plain" >> gpurun_out/r6/call14_micro.log; done
cat gpurun_out/r6/call14_micro.log
