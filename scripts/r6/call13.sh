#!/bin/bash
mkdir -p gpurun_out/r6
STRIP_QUICK=1 python scripts/r6/strip_micro.py > gpurun_out/r6/call13_micro.log 2>&1; grep -v amdgpu gpurun_out/r6/call13_micro.log
