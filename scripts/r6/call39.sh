#!/bin/bash
# round 6 call 39: the FFN-up input-gradient product alone: tiled / W-stationary / column-sliced strips
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
timeout 300 python scripts/r6/strip_wide_micro.py 2>&1 | grep -v amdgpu.ids | tee $O/call39_micro.log
