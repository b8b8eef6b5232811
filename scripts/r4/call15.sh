#!/bin/bash
# GPU call 15 of round 4: kernel table of the SCST re-scoring phase (TF forward over 16 sampled rows + REINFORCE + decoder backward)
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/rescore_prof -- python3 $GRAFT_REPO_ROOT/scripts/r4/rescore_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r4/rescore_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r4/rescore_prof.log
python scripts/kstats.py gpurun_out/r4/rescore_prof 21 45
