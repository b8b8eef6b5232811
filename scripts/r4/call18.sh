#!/bin/bash
# GPU call 18 of round 4: SCST re-scoring pass -- cross K/V shared with the decode session's prefill, LM head on the sampled positions only, wave-aggregated
# histogram atomics in the top-k threshold kernel: parity, then the phase times
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_reward_scst_gpu.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r4/t18.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t18.log
tail -5 gpurun_out/r4/t18.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "topk or top_k or select or sample or ce or loss or beam" > gpurun_out/r4/t18b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t18b.log
tail -5 gpurun_out/r4/t18b.log
for rep in 1 2; do
  timeout 300 python scripts/scst_breakdown.py 2>/dev/null | grep -E "encoder|sample|re-score|reward"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/rescore_prof2 -- python3 $GRAFT_REPO_ROOT/scripts/r4/rescore_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r4/rescore_prof2.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/kstats.py gpurun_out/r4/rescore_prof2 21 40
