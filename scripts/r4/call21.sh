#!/bin/bash
# GPU call 21 of round 4: the SCST drop-in caller sequence under the kernel trace (what torch passes are left), after the LM head of its re-scoring pass
# moved to the sampled positions only
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reward_scst_gpu.py -x -q -m gpu -k "scst or caller or boundary or speculative or wrapped or scores" > gpurun_out/r4/t21.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t21.log
tail -4 gpurun_out/r4/t21.log
timeout 600 python scripts/dropin_profile.py scst 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/dropin_prof -- python3 $GRAFT_REPO_ROOT/scripts/dropin_profile.py scst > $GRAFT_REPO_ROOT/gpurun_out/r4/dropin_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/kstats.py gpurun_out/r4/dropin_prof 4 80 | grep -v "dec_gemm\|attn_cross_mfma\|attn_decode\|sample_topk\|decode_step" | head -60
