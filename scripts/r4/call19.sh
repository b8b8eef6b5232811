#!/bin/bash
# GPU call 19 of round 4: reworked LoRA up / outer kernels -- parity, SCST tests, phase times, re-scoring kernel table
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lora or topk or top_k or select or sample" > gpurun_out/r4/t19.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t19.log
tail -4 gpurun_out/r4/t19.log
timeout 1200 python -m pytest tests/test_reward_scst_gpu.py -x -q -m gpu > gpurun_out/r4/t19b.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t19b.log
tail -4 gpurun_out/r4/t19b.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "longitudinal or lora or train" > gpurun_out/r4/t19c.log 2>&1; echo "rc=$?" >> gpurun_out/r4/t19c.log
tail -4 gpurun_out/r4/t19c.log
for rep in 1 2; do
  timeout 300 python scripts/scst_breakdown.py 2>/dev/null | grep -E "sample|re-score"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/rescore_prof3 -- python3 $GRAFT_REPO_ROOT/scripts/r4/rescore_profile.py > $GRAFT_REPO_ROOT/gpurun_out/r4/rescore_prof3.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/kstats.py gpurun_out/r4/rescore_prof3 21 40 | grep -v "dec_gemm\|attn_cross_mfma\|attn_decode\|sample_topk\|decode_step"
