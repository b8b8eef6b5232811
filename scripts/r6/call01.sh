#!/bin/bash
# round 6 call 1: the new full-size gradient parity tests + the stage-1 / stage-2 attention backward shapes
mkdir -p gpurun_out/r6
python -m pytest tests/test_model_gpu.py -x -q -s -k "full_size_tf_gradients or c5_scst_reinforce" > gpurun_out/r6/call01_grads.log 2>&1
echo "rc=$?" >> gpurun_out/r6/call01_grads.log
python -m pytest tests/test_kernels_gpu.py -q -k "attention_bwd or attention_dropout_fwd_bwd" > gpurun_out/r6/call01_attn.log 2>&1
echo "rc=$?" >> gpurun_out/r6/call01_attn.log
tail -5 gpurun_out/r6/call01_grads.log gpurun_out/r6/call01_attn.log
