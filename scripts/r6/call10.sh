#!/bin/bash
# round 6 call 10: third-stream (gradient all-reduce stand-in) A/B in the TF step; aten operators of the forward; the fixed switch group
mkdir -p gpurun_out/r6
python scripts/r6/third_stream_ab.py > gpurun_out/r6/call10_third_stream.log 2>&1
python scripts/r6/aten_fwd.py > gpurun_out/r6/call10_aten_fwd.log 2>&1
python -m pytest tests/test_model_gpu.py -q -k "shipped_ab_switches and training" > gpurun_out/r6/call10_tests.log 2>&1
grep -v amdgpu.ids gpurun_out/r6/call10_third_stream.log; tail -n 32 gpurun_out/r6/call10_aten_fwd.log; tail -n 3 gpurun_out/r6/call10_tests.log
