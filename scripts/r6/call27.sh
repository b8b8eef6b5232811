#!/bin/bash
# round 6 call 27: decode-step GEMM launches with L2- / Infinity-Cache- / HBM-resident weights (upper bound of a cross-kernel weight prefetch)
mkdir -p gpurun_out/r6
timeout 600 python scripts/r6/dec_warm_micro.py > gpurun_out/r6/call27_micro.log 2>&1; tail -n 12 gpurun_out/r6/call27_micro.log
