#!/bin/bash
# GPU call 17 of round 4: SCST phase times with the per-problem (0) and the several-problems-per-launch (1) LoRA kernels, alternated
mkdir -p gpurun_out/r4
for rep in 1 2 3; do for v in 0 1; do
  echo "== CXR_LORA_MULTI=$v rep $rep"
  CXR_LORA_MULTI=$v timeout 300 python scripts/scst_breakdown.py 2>/dev/null | grep -E "sample|re-score"
done; done
