#!/bin/bash
# round 6 call 11: hand-off laboratory (persistent-launch edge vs kernel boundary at this model's vector sizes), twice
mkdir -p gpurun_out/r6
for rep in 1 2; do timeout 300 scripts/lab/handoff_lab >> gpurun_out/r6/call11_handoff.log 2>&1; done
cat gpurun_out/r6/call11_handoff.log
