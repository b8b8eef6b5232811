#!/usr/bin/env python3
"""The two decodes of one SCST step (16 studies x 2 images, sample + greedy as one 32-row batch, train mode) for rocprofv3 --kernel-trace --stats:
   cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o scst_decode -- python3 scripts/scst_decode_profile.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
dev = torch.device("cuda")
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).train()
images = torch.randn(16, 2, 3, 384, 384, device=dev)
prompt = torch.tensor([[8, 10, 9, 11, 1]] * 16, device=dev)
if os.environ.get("CXR_PROFILE_EAGER") == "1":      # counter (--pmc) passes: the same kernels launched one by one (hipGraph replays under --pmc take tens of minutes)
    m.graph_decode = False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    eo = m.encoder(images)
    for _ in range(reps):
        m.sample_and_greedy(eo, prompt, [1, 3], [9, 1, 3], 4, 256 + 5, 1, None, 4)
torch.cuda.synchronize()
