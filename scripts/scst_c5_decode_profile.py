#!/usr/bin/env python3
"""The two decodes of one configs[4]-shaped SCST step (16 studies x 3 images, 128-token prompt, sample + greedy as one 32-row batch, train mode) for
rocprofv3 (--kernel-trace --stats, or the --pmc FETCH_SIZE / WRITE_SIZE passes):  python3 scripts/scst_c5_decode_profile.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
dev = torch.device("cuda")
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).train()
g = torch.Generator().manual_seed(0)
images = torch.randn(16, 3, 3, 384, 384, generator=g).to(dev)
prompt = torch.cat([torch.full((16, 1), 8), torch.randint(12, 30000, (16, 62), generator=g), torch.full((16, 1), 9),
                    torch.randint(12, 30000, (16, 63), generator=g), torch.full((16, 1), 1)], 1).to(dev)      # as bench.py scst_c5: 128 prompt tokens
if os.environ.get("CXR_PROFILE_EAGER") == "1":      # counter (--pmc) passes: the same kernels launched one by one (hipGraph replays under --pmc take tens of minutes)
    m.graph_decode = False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
new_tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 255
with torch.no_grad():
    eo = m.encoder(images)
    for _ in range(reps):
        m.sample_and_greedy(eo, prompt, [1, 3], [9, 1, 3], 4, new_tokens + 1 + prompt.shape[1], 1, None, 4)
torch.cuda.synchronize()
print("prompt tokens", prompt.shape[1])
