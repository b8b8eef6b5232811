#!/usr/bin/env python3
"""Encoder + decoder FORWARD only (bench.py `forward_only` workload) for rocprofv3 --kernel-trace --stats: python3 scripts/fwd_profile.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import MultiCXREncoderDecoderModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = EncoderDecoderConfig()
m = MultiCXREncoderDecoderModel(cfg, device="cuda", seed=0)
m.train()
px, inp, am, lab = bench.synth_batch(32, 256, 30000, "cuda", 77, 2)
tt = m.token_ids_to_token_type_ids(inp, [3])
with torch.no_grad():
    for _ in range(n):
        m(pixel_values=px, decoder_input_ids=inp, decoder_attention_mask=am, decoder_token_type_ids=tt).logits
torch.cuda.synchronize()
