#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
dev = torch.device("cuda")
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0)
m.graph_decode = False
images = torch.randn(16, 2, 3, 384, 384, device=dev)
prompt = torch.tensor([[8, 10, 9, 11, 1]] * 16, device=dev)
with torch.no_grad():
    eo = m.encoder(images)
    for _ in range(2):
        m.generate(encoder_outputs=eo, decoder_input_ids=prompt, special_token_ids=[9, 1, 3], max_length=256 + 5, bos_token_id=1, eos_token_id=None,
                   pad_token_id=4, mask_token_id=4, num_beams=1, use_cache=True)
torch.cuda.synchronize()
