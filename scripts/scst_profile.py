#!/usr/bin/env python3
"""Wall-clock breakdown of one SCST step (configs[3] per-GPU shape)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel
from cxrmate_amd.reward import CXRBERTReward

dev = torch.device("cuda")
cfg = EncoderDecoderConfig()
B, N = 16, 2
m = LongitudinalPromptMultiCXREncoderDecoderModel(cfg, device=dev, seed=0)
reward = CXRBERTReward(dev, seed=1)
g = torch.Generator().manual_seed(0)
images = torch.randn(B, N, 3, 384, 384, generator=g).to(dev)
prompt = torch.tensor([[8, 10, 9, 11, 1]] * B, device=dev)
ones = torch.ones(B, 128, dtype=torch.int64, device=dev)
lab = torch.randint(1000, 30000, (B, 128), generator=g).to(dev)

def T(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, out

with torch.no_grad():
    ms, eo = T(lambda: m.encoder(images)); print(f"encoder 32 images           {ms:8.1f} ms")
    for gd in (True, False):
        m.graph_decode = gd
        ms, s = T(lambda: m.generate(input_ids=prompt, special_token_ids=[1, 3], encoder_outputs=eo, bos_token_id=1, eos_token_id=None, pad_token_id=4,
                                     mask_token_id=4, do_sample=True, top_k=50, max_new_tokens=255, use_cache=True), 2)
        print(f"sample 255 tokens graph={gd}   {ms:8.1f} ms  ({ms/255:.3f} ms/token)")
        ms, s = T(lambda: m.generate(encoder_outputs=eo, decoder_input_ids=prompt, special_token_ids=[9, 1, 3], max_length=256 + 5, bos_token_id=1,
                                     eos_token_id=None, pad_token_id=4, mask_token_id=4, num_beams=1, use_cache=True), 2)
        print(f"greedy 255 tokens graph={gd}   {ms:8.1f} ms  ({ms/255:.3f} ms/token)")
    ms, _ = T(lambda: reward.reward_from_ids(lab, ones, lab, ones)); print(f"reward (2 BERT fwd, R=128)  {ms:8.1f} ms")
