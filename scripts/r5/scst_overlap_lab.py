#!/usr/bin/env python3
"""Lab (round 5): the NEXT step's frozen-encoder forward beside the CURRENT step's cached decode. The decode is latency-bound (48 dependent ~7-us kernels per
token-step, 0.22 of the HBM roofline): does an encoder forward of the next batch (6 ms alone) fit into its idle resources? Encoder on a CU-masked stream
(k CUs in every XCD, ops.masked_stream), decode on the main stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import LongitudinalPromptMultiCXREncoderDecoderModel

dev = torch.device("cuda")
B, N = 16, 2
m = LongitudinalPromptMultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).train()
g = torch.Generator().manual_seed(0)
images = torch.randn(B, N, 3, 384, 384, generator=g).to(dev)
images2 = torch.randn(B, N, 3, 384, 384, generator=g).to(dev)
prompt = torch.tensor([[8, 10, 9, 11, 1]] * B, device=dev)

def decode(eo):
    return m.sample_and_greedy(eo, prompt, [1, 3], [9, 1, 3], 4, 256 + 5, 1, None, 4)

def mask_words(k):
    w = [0] * 8
    for b in range(256):
        if b // 8 < k:
            w[b // 32] |= 1 << (b % 32)
    return w

with torch.no_grad():
    eo = m.encoder(images)
    for _ in range(2):
        decode(eo); m.encoder(images2)
    torch.cuda.synchronize()

    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    t_enc = timed(lambda: m.encoder(images2))
    t_dec = timed(lambda: decode(eo))
    t_seq = timed(lambda: (m.encoder(images2), decode(eo)))
    print(f"encoder alone {t_enc:.2f} ms, decode alone {t_dec:.2f} ms, one after the other {t_seq:.2f} ms")
    for k in (4, 8, 12, 16, 32):
        side = ops.masked_stream(mask_words(k)) if k < 32 else torch.cuda.Stream()

        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m.encoder(images2)
            decode(eo)
            torch.cuda.current_stream().wait_stream(side)

        print(f"encoder on {'an unmasked second stream' if k == 32 else f'{k} CUs per XCD ({8 * k} CUs)'} beside the decode: {timed(both):.2f} ms")
