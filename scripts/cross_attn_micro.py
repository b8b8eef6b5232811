"""Cached cross-attention step: VALU decode kernel vs the MFMA kernel (graph-replayed chains of 48 launches, as in the decode loop): python scripts/cross_attn_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops  # noqa: E402


def chain(fn, n=48, reps=9):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for Bkv, share, Tk in [(16, 2, 1152), (8, 4, 1152), (16, 2, 576), (16, 1, 1152)]:
    H, D, B = 12, 768, Bkv * share
    q = torch.randn(B, D, device="cuda").bfloat16()
    ks = [torch.randn(Bkv, Tk, D, device="cuda").bfloat16() for _ in range(6)]          # six layers' K / V: the chain streams 6 x 57 MB like a token-step
    vs = [torch.randn(Bkv, Tk, D, device="cuda").bfloat16() for _ in range(6)]
    pks = [ops.pack_cross_kv(k, v, H) for k, v in zip(ks, vs)]
    kpm = torch.ones(Bkv, Tk, dtype=torch.uint8, device="cuda")
    bits = ops.pack_mask_bits(kpm)
    seed = torch.tensor([5], dtype=torch.int32, device="cuda")
    dr = (0.1, seed, 3, 7)
    out = torch.empty(ops.dal_rows(B), D, device="cuda", dtype=torch.bfloat16)
    i = [0]

    def old():
        l = i[0] % 6; i[0] += 1
        ops.attention_decode(q, ks[l], vs[l], H, 0.125, kpm_bits=bits, drop=dr, out=out, out_dal=True)

    def new():
        l = i[0] % 6; i[0] += 1
        ops.attention_cross_mfma(q, pks[l], Bkv, Tk, H, 0.125, kpm_bits=bits, drop=dr, out=out, out_dal=True)

    print(f"Bkv={Bkv} share={share} Tk={Tk}: VALU kernel {chain(old):6.2f} us   MFMA kernel {chain(new):6.2f} us   ({2 * Bkv * Tk * D * 2 / 1e6:.0f} MB of K/V per launch)")
