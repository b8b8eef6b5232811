#!/usr/bin/env python3
"""Cross-attention decode kernel: token-major K/V [B,S,H*64] vs head-major [B,H,S,64] (same kernel, different strides)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
dev = torch.device("cuda")
B, H, S, D = 32, 12, 1152, 768
Bkv = 16
q = torch.randn(B, D, device=dev).bfloat16()
k = torch.randn(Bkv, S, D, device=dev).bfloat16()
v = torch.randn(Bkv, S, D, device=dev).bfloat16()
kh = k.view(Bkv, S, H, 64).permute(0, 2, 1, 3).contiguous()      # [Bkv, H, S, 64]
vh = v.view(Bkv, S, H, 64).permute(0, 2, 1, 3).contiguous()
out = torch.empty(B, D, device=dev, dtype=torch.bfloat16)
ws = torch.empty(B * H * 8 * 66, device=dev)
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)

def run(kk, vv, k_bs, k_rs, k_hs, n=20):
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        LIB.call("cxr_attn_decode_bf16", q.data_ptr(), kk.data_ptr(), vv.data_ptr(), out.data_ptr(), None, q.stride(0), k_bs, k_rs, k_bs, k_rs,
                 out.stride(0), 0, B, H, S, 0.125, 2, ws.data_ptr(), int(k_hs), 0.0, None, 0, 0, ops._s())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

t0 = run(k, v, S * D, D, 64)
r0 = out.clone()
t1 = run(kh, vh, H * S * 64, 64, S * 64)
print("token-major %.1f us  head-major %.1f us  max diff %.4f  bytes %.1f MB" % (t0, t1, (out.float() - r0.float()).abs().max().item(), 2 * Bkv * S * D * 2 / 1e6))
