#!/usr/bin/env python3
"""NT GEMM with cache-warm operands (same buffers every call) vs cache-cold operands (rotating through > 512 MB of buffers, as inside a
training step where every activation was written tens of kernels earlier)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
BF = torch.bfloat16
for M, N, K in [(8192, 768, 768), (18464, 384, 384), (18464, 1536, 384), (18464, 384, 1536), (8192, 3072, 768), (8192, 768, 3072)]:
    per = (M * K + M * N) * 2
    nb = max(2, int(600e6 // per))
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    res = {}
    for mode in ("warm", "cold"):
        for _ in range(3): ops.gemm_nt(As[0], w, bias=bias, out=Cs[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 4 * nb
        torch.cuda._sleep(int(2.4e9 * 0.02))
        e0.record()
        for i in range(n):
            j = i % nb if mode == "cold" else 0
            ops.gemm_nt(As[j], w, bias=bias, out=Cs[j])
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}  warm {res['warm']:7.1f} us {fl/res['warm']/1e6:6.0f} TF/s   cold {res['cold']:7.1f} us {fl/res['cold']/1e6:6.0f} TF/s   ({nb} buffers)")
    del As, Cs
