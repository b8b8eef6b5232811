#!/usr/bin/env python3
"""128x128 against 128x64 tiles of gemm_nt_kernel for problems of 160 .. 300 128-wide tiles (run twice: default and CXR_GEMM_BN=64; the tile kernels only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
ops.gemm_exclusive(False)
BF = torch.bfloat16
for N, K in ((384, 384), (768, 768), (384, 1536), (1536, 384)):
    for tiles in (128, 162, 192, 219, 240, 255, 288, 320):
        tm = max(1, tiles // ((N + 127) // 128)); M = tm * 128
        nb = 8
        As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
        Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
        w = (torch.randn(N, K, device="cuda") * 0.05).to(BF); bias = torch.randn(N, device="cuda")
        for j in range(3): ops.gemm_nt(As[j], w, bias=bias, out=Cs[j])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(64): ops.gemm_nt(As[i % nb], w, bias=bias, out=Cs[i % nb])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 64
        print(f"N {N:5d} K {K:5d} M {M:6d} tiles128 {tm * ((N + 127) // 128):4d}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
