#!/usr/bin/env python3
"""Back-to-back timing of single GEMM shapes: python scripts/gemm_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 768, 768), (8192, 768, 3072), (8192, 3072, 768), (18464, 384, 384), (18464, 1536, 384),
          (18464, 384, 1536), (4640, 384, 384), (73728, 192, 192), (8192, 30000, 768), (32768, 768, 768)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for mode in ("plain", "bias+gelu"):
        kw = dict(bias=bias, act=1) if mode != "plain" else {}
        for _ in range(3): ops.gemm_nt(a, w, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): ops.gemm_nt(a, w, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{M:6d} {N:6d} {K:6d} {mode:10s} {us:9.1f} us {2.0*M*N*K/us/1e6:8.1f} TF/s  blocks {((M+127)//128)*((N+127)//128)}")
