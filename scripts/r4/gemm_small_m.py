#!/usr/bin/env python3
"""NT GEMM shapes of the SCST re-scoring pass and of the reward model (about 4096 rows), alone: python scripts/r4/gemm_small_m.py   (CXR_GEMM_STAGES / CXR_GEMM_BN select the kernel)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops
BF = torch.bfloat16
ops.gemm_exclusive(False)
shapes = [(4080, 768, 768, 0), (4080, 768, 768, 1), (4096, 2304, 768, 0), (4096, 3072, 768, 2), (4096, 768, 3072, 1), (2048, 768, 768, 1), (8192, 768, 768, 1)]
for M, N, K, mode in shapes:
    nb = 8
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
    Rs = [torch.randn(M, N, device="cuda").to(BF) for _ in range(nb)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    def run(j):
        if mode == 0: ops.gemm_nt(As[j], w, bias=bias, out=Cs[j])
        elif mode == 1: ops.gemm_nt(As[j], w, bias=bias, residual=Rs[j], out=Cs[j])
        else: ops.gemm_nt(As[j], w, bias=bias, act=1, out=Cs[j])
    for j in range(3): run(j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 64
    e0.record()
    for i in range(n): run(i % nb)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{M:6d} {N:5d} {K:5d} mode {mode}  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")
