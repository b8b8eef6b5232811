#!/usr/bin/env python3
"""Timing of single GEMM shapes of the 2-image TF step, cache-cold (rotating through > 600 MB of operands, as inside a training step where
every activation was written tens of kernels earlier): python scripts/gemm_micro.py [pk]   (CXR_GEMM_PK selects the kernel family)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
BF = torch.bfloat16
if os.environ.get("GEMM_MICRO_NOEXCL") == "1":
    ops.gemm_exclusive(False)      # the tile kernels only, as in the backward passes (no persistent one-workgroup-per-CU kernels)
shapes = [(4096, 4096, 4096, 0), (8192, 8192, 8192, 0),
          (36928, 384, 384, 0), (36928, 384, 384, 1), (36928, 1536, 384, 2), (36928, 1536, 384, 3), (36928, 384, 1536, 1), (9280, 384, 384, 0),
          (147456, 192, 192, 0), (147456, 768, 192, 2), (147456, 192, 768, 1), (589824, 64, 64, 0), (589824, 256, 64, 2), (589824, 64, 256, 1),
          (8192, 768, 768, 1), (8192, 2304, 768, 0), (8192, 3072, 768, 3), (8192, 768, 3072, 1), (36864, 9216, 768, 0), (8192, 30000, 768, 0),
          (8192, 768, 30016, 0), (36928, 768, 384, 0), (36928, 384, 1728, 0), (147456, 192, 576, 0)]
mode_name = {0: "bias", 1: "bias+residual", 2: "bias+gelu", 3: "bias+gelu+save"}
print(f"{'M':>7} {'N':>6} {'K':>6} {'epilogue':>15} {'us':>9} {'TF/s':>8} {'GB/s alg':>9}")
for M, N, K, mode in shapes:
    per = (M * K + M * N * (2 if mode in (1, 3) else 1)) * 2
    nb = max(2, min(12, int(700e6 // per)))
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
    Rs = [torch.randn(M, N, device="cuda").to(BF) for _ in range(nb)] if mode in (1, 3) else None
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")

    def run(j):
        if mode == 0: ops.gemm_nt(As[j], w, bias=bias, out=Cs[j])
        elif mode == 1: ops.gemm_nt(As[j], w, bias=bias, residual=Rs[j], out=Cs[j])
        elif mode == 2: ops.gemm_nt(As[j], w, bias=bias, act=1, out=Cs[j])
        else: ops.gemm_nt(As[j], w, bias=bias, act=1, aux=Rs[j], out=Cs[j])
    for j in range(min(nb, 3)): run(j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(2 * nb, 12)
    e0.record()
    for i in range(n): run(i % nb)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byt = 2.0 * (M * K + N * K) + 2.0 * M * N * (2 if mode in (1, 3) else 1)
    print(f"{M:7d} {N:6d} {K:6d} {mode_name[mode]:>15} {us:9.1f} {2.0*M*N*K/us/1e6:8.1f} {byt/us/1e3:9.0f}")
    del As, Cs, Rs
