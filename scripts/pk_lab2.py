#!/usr/bin/env python3
"""Where the persistent GEMM's time goes: timing with parts switched off (CXR_PK_DEBUG bits) on a few shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16
shapes = [(36928, 384, 384, 0, 128), (36928, 384, 384, 1, 128), (36928, 1536, 384, 2, 128), (36928, 1536, 384, 2, 256), (36928, 384, 1536, 1, 128), (589824, 64, 64, 0, 128), (8192, 3072, 768, 2, 256)]
for M, N, K, mode, bn in shapes:
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
    line = f"{M:7d} {N:5d} {K:5d} mode {mode} bn {bn}:"
    for dbg, name in ((0, "full"), (2, "no-stores"), (1, "no-epilogue")):
        LIB.call("cxr_gemm_pk_config", 1, bn, 1, -100 - dbg)
        for j in range(min(nb, 3)): run(j)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(2 * nb, 12)
            e0.record()
            for i in range(n): run(i % nb)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        line += f"  {name} {best:6.1f}"
    print(line, flush=True)
    del As, Cs, Rs
