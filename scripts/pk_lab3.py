#!/usr/bin/env python3
"""Partition / walk-order variants of the persistent GEMM on the HBM-bound shapes (cache-cold), each checked against gemm_nt_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16
shapes = [(36928, 384, 384, 0), (36928, 384, 384, 1), (36928, 1536, 384, 2), (36928, 384, 1536, 1), (147456, 192, 192, 0), (147456, 768, 192, 2), (147456, 192, 768, 1),
          (589824, 64, 64, 0), (589824, 256, 64, 2), (589824, 64, 256, 1), (8192, 768, 768, 1), (8192, 3072, 768, 2), (8192, 768, 3072, 1)]
variants = [("old", 0, 0, 0), ("c0", 1, 1000, 0), ("c0+rev", 1, 1000, 16), ("c0+own", 1, 1000, 8), ("c0+own+rev", 1, 1000, 24), ("c6", 1, 1006, 0), ("c6+rev", 1, 1006, 16),
            ("c6+own", 1, 1006, 8), ("c6+own+rev", 1, 1006, 24), ("c1+own", 1, 1001, 8)]
print(f"{'M':>7} {'N':>5} {'K':>5} m | " + " | ".join(f"{v[0]:>10}" for v in variants))
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
    line = f"{M:7d} {N:5d} {K:5d} {mode} |"
    ref = None
    for name, en, cfgid, dbg in variants:
        if cfgid == 1001 and mode == 1:
            line += f" {'-':>10} |"; continue
        LIB.call("cxr_gemm_pk_config", en, cfgid, 1, -100 - dbg)
        Cs[0].zero_(); run(0); torch.cuda.synchronize()
        if ref is None: ref = Cs[0].clone()
        ok = torch.equal(ref, Cs[0])
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
        line += f" {best:8.1f}{' ' if ok else '!'}  |"
    print(line, flush=True)
    del As, Cs, Rs
