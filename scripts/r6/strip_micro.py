#!/usr/bin/env python3
"""Row-strip GEMM (csrc/gemm_strip.hip) against the tiled / persistent kernels on the M x 384 x K shapes of CvT stage 3, cache-cold (operands rotate through
1.2 GB), epilogues as in the model: plain (dX products), bias + residual + DropPath factor (attention output / FFN down projections)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16


def timeit(fn, n=24):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def route(name):
    LIB.call("cxr_gemm_strip_config", 0, 0, -2, 0); LIB.call("cxr_gemm_set_exclusive", 1)
    if name == "tiled":
        LIB.call("cxr_gemm_set_exclusive", 0)
    elif name.startswith("strip"):
        _, mt, st = name.split(":")
        LIB.call("cxr_gemm_set_exclusive", 0); LIB.call("cxr_gemm_strip_config", 1, int(mt), 1, int(st))


N192 = os.environ.get("STRIP_N192") == "1"
NC = 192 if N192 else 384
SHAPES = [(147456, 768), (147456, 192), (36864, 192)] if N192 else [(36928, 1536), (36928, 384)] if os.environ.get("STRIP_QUICK") else [(36928, 1536), (36928, 384), (9280, 384), (18464, 1536), (18464, 384)]
ROUTES = ("tiled", "persistent", "strip:8:0", "strip:12:0", "strip:16:0") if N192 else ("tiled", "persistent", "strip:10:0", "strip:10:3") if os.environ.get("STRIP_QUICK") else ("tiled", "persistent", "strip:10:0", "strip:10:2", "strip:6:0", "strip:6:2", "strip:4:0", "strip:4:2", "strip:2:0", "strip:2:2")
for M, K in SHAPES:
    nb = max(2, int(1.2e9 / (M * K * 2)))
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    w = (torch.randn(NC, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(NC, device="cuda"); res = torch.randn(M, NC, device="cuda").to(BF); rs = torch.rand((M + 576) // 577, device="cuda")
    out = torch.empty(M, NC, device="cuda", dtype=BF)
    i = [0]
    for epi, kw in (("plain", {}), ("bias+res+droppath", dict(bias=bias, residual=res, row_scale=(rs, 577, True)))):
        row = []
        for r in ROUTES:
            route(r)

            def fn():
                i[0] += 1
                ops.gemm_nt(As[i[0] % nb], w, out=out, **kw)
            us = timeit(fn)
            row.append(f"{r} {us:6.1f}")
        gf = 2.0 * M * NC * K
        best = min(float(x.split()[-1]) for x in row)
        print(f"{M:6d} x {NC} x {K:4d} {epi:18s} " + " | ".join(row) + f"  || best {gf / best / 1e6:6.0f} TF/s")
route("default")
LIB.call("cxr_gemm_strip_config", 1, 0, -2, 0)
