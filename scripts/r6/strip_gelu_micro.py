#!/usr/bin/env python3
"""(LAB RECORD: the column-sliced / GELU forms of the strip kernels this script timed were removed again -- profiles/r06_gemm_strip.txt, REVERSAL; on the
shipped library its 'strip slices' route falls through to the tiled kernel.)
Round 6: FFN-up products with the GELU epilogue + saved pre-activation (forward, train mode) alone on the chip, cache-cold: tiled kernel, what the
forward uses today (persistent / W-stationary), column-sliced row strips."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def route(name):
    LIB.call("cxr_gemm_strip_config", 0, 0, -2, 0); LIB.call("cxr_gemm_set_exclusive", 0)
    if name == "forward default":
        LIB.call("cxr_gemm_set_exclusive", 1)
    elif name == "strip slices":
        LIB.call("cxr_gemm_strip_config", 1, 0, -2, 0)


for M, N, K in ((147456, 768, 192), (36928, 1536, 384), (147456, 576, 192)):
    nb = 4
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=BF); aux = torch.empty(M, N, device="cuda", dtype=BF)
    i = [0]
    for epi, kw in (("gelu+saved", dict(bias=bias, act=1, aux=aux)), ("gelu", dict(bias=bias, act=1)), ("bias", dict(bias=bias))):
        row = []
        for r in ("tiled", "forward default", "strip slices"):
            route(r)

            def fn():
                i[0] += 1
                ops.gemm_nt(As[i[0] % nb], w, out=out, **kw)
            row.append(f"{r} {timeit(fn):6.1f} us")
        print(f"{M} x {N} x {K} {epi:10s} " + " | ".join(row), flush=True)
LIB.call("cxr_gemm_strip_config", 1, 0, -2, 0); LIB.call("cxr_gemm_set_exclusive", 1)
