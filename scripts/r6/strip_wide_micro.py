#!/usr/bin/env python3
"""(LAB RECORD: the column-sliced / GELU forms of the strip kernels this script timed were removed again -- profiles/r06_gemm_strip.txt, REVERSAL; on the
shipped library its 'strip slices' route falls through to the tiled kernel.)
Round 6: the FFN-up input-gradient product of CvT stage 3 (36928 x 1536 x 384, multiplied by GELU'(saved pre-activation)) alone on the chip, cache-cold
(operands rotate through > 1 GB): tiled kernel, W-stationary kernel (forced: it declines second-operand products by default), column-sliced row strips."""
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
    LIB.call("cxr_gemm_strip_config", 0, 0, -2, 0); LIB.call("cxr_gemm_set_exclusive", 0); LIB.call("cxr_gemm_ws_config", 1, 0, 2048, -1, 0)
    if name == "w-stationary":
        LIB.call("cxr_gemm_set_exclusive", 1); LIB.call("cxr_gemm_ws_config", 1, 1, 1, -1, 0)
    elif name == "strip slices":
        LIB.call("cxr_gemm_strip_config", 1, 0, -2, 0)


for M, N, K in ((36928, 1536, 384), (36928, 768, 384)):
    nb = 6
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    aux = [torch.randn(M, N, device="cuda").to(BF) for _ in range(nb)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    i = [0]
    for epi in ("gelu'", "plain"):
        row = []
        for r in ("tiled", "w-stationary", "strip slices"):
            route(r)

            def fn():
                i[0] += 1
                if epi == "plain":
                    ops.gemm_nt(As[i[0] % nb], w, out=out)
                else:
                    ops.gemm_nt(As[i[0] % nb], w, out=out, act=2, aux=aux[i[0] % nb])
            row.append(f"{r} {timeit(fn):6.1f} us")
        print(f"{M} x {N} x {K} {epi:6s} " + " | ".join(row), flush=True)
LIB.call("cxr_gemm_strip_config", 1, 0, -2, 0); LIB.call("cxr_gemm_set_exclusive", 1); LIB.call("cxr_gemm_ws_config", 1, 0, 2048, -1, 0)
