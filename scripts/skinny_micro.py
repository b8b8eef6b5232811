#!/usr/bin/env python3
"""Latency of the decode GEMM variants (M = 32 rows) measured back to back inside one hipGraph (as in the decode loop)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
dev = torch.device("cuda")
M = 32
x = torch.randn(M, 768, device=dev).bfloat16()
x3 = torch.randn(M, 3072, device=dev).bfloat16()
res = torch.randn(M, 768, device=dev).bfloat16()
g, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
stats = torch.zeros(M, 2, device=dev)
seed = torch.full((1,), 3, dtype=torch.int32, device=dev)
W = lambda n, k: (torch.randn(n, k, device=dev) * 0.02).bfloat16()
w768, w3072, w2, wv = W(768, 768), W(3072, 768), W(768, 3072), W(30000, 768)
bias768, bias3072, biasv = torch.zeros(768, device=dev), torch.zeros(3072, device=dev), torch.zeros(30000, device=dev)
ws = [W(768, 768) for _ in range(64)]          # distinct weights per launch: no cache reuse between repetitions

def bench(name, fn, reps=64):
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(gph):
            for i in range(reps):
                fn(i)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    print(f"{name:42s} {min(ts):6.2f} us per launch")

out = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
bench("768x768 plain", lambda i: ops.gemm_skinny(x, ws[i % 64], bias=bias768, out=out))
bench("768x768 +residual", lambda i: ops.gemm_skinny(x, ws[i % 64], bias=bias768, residual=res, out=out))
bench("768x768 +residual +dropout", lambda i: ops.gemm_skinny(x, ws[i % 64], bias=bias768, residual=res, out=out, drop=(0.1, seed, 5, 9)))
bench("768x768 ln_a", lambda i: ops.gemm_skinny(x, ws[i % 64], bias=bias768, out=out, ln_a=(g, b, 1e-12, stats)))
bench("768x768 ln_r + dropout", lambda i: ops.gemm_skinny(x, ws[i % 64], bias=bias768, residual=res, out=out, ln_r=(stats, g, b), drop=(0.1, seed, 5, 9)))
o3 = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
bench("3072x768 gelu ln_a", lambda i: ops.gemm_skinny(x, w3072, bias=bias3072, act=1, out=o3, ln_a=(g, b, 1e-12, stats)))
bench("768x3072 +residual", lambda i: ops.gemm_skinny(x3, w2, bias=bias768, residual=res, out=out))
q, k, v = (torch.empty(M, 768, device=dev, dtype=torch.bfloat16) for _ in range(3))
bench("qkv grouped (3 x 768x768)", lambda i: ops.gemm_skinny3(x, ws[i % 64], bias768, q, ws[(i + 1) % 64], bias768, k, ws[(i + 2) % 64], bias768, v))
ol = torch.empty(M, 30000, device=dev)
bench("30000x768 f32 (LM head)", lambda i: ops.gemm_skinny(x, wv, bias=biasv, out=ol, out_f32=True), reps=8)
y = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
bench("layernorm 32x768", lambda i: ops.layernorm(x, g, b, 1e-12, out=y))
bench("empty-ish kernel (increment)", lambda i: ops.increment_(seed))
