#!/usr/bin/env python3
"""LayerNorm backward variants at the model's shapes (GPU time only): with / without the parameter-gradient partials, with / without the residual add."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
BF = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.4e9 * 0.02))
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for rows, C in [(36928, 384), (8192, 768), (147456, 192), (589824, 64)]:
    nb = 12
    xs = [torch.randn(rows, C, device="cuda").to(BF) for _ in range(nb)]
    dys = [torch.randn(rows, C, device="cuda").to(BF) for _ in range(nb)]
    adds = [torch.randn(rows, C, device="cuda").to(BF) for _ in range(nb)]
    g = torch.randn(C, device="cuda")
    _, st = ops.layernorm(xs[0], g, g, 1e-5, need_stats=True)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    out = torch.empty_like(xs[0])
    i = [0]

    def run(with_dg, with_add):
        j = i[0] % nb; i[0] += 1
        ops.layernorm_bwd(xs[j], dys[j], g, st, dg if with_dg else None, db if with_dg else None, add=adds[j] if with_add else None, out=out)
    mb = rows * C * 2 / 1e6
    r = {k: timeit(lambda a=a, b=b: run(a, b)) for k, (a, b) in dict(full=(True, True), no_dgamma=(False, True), no_add=(True, False), bare=(False, False)).items()}
    fwd = timeit(lambda: ops.layernorm(xs[i[0] % nb], g, g, 1e-5))
    print(f"rows={rows:6d} C={C:4d} ({mb:5.1f} MB/tensor)  " + "  ".join(f"{k} {v:6.1f}us" for k, v in r.items()) + f"  | fwd {fwd:6.1f}us | full = {4*mb/r['full']/1e3:.2f} TB/s")
