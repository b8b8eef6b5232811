#!/usr/bin/env python3
"""Round 6 lab: what would the weight-streaming launches of the cached decode step cost if their weights were already in the XCD-local L2 / in the
Infinity Cache when they start?  A dependent chain ffn1 (768 -> 3072, LayerNorm folded, GELU) -> ffn2 (3072 -> 768, residual, statistics out) of
the decode-step GEMM kernels (csrc/decode_gemm.hip) at 32 rows, replayed from one hipGraph, with the chain walking over P distinct weight pairs
(9.4 MB per pair): P = 1 -> every launch finds its weights in L2; P = 12 -> 113 MB, Infinity-Cache resident; P = 40 -> 377 MB, HBM (what the
decode step sees: 580 MB per token-step).  python3 scripts/r6/dec_warm_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops

dev = "cuda"
M, D, F, eps = 32, 768, 3072, 1e-12
torch.manual_seed(0)
NPAIR_MAX, CHAIN = 40, 40


def packs(n):
    out = []
    for i in range(n):
        w1 = (torch.randn(F, D, device=dev) * 0.03).to(torch.bfloat16)
        w2 = (torch.randn(D, F, device=dev) * 0.02).to(torch.bfloat16)
        g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
        p1 = ops.dec_pack_weight(w1, g, b, torch.zeros(F, device=dev))
        p2 = ops.dec_pack_weight(w2, None, None, torch.zeros(D, device=dev))
        out.append((p1, p2))
    return out


ALL = packs(NPAIR_MAX)
rgb = torch.stack([torch.ones(D, device=dev), torch.zeros(D, device=dev)], 1).contiguous()
x = (torch.randn(M, D, device=dev)).to(torch.bfloat16)
x0, st0 = ops.dec_to_dal(x, want_stats=True)


def chain(npair):
    cur, st = x0, st0
    for i in range(CHAIN):
        p1, p2 = ALL[i % npair]
        (f,), _ = ops.dec_gemm(cur, M, D, [dict(wp=p1[0], bc=p1[1], N=F, fold=True)], act=1, stats=st, eps=eps)
        (a,), st = ops.dec_gemm(f, M, F, [dict(wp=p2[0], bc=p2[1], N=D)], out_stats=True, residual=cur, stats=st, rgb=rgb, eps=eps)
        cur = a
    return cur


for npair in (1, 2, 4, 12, 40, 1, 40):
    chain(npair); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g) if hasattr(ops, "graph_capture") else torch.cuda.graph(g):
        out = chain(npair)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (2 * CHAIN))
    ts.sort()
    print(f"distinct weight pairs {npair:3d} ({npair * 9.4:6.1f} MB)  per launch: median {ts[len(ts) // 2]:6.2f} us  min {ts[0]:6.2f} us", flush=True)
