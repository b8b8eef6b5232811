#!/usr/bin/env python3
"""W-stationary GEMM (csrc/gemm_ws.hip) against gemm_nt_kernel: bit-exact check, then cache-cold timing of the K = 384 shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16


def cfg(ws, bm=0, dbg=0, wgs=256):
    LIB.call("cxr_gemm_ws_config", ws, bm, 1, wgs, dbg)
    LIB.call("cxr_gemm_pk_config", 0, 0, -1, -1)


def check():
    bad = 0
    for M, N in [(36928, 384), (9280, 384), (36928, 1536), (1000, 384), (70, 768), (36864, 768), (4999, 1536)]:
        K = 384
        torch.manual_seed(M)
        a = torch.randn(M + 3, K + 8, device="cuda").to(BF)[:M, :K]
        w = (torch.randn(N, K, device="cuda") * 0.1).to(BF)
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda").to(BF)
        rs = torch.rand(max(1, (M + 576) // 577), device="cuda") * 2
        variants = {"plain": dict(), "bias": dict(bias=bias), "bias+res": dict(bias=bias, residual=res), "gelu+save": dict(bias=bias, act=1, aux="new"),
                    "gelu": dict(bias=bias, act=1), "gelu'": dict(act=2, aux=res), "droppath": dict(bias=bias, residual=res, row_scale=(rs, 577, True)),
                    "droppath-before": dict(bias=bias, residual=res, row_scale=(rs, 577, False)), "alpha": dict(alpha=0.37, bias=bias)}
        for name, kw in variants.items():
            outs = []
            rop = ("residual" in kw) or kw.get("act") == 2
            for ws, bm in ((0, 0), (1, 32)):
                cfg(ws, bm)
                k2 = dict(kw)
                aux = None
                if k2.get("aux") == "new":
                    aux = k2["aux"] = torch.zeros(M, N, device="cuda", dtype=BF)
                o = ops.gemm_nt(a, w, **k2)
                outs.append((o, aux))
            torch.cuda.synchronize()
            for i in range(1, len(outs)):
                same = torch.equal(outs[0][0], outs[i][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[i][1]))
                if not same:
                    bad += 1
                    d = (outs[0][0].float() - outs[i][0].float()).abs()
                    print(f"MISMATCH M={M} N={N} {name} variant {i}: max|d|={float(d.max()):.4g} n_bad={int((d > 0).sum())} rows {(d > 0).any(1).nonzero()[:4].flatten().tolist()} cols {(d > 0).any(0).nonzero()[:6].flatten().tolist()}")
        print(f"checked {M}x{N}", flush=True)
    print("CHECK", "FAILED" if bad else "OK", bad)
    return bad


def timing():
    shapes = [(36928, 384, 0), (36928, 384, 1), (36928, 1536, 2), (36928, 1536, 3), (36928, 1536, 4), (9280, 384, 0), (36864, 768, 0)]
    mode_name = {0: "bias", 1: "bias+residual", 2: "bias+gelu", 3: "bias+gelu+save", 4: "gelu'(aux)"}
    K = 384
    print(f"{'M':>7} {'N':>6} {'epilogue':>15} | {'old us':>7} {'TF/s':>6} | ws   us TF/s  GB/s | ws+prio  TF/s  GB/s | ws no-epi | ws no-stores")
    for M, N, mode in shapes:
        two = mode in (1, 3, 4)
        per = (M * K + M * N * (2 if two else 1)) * 2
        nb = max(2, min(12, int(700e6 // per)))
        As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
        Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
        Rs = [torch.randn(M, N, device="cuda").to(BF) for _ in range(nb)] if two else None
        w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda")

        def run(j):
            if mode == 0: ops.gemm_nt(As[j], w, bias=bias, out=Cs[j])
            elif mode == 1: ops.gemm_nt(As[j], w, bias=bias, residual=Rs[j], out=Cs[j])
            elif mode == 2: ops.gemm_nt(As[j], w, bias=bias, act=1, out=Cs[j])
            elif mode == 3: ops.gemm_nt(As[j], w, bias=bias, act=1, aux=Rs[j], out=Cs[j])
            else: ops.gemm_nt(As[j], w, act=2, aux=Rs[j], out=Cs[j])
        res = []
        for ws, bm, dbg in ((0, 0, 0), (1, 32, 0), (1, 32, 8), (1, 32, 1), (1, 32, 2)):
            cfg(ws, bm, dbg)
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
            res.append(best)
        byt = 2.0 * (M * K + N * K) + 2.0 * M * N * (2 if two else 1)
        fl = 2.0 * M * N * K
        print(f"{M:7d} {N:6d} {mode_name[mode]:>15} | {res[0]:7.1f} {fl/res[0]/1e6:6.0f} | {res[1]:6.1f} {fl/res[1]/1e6:5.0f} {byt/res[1]/1e3:5.0f} | {res[2]:6.1f} {fl/res[2]/1e6:5.0f} {byt/res[2]/1e3:5.0f} | {res[3]:6.1f} | {res[4]:6.1f}", flush=True)
        del As, Cs, Rs


bad = check()
timing()
