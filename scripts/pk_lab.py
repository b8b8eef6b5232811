#!/usr/bin/env python3
"""Persistent big-tile GEMM (csrc/gemm_pk.hip) against gemm_nt_kernel: bit-exact comparison on ragged / epilogue cases, then cache-cold timing
of the TF step's shapes with both kernels and both tile widths: python scripts/pk_lab.py [check|time]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def cfg(enabled, bn=0, min_rows=2048, wgs=256):
    LIB.call("cxr_gemm_pk_config", enabled, bn, min_rows, wgs)


def check():
    torch.manual_seed(0)
    bad = 0
    cases = [(36928, 384, 384), (9280, 384, 384), (4000, 392, 128), (2049, 64, 64), (8192, 768, 768), (5000, 1000, 192), (147456, 192, 192), (8192, 3072, 768),
             (3000, 30000, 128), (2500, 200, 64), (300, 384, 384), (70, 136, 64)]
    for M, N, K in cases:
        a = torch.randn(M + 3, K + 8, device="cuda").to(BF)[:M, :K]          # row stride != K
        w = (torch.randn(N, K, device="cuda") * 0.1).to(BF)
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda").to(BF)
        seed = torch.tensor([1234], dtype=torch.int32, device="cuda")
        rs = torch.rand(max(1, (M + 576) // 577), device="cuda") * 2
        variants = {
            "plain": dict(),
            "bias": dict(bias=bias),
            "bias+res": dict(bias=bias, residual=res),
            "gelu+save": dict(bias=bias, act=1, aux="new"),
            "gelu'": dict(act=2, aux=res),
            "f32": dict(bias=bias, out_f32=True),
            "f32acc": dict(out_f32=True, accumulate=True, out="rand"),
            "drop+res": dict(bias=bias, residual=res, drop=(0.1, seed, 7, 256, 3)),
            "droppath": dict(bias=bias, residual=res, row_scale=(rs, 577, True)),
            "alpha": dict(alpha=0.37, bias=bias),
        }
        for name, kw in variants.items():
            outs = []
            for mode in ((0, 0), (1, 1000), (1, 1001), (1, 1002), (1, 1003), (1, 1004), (1, 1005)):
                cfg(mode[0], mode[1], 1)
                k2 = dict(kw)
                aux = None
                if k2.get("aux") == "new":
                    aux = k2["aux"] = torch.zeros(M, N, device="cuda", dtype=BF)
                if k2.get("out") == "rand":
                    torch.manual_seed(5)
                    k2["out"] = torch.randn(M, N, device="cuda")
                o = ops.gemm_nt(a, w, **k2)
                outs.append((o, aux))
            torch.cuda.synchronize()
            for i in range(1, len(outs)):
                same = torch.equal(outs[0][0], outs[i][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[i][1]))
                if not same:
                    bad += 1
                    d = (outs[0][0].float() - outs[i][0].float()).abs()
                    print(f"MISMATCH M={M} N={N} K={K} {name} cfg={i - 1}: max|d|={float(d.max()):.4g} n_bad={int((d > 0).sum())} first_bad_row={int((d > 0).any(1).nonzero()[0])}")
        print(f"checked {M}x{N}x{K}", flush=True)
    print("CHECK", "FAILED" if bad else "OK", bad)
    return bad


def timing():
    shapes = [(4096, 4096, 4096, 0), (8192, 8192, 8192, 0),
              (36928, 384, 384, 0), (36928, 384, 384, 1), (36928, 1536, 384, 2), (36928, 1536, 384, 3), (36928, 384, 1536, 1), (9280, 384, 384, 0),
              (147456, 192, 192, 0), (147456, 768, 192, 2), (147456, 192, 768, 1), (589824, 64, 64, 0), (589824, 256, 64, 2), (589824, 64, 256, 1),
              (8192, 768, 768, 1), (8192, 2304, 768, 0), (8192, 3072, 768, 3), (8192, 768, 3072, 1), (36864, 9216, 768, 0), (8192, 30000, 768, 0),
              (8192, 768, 30016, 0), (36928, 768, 384, 0), (36928, 384, 1728, 0), (147456, 192, 576, 0)]
    mode_name = {0: "bias", 1: "bias+residual", 2: "bias+gelu", 3: "bias+gelu+save"}
    print(f"{'M':>7} {'N':>6} {'K':>6} {'epilogue':>15} | {'old us':>7} {'TF/s':>6} | " + " | ".join(f"c{i} us  TF/s" for i in range(6)) + " | best GB/s")
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
        res = []
        for en, bn in ((0, 0), (1, 1000), (1, 1001), (1, 1002), (1, 1003), (1, 1004), (1, 1005)):
            cfg(en, bn, 1)
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
        byt = 2.0 * (M * K + N * K) + 2.0 * M * N * (2 if mode in (1, 3) else 1)
        fl = 2.0 * M * N * K
        print(f"{M:7d} {N:6d} {K:6d} {mode_name[mode]:>15} | {res[0]:7.1f} {fl/res[0]/1e6:6.0f} | " + " | ".join(f"{r:6.1f} {fl/r/1e6:5.0f}" for r in res[1:])
              + f" | {byt/min(res[1:])/1e3:5.0f}", flush=True)
        del As, Cs, Rs


if what in ("check", "all"):
    if check() and what == "all":
        sys.exit(1)
if what in ("time", "all"):
    timing()
