#!/usr/bin/env python3
"""(The N > 384 cases exercised the column-sliced strip form while it existed; on the shipped library they compare the tiled kernel with itself.)
Random-shape checks of the round-6 GEMM kernels: row-strip NT GEMM (N = 384 / 192 and the column-sliced form N = 768 / 1152 / 1536; every epilogue incl. GELU
and GELU') bit-identical to the tiled kernel, and the 384 x 192-block weight-gradient kernel (gemm_tn5_kernel) against fp32 torch + run-to-run identical.
python3 scripts/r6/fuzz_r6.py   (FUZZ_SEED=n for another sequence)"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
seed = int(os.environ.get("FUZZ_SEED", "0")); torch.manual_seed(seed); random.seed(seed)
BF = torch.bfloat16


def route(strip):
    LIB.call("cxr_gemm_set_exclusive", 0)
    LIB.call("cxr_gemm_strip_config", 1 if strip else 0, 0, 1 if strip else -2, 0)      # strip: every row count (min_rows 1), automatic strip height


bad = 0
for it in range(70):
    N = random.choice([384, 384, 192, 768, 1152, 1536])
    M = random.choice([1, 15, 16, 17, 159, 160, 161, 191, 193, 1000, 2561, 5003, 9280, 20001, 36928, 40000])
    K = 64 * random.choice([1, 2, 3, 6, 8, 12, 24])
    if N > 384 and K > 1536: K = 1536
    lda = K + random.choice([0, 8, 64])
    a = torch.randn(M, lda, device="cuda").to(BF)[:, :K]
    w = (torch.randn(N, K, device="cuda") * 0.1).to(BF)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(BF)
    rs = torch.rand((M + 576) // 577, device="cuda") * 2
    epi = random.choice(["plain", "bias", "bias+res", "gelu+saved", "gelu", "gelu'", "dp after", "dp before", "alpha"])
    kw = {"plain": {}, "bias": dict(bias=bias), "bias+res": dict(bias=bias, residual=res), "gelu+saved": dict(bias=bias, act=1, aux="new"), "gelu": dict(bias=bias, act=1),
          "gelu'": dict(act=2, aux=res), "dp after": dict(bias=bias, residual=res, row_scale=(rs, 577, True)), "dp before": dict(bias=bias, residual=res, row_scale=(rs, 577, False)),
          "alpha": dict(alpha=0.37, bias=bias)}[epi]
    outs = []
    for strip in (False, True):
        route(strip)
        k2 = dict(kw); aux = None
        if k2.get("aux") == "new":
            aux = k2["aux"] = torch.zeros(M, N, device="cuda", dtype=BF)
        outs.append((ops.gemm_nt(a, w, **k2), aux))
    same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    ok = bool(torch.isfinite(outs[1][0].float()).all())
    if epi == "plain":
        ref = a.float() @ w.float().t()
        ok = ok and ((outs[1][0].float() - ref).abs().max() <= 1e-2 * ref.abs().max() + 1e-2).item()
    if not (same and ok):
        bad += 1; print("BAD strip", M, N, K, lda, epi, same, ok)
LIB.call("cxr_gemm_strip_config", 1, 0, -2, 0); LIB.call("cxr_gemm_set_exclusive", 1)
print("row-strip fuzz done, bad =", bad, flush=True)
bad = 0
for it in range(30):
    I = 384 * random.choice([1, 1, 2, 3, 4]); J = 192 * random.choice([1, 2, 3, 5, 7, 8]); R = random.choice([4097, 8192, 9280, 20000, 36928, 50001])
    ldp, ldq = I + random.choice([0, 8]), J + random.choice([0, 16])
    p = torch.randn(R, ldp, device="cuda").to(BF)[:, :I]; q = torch.randn(R, ldq, device="cuda").to(BF)[:, :J]
    out = torch.zeros(I, J, device="cuda"); db = torch.zeros(I, device="cuda")
    ops.gemm_tn(p, q, out, dbias=db)
    ref = p.float().t() @ q.float()
    e = ((out - ref).abs().max() / ref.abs().max()).item(); eb = ((db - p.float().sum(0)).abs().max() / (p.float().sum(0).abs().max() + 1e-6)).item()
    out2 = torch.zeros(I, J, device="cuda"); db2 = torch.zeros(I, device="cuda"); ops.gemm_tn(p, q, out2, dbias=db2)
    if e > 2e-3 or eb > 2e-3 or not torch.equal(out, out2) or not torch.equal(db, db2):
        bad += 1; print("BAD tn5", R, I, J, e, eb, torch.equal(out, out2))
print("384 x 192-block weight-gradient fuzz done, bad =", bad, flush=True)
