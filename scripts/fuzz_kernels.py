"""Random-shape checks of the round-3 kernels against fp32 torch references (attention forward / backward in both kernel generations, weight-gradient GEMM incl.\nits determinism): python scripts/fuzz_kernels.py"""
import os, sys, torch, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
_SEED = int(os.environ.get("FUZZ_SEED", "0")); torch.manual_seed(_SEED); random.seed(_SEED)
BF = torch.bfloat16
def ref_attn(q, k, v, H, scale, kpm, causal, shift):
    B, Tq, D = q.shape; Tk = k.shape[1]
    qh = q.float().view(B, Tq, H, 64).transpose(1, 2); kh = k.float().view(B, Tk, H, 64).transpose(1, 2); vh = v.float().view(B, Tk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) * scale
    neg = torch.finfo(torch.float32).min
    if kpm is not None: s = s.masked_fill(~kpm.bool().view(B, 1, 1, Tk), neg)
    if causal:
        i = torch.arange(Tq, device=q.device).view(Tq, 1); j = torch.arange(Tk, device=q.device).view(1, Tk)
        s = s.masked_fill(j > i + shift, neg)
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Tq, D)
bad = 0
for it in range(60):
    B, H = random.choice([1, 2, 3]), random.choice([1, 2, 6, 12])
    Tq, Tk = random.choice([1, 7, 64, 65, 129, 255, 300, 577, 1030, 1500]), random.choice([1, 5, 64, 70, 128, 145, 200, 300, 576, 700])
    causal = random.random() < 0.3 and Tk >= Tq
    masked = random.random() < 0.5
    q, k, v = (torch.randn(B, T, H * 64, device="cuda").to(BF) for T in (Tq, Tk, Tk))
    kpm = None
    if masked:
        kpm = (torch.rand(B, Tk, device="cuda") < 0.7).to(torch.uint8)
        kpm[:, 0] = 1
        if random.random() < 0.5 and Tk > 128: kpm[0, 64:128] = 0          # a fully masked tile
    for ver in (1, 2):
        ops.attention_config(ver, ver)
        o, lse = ops.attention(q, k, v, H, 0.125, kpm=kpm, causal=causal, need_lse=True)
        r = ref_attn(q, k, v, H, 0.125, kpm, causal, Tk - Tq)
        err = (o.float() - r).abs().max().item()
        do = torch.randn_like(o)
        dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, kpm=kpm, causal=causal)
        qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
        ref_attn(qr, kr, vr, H, 0.125, kpm, causal, Tk - Tq).backward(do.float())
        eg = max((a.float() - b.grad).abs().max().item() / max(b.grad.abs().max().item(), 0.05) for a, b in ((dq, qr), (dk, kr), (dv, vr)))
        if err > 3e-2 or eg > 5e-2 or not torch.isfinite(o.float()).all():
            bad += 1; print("BAD attn", ver, B, H, Tq, Tk, causal, masked, err, eg)
ops.attention_config(2, 2)
print("attention fuzz done, bad =", bad)
bad = 0
for it in range(25):
    R = random.choice([4100, 9000, 20000, 40000]); I = random.choice([136, 192, 264, 384, 520, 776]); J = random.choice([136, 200, 256, 392, 768, 1032])
    p = torch.randn(R, I, device="cuda").to(BF); q = torch.randn(R, J, device="cuda").to(BF)
    out = torch.zeros(I, J, device="cuda"); db = torch.zeros(I, device="cuda")
    ops.gemm_tn(p, q, out, dbias=db)
    ref = p.float().t() @ q.float()
    e = (out - ref).abs().max().item() / ref.abs().max().item(); eb = (db - p.float().sum(0)).abs().max().item() / (p.float().sum(0).abs().max().item() + 1e-6)
    out2 = torch.zeros(I, J, device="cuda"); db2 = torch.zeros(I, device="cuda"); ops.gemm_tn(p, q, out2, dbias=db2)
    if e > 2e-3 or eb > 2e-3 or not torch.equal(out, out2) or not torch.equal(db, db2):
        bad += 1; print("BAD tn", R, I, J, e, eb, torch.equal(out, out2))
print("tn fuzz done, bad =", bad)
bad = 0
for it in range(60):
    G = random.choice([1, 2, 3, 4]); Bkv = random.choice([1, 2, 5, 8, 16]); Tk = 32 * random.choice([1, 2, 7, 12, 13, 24, 25, 36, 37, 54, 60])
    if Bkv * G > 64: continue
    B, H, D = Bkv * G, 12, 768
    raw = (torch.randn(B, D, device="cuda") * random.choice([0.5, 2.0]) + random.choice([0.0, 0.5])).to(BF)
    w, bias = (torch.randn(D, D, device="cuda") * 0.04).to(BF), torch.randn(D, device="cuda") * 0.1
    g_, b_ = 1 + 0.1 * torch.randn(D, device="cuda"), 0.1 * torch.randn(D, device="cuda")
    k = (torch.randn(Bkv, Tk, D, device="cuda") * 0.7).to(BF); v = torch.randn(Bkv, Tk, D, device="cuda").to(BF)
    kpm = None
    if random.random() < 0.6:
        kpm = (torch.rand(Bkv, Tk, device="cuda") < 0.7).to(torch.uint8); kpm[:, 0] = 1
    bits = ops.pack_mask_bits(kpm) if kpm is not None else None
    pk = ops.pack_cross_kv(k, v, H)
    x_dal, st = ops.dec_to_dal(raw, want_stats=True)
    wf, bcf = ops.dec_pack_weight(w, g_, b_, bias)
    one = ops.attention_cross_mfma_q(x_dal, B, st, 1e-12, wf, bcf, pk, Bkv, Tk, H, 0.125, kpm_bits=bits, out_dal=False)
    qf = torch.nn.functional.layer_norm(raw.float(), (D,), g_, b_, 1e-12) @ w.float().t() + bias
    kk, vv = k.float().repeat(G, 1, 1), v.float().repeat(G, 1, 1)
    qh, kh, vh = qf.view(B, H, 1, 64), kk.view(B, Tk, H, 64).transpose(1, 2), vv.view(B, Tk, H, 64).transpose(1, 2)
    s_ = (qh @ kh.transpose(2, 3)) * 0.125
    if kpm is not None: s_ = s_.masked_fill(~kpm.bool().repeat(G, 1).view(B, 1, 1, Tk), torch.finfo(torch.float32).min)
    full = (torch.softmax(s_, -1) @ vh).transpose(1, 2).reshape(B, D)
    e = (one.float() - full).abs().max().item() / max(full.abs().max().item(), 0.05)
    if e > 4e-2 or not torch.isfinite(one.float()).all():
        bad += 1; print("BAD cross-q", G, Bkv, Tk, kpm is not None, e)
print("fused cross-attention fuzz done, bad =", bad)
