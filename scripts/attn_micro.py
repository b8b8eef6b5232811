"""Forward / backward attention kernels alone on the TF step's shapes (2-image batch: 64 images / 32 studies), both kernel generations:
python scripts/attn_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops  # noqa: E402


def bench(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = {"cvt stage 1": (64, 1, 9216, 2304, False, False, 1), "cvt stage 2": (64, 3, 2304, 576, False, False, 4),
          "cvt stage 3": (64, 6, 577, 145, False, False, 16), "decoder self": (32, 12, 256, 256, True, True, 6),
          "decoder cross": (32, 12, 256, 1152, False, True, 6)}
tot = {}
for name, (B, H, Tq, Tk, causal, masked, calls) in SHAPES.items():
    D = H * 64
    q = torch.randn(B, Tq, D, device="cuda").bfloat16(); k = torch.randn(B, Tk, D, device="cuda").bfloat16(); v = torch.randn(B, Tk, D, device="cuda").bfloat16()
    kpm = torch.ones(B, Tk, dtype=torch.uint8, device="cuda") if masked else None
    if masked:
        kpm[:, Tk - 37:] = 0
    fl = 4.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)
    res = {}
    for ver in (1, 2):
        ops.attention_config(ver, ver)
        o, lse = ops.attention(q, k, v, H, 0.125, kpm=kpm, causal=causal, need_lse=True)
        do = torch.randn_like(o) if ver == 1 else res[1][2]
        grads = ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, kpm=kpm, causal=causal)
        tf = bench(lambda: ops.attention(q, k, v, H, 0.125, kpm=kpm, causal=causal, need_lse=True))
        tb = bench(lambda: ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, kpm=kpm, causal=causal))
        res[ver] = (o, lse, do, grads)
        tot[ver] = tot.get(ver, 0.0) + (tf + tb) * calls / 1e3
        print(f"v{ver} {name:14s} B={B} H={H} Tq={Tq} Tk={Tk}: fwd {tf:7.1f} us {fl / tf * 1e-6:6.0f} TF/s | bwd {tb:7.1f} us {2.5 * fl / tb * 1e-6:6.0f} TF/s | per step (x{calls}): "
              f"fwd {tf * calls / 1e3:5.2f} ms bwd {tb * calls / 1e3:5.2f} ms", flush=True)
    d_o = (res[1][0].float() - res[2][0].float()).abs().max().item()
    d_l = (res[1][1] - res[2][1]).abs().max().item()
    d_g = max((a.float() - b.float()).abs().max().item() for a, b in zip(res[1][3], res[2][3]))
    # fp32 reference of one batch element
    qf, kf, vf = q[:1].float().view(1, Tq, H, 64).transpose(1, 2), k[:1].float().view(1, Tk, H, 64).transpose(1, 2), v[:1].float().view(1, Tk, H, 64).transpose(1, 2)
    sc = qf @ kf.transpose(-1, -2) * 0.125
    if masked:
        sc = sc.masked_fill(kpm[:1, None, None, :] == 0, float("-inf"))
    if causal:
        sc = sc.masked_fill(torch.ones(Tq, Tk, device="cuda", dtype=torch.bool).triu(1), float("-inf"))
    ref = (sc.softmax(-1) @ vf).transpose(1, 2).reshape(1, Tq, D)
    e1 = (res[1][0][:1].float() - ref).abs().max().item(); e2 = (res[2][0][:1].float() - ref).abs().max().item()
    print(f"   v1 vs v2: max |dO| {d_o:.3e}  |dLSE| {d_l:.3e}  |dgrads| {d_g:.3e};  vs fp32: v1 {e1:.3e} v2 {e2:.3e}", flush=True)
print("attention per step (ms): " + "  ".join(f"v{v}: {t:.2f}" for v, t in tot.items()))
