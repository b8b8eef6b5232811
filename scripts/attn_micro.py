"""Forward / backward attention kernels alone on the TF step's shapes (2-image batch: 64 images / 32 studies): python scripts/attn_micro.py"""
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


for name, (B, H, Tq, Tk, causal, masked, calls) in {"cvt stage 1": (64, 1, 9216, 2304, False, False, 1), "cvt stage 2": (64, 3, 2304, 576, False, False, 4),
                                                     "cvt stage 3": (64, 6, 577, 145, False, False, 16), "decoder self": (32, 12, 256, 256, True, True, 6),
                                                     "decoder cross": (32, 12, 256, 1152, False, True, 6)}.items():
    D = H * 64
    q = torch.randn(B, Tq, D, device="cuda").bfloat16(); k = torch.randn(B, Tk, D, device="cuda").bfloat16(); v = torch.randn(B, Tk, D, device="cuda").bfloat16()
    kpm = torch.ones(B, Tk, dtype=torch.uint8, device="cuda") if masked else None
    fl = 4.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)
    o, lse = ops.attention(q, k, v, H, 0.125, kpm=kpm, causal=causal, need_lse=True)
    do = torch.randn_like(o)
    tf = bench(lambda: ops.attention(q, k, v, H, 0.125, kpm=kpm, causal=causal, need_lse=True))
    tb = bench(lambda: ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, kpm=kpm, causal=causal))
    print(f"{name:14s} B={B} H={H} Tq={Tq} Tk={Tk}: fwd {tf:7.1f} us {fl / tf * 1e-6:6.0f} TF/s | bwd {tb:7.1f} us {2.5 * fl / tb * 1e-6:6.0f} TF/s | per step (x{calls}): "
          f"fwd {tf * calls / 1e3:5.2f} ms bwd {tb * calls / 1e3:5.2f} ms")
