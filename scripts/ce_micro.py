#!/usr/bin/env python3
"""Fused softmax + cross-entropy + gradient over the training step's logits (8192 rows x 30000 bf16, gradient row stride 30016): python scripts/ce_micro.py
(CXR_CE_BF16ROW=0: the fp32-resident kernel). Prints the time and a checksum of loss / gradient for comparing the two kernels bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
torch.manual_seed(0)
R, V = 8192, 30000
lg = [(torch.randn(R, V, device="cuda") * 3).to(torch.bfloat16) for _ in range(2)]
lab = torch.randint(0, V, (R,), device="cuda"); lab[::7] = 4
w = ops.ce_weights(lab, 4)
for i in range(3): loss, rl, dl = ops.softmax_ce(lg[i % 2], lab, 4, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20): loss, rl, dl = ops.softmax_ce(lg[i % 2], lab, 4, w)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
loss, rl, dl = ops.softmax_ce(lg[0], lab, 4, w)
print(f"softmax_ce {R} x {V}: {us:.1f} us ({(R * V * 2 + R * dl.shape[1] * 2) / us / 1e3:.0f} GB/s)  loss {float(loss):.9f}  "
      f"grad checksum {int(dl.view(torch.int16).to(torch.int64).sum())}  abs {float(dl.float().abs().sum()):.6f}")
