#!/usr/bin/env python3
"""The train-mode LoRA kernels alone at the SCST re-scoring shape (4080 rows, 768 columns): per-problem kernels vs the several-problems-per-launch ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops
BF = torch.bfloat16
R, K, T, r = 4080, 768, 255, 8
g = torch.Generator().manual_seed(0)
x, dq, dk = (torch.randn(R, K, generator=g).to(BF).cuda() for _ in range(3))
A, A2 = (0.05 * torch.randn(r, K, generator=g)).to(BF).cuda(), (0.05 * torch.randn(r, K, generator=g)).to(BF).cuda()
Bq, Bk = (0.05 * torch.randn(K, r, generator=g)).to(BF).cuda(), (0.05 * torch.randn(K, r, generator=g)).to(BF).cuda()
seed = torch.full((1,), 77, dtype=torch.int32, device="cuda")
t, t2 = ops.lora_down_multi([dict(x=x, W=A, drop=(0.1, 21)), dict(x=x, W=A2, drop=(0.1, 22))], rows_per_b=T, seed=seed, scale=4.0)
dt, dt2 = ops.lora_down_multi([dict(x=dq, W=Bq, w_is_b=True), dict(x=dk, W=Bk, w_is_b=True)], scale=4.0)
y, y2 = x.clone(), x.clone()
dB, dB2, dA, dA2 = torch.zeros(K, r, device="cuda"), torch.zeros(K, r, device="cuda"), torch.zeros(r, K, device="cuda"), torch.zeros(r, K, device="cuda")

def T_(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:72s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us")

T_("down, 2 adapters, per-problem kernel (one launch)", lambda: ops.lora_down(x, A, drop0=(0.1, 21), W1=A2, drop1=(0.1, 22), rows_per_b=T, seed=seed, scale=4.0))
T_("down, 2 adapters, matrix cores", lambda: ops.lora_down_multi([dict(x=x, W=A, drop=(0.1, 21)), dict(x=x, W=A2, drop=(0.1, 22))], rows_per_b=T, seed=seed, scale=4.0))
T_("dt = dy B, 2 x per-problem kernel", lambda: (ops.lora_down(dq, Bq, w_is_b=True, scale=4.0), ops.lora_down(dk, Bk, w_is_b=True, scale=4.0)))
T_("dt = dy B, 2 adapters, matrix cores", lambda: ops.lora_down_multi([dict(x=dq, W=Bq, w_is_b=True), dict(x=dk, W=Bk, w_is_b=True)], scale=4.0))
T_("up (forward), 2 x per-problem kernel", lambda: (ops.lora_up_add_(y, t, Bq, True), ops.lora_up_add_(y2, t2, Bk, True)))
T_("up (forward), 2 outputs in one launch", lambda: ops.lora_up_add_multi_([dict(y=y, t=t, W=Bq, w_is_b=True), dict(y=y2, t=t2, W=Bk, w_is_b=True)]))
T_("dx += (backward, masked), 2 x per-problem kernel", lambda: (ops.lora_up_add_(y, dt, A, False, drop=(0.1, 21), rows_per_b=T, seed=seed), ops.lora_up_add_(y, dt2, A2, False, drop=(0.1, 22), rows_per_b=T, seed=seed)))
T_("dx += (backward, masked), both adapters in one pass", lambda: ops.lora_up_add_multi_([dict(y=y, t=dt, W=A, drop=(0.1, 21)), dict(y=y, t=dt2, W=A2, drop=(0.1, 22))], rows_per_b=T, seed=seed))
T_("dB, dA of 2 adapters: 4 x per-problem kernel", lambda: (ops.lora_outer_into(dq, t, dB, r, 1), ops.lora_outer_into(x, dt, dA, 1, K, drop=(0.1, 21), rows_per_b=T, seed=seed),
                                                          ops.lora_outer_into(dk, t2, dB2, r, 1), ops.lora_outer_into(x, dt2, dA2, 1, K, drop=(0.1, 22), rows_per_b=T, seed=seed)))
T_("dB, dA of 2 adapters in one launch", lambda: ops.lora_outer_multi_into([dict(a=dq, t=t, G=dB, g_ks=r, g_rs=1), dict(a=x, t=dt, G=dA, g_ks=1, g_rs=K, drop=(0.1, 21)),
                                                                            dict(a=dk, t=t2, G=dB2, g_ks=r, g_rs=1), dict(a=x, t=dt2, G=dA2, g_ks=1, g_rs=K, drop=(0.1, 22))], rows_per_b=T, seed=seed))
