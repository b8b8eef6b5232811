#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
for R, I, J in [(8192, 768, 768), (36928, 384, 384), (36928, 1536, 384), (36928, 384, 1536), (8192, 3072, 768), (8192, 768, 3072), (36864, 768, 768), (147456, 192, 192),
                (147456, 768, 192), (147456, 192, 768), (589824, 64, 64), (8192, 30000, 768), (9280, 384, 384), (36864, 9216, 768), (8192, 2304, 768), (36864, 192, 192),
                (36864, 384, 1728), (147456, 192, 576)]:
    p = torch.randn(R, I, device="cuda").bfloat16(); q = torch.randn(R, J, device="cuda").bfloat16()
    out = torch.zeros(I, J, device="cuda")
    for _ in range(3): ops.gemm_tn(p, q, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_tn(p, q, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"R={R:6d} I={I:5d} J={J:5d}  {us:8.1f} us  {2.0*R*I*J/us/1e6:7.1f} TF/s")
