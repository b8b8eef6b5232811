#!/usr/bin/env python3
"""Stage-1 patch-embedding im2col (64 images 3 x 384 x 384 fp32 -> [589824, 192] bf16): python scripts/im2col_micro.py   (CXR_IM2COL_ROWS=0: gather kernel)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
px = [torch.randn(64, 3, 384, 384, device="cuda") for _ in range(4)]
for i in range(3): ops.im2col_pixels(px[i], 7, 4, 2, 192)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40): ops.im2col_pixels(px[i % 4], 7, 4, 2, 192)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 40
print(f"im2col 64 x 3 x 384 x 384: {us:.1f} us  ({(64*3*384*384*4 + 589824*192*2) / us / 1e3:.0f} GB/s algorithmic)")
