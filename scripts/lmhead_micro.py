"""LM-head forward GEMM (8192 x 30000 x 768) in the persistent kernel: team-major vs slot-major workgroup schedule (csrc/gemm_pk.hip): python scripts/lmhead_micro.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
M, N, K = 8192, 30000, 768
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
def t(n=20):
    for _ in range(3): ops.gemm_nt(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm_nt(a, w)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = None
for dbg, name in ((-100 - 32, "team-major (old)"), (-100, "slot-major")):
    LIB.call("cxr_gemm_pk_config", -1, -1, -1, dbg)
    us = t(); out = ops.gemm_nt(a, w)
    if ref is None: ref = out
    print(f"{name}: {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF/s  equal to first: {torch.equal(out, ref)}")
LIB.call("cxr_gemm_pk_config", -1, -1, -1, -100)
