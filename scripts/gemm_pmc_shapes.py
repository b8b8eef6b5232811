"""A handful of gemm_nt launches on the TF step's shapes, for rocprofv3 --pmc runs (MFMA busy / wait / LDS counters per shape)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
for M, N, K in [(36928, 384, 384), (36928, 1536, 384), (36928, 384, 1536), (8192, 768, 768), (8192, 3072, 768), (36864, 768, 768), (147456, 192, 192)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
    for _ in range(4):
        ops.gemm_nt(a, w, bias=b)
torch.cuda.synchronize()
