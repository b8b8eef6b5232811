"""A few attention launches on the CvT stage-1 / stage-2 shapes for rocprofv3 --pmc runs (MFMA busy / wait / LDS counters per kernel):
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d out -- python3 scripts/attn_pmc_shapes.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
for (B, H, Tq, Tk) in [(64, 1, 9216, 2304), (64, 3, 2304, 576)]:
    D = H * 64
    q = torch.randn(B, Tq, D, device="cuda").bfloat16(); k = torch.randn(B, Tk, D, device="cuda").bfloat16(); v = torch.randn(B, Tk, D, device="cuda").bfloat16()
    for ver in (1, 2):
        ops.attention_config(ver, ver)
        for _ in range(3):
            o, lse = ops.attention(q, k, v, H, 0.125, need_lse=True)
            ops.attention_bwd(q, k, v, o, torch.randn_like(o), lse, H, 0.125)
torch.cuda.synchronize()
