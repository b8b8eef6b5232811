"""Out-of-bounds check of the attention backward kernels: outputs are column slices of wider, canary-filled matrices with guard rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
torch.manual_seed(0)
for (B, H, Tq, Tk) in [(2, 1, 9216, 2304), (2, 3, 2304, 576), (2, 6, 1100, 145), (3, 2, 1025, 70), (2, 1, 2000, 33)]:
    D = H * 64
    G = 64
    def fused(T):
        big = torch.randn(B, T + G, 3 * D, device="cuda").bfloat16()
        return big
    qkv_q = fused(Tq); qkv_k = fused(Tk)
    q = qkv_q[:, :Tq, 0:D]; k = qkv_k[:, :Tk, D:2 * D]; v = qkv_k[:, :Tk, 2 * D:3 * D]
    res = {}
    for ver in (1, 2):
        ops.attention_config(2, ver)
        o, lse = ops.attention(q, k, v, H, 0.125, need_lse=True)
        do = torch.randn_like(o) if ver == 1 else res[1][0]
        gq = torch.full((B, Tq + G, 3 * D), 7.0, device="cuda").bfloat16(); gk = torch.full((B, Tk + G, 3 * D), 7.0, device="cuda").bfloat16()
        dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, H, 0.125, dq_out=gq[:, :Tq, 0:D], dk_out=gk[:, :Tk, D:2 * D], dv_out=gk[:, :Tk, 2 * D:])
        torch.cuda.synchronize()
        bad = int((gq[:, Tq:] != 7.0).sum()) + int((gq[:, :Tq, D:] != 7.0).sum()) + int((gk[:, Tk:] != 7.0).sum()) + int((gk[:, :Tk, :D] != 7.0).sum())
        res[ver] = (do, dq.clone(), dk.clone(), dv.clone(), bad)
    d = max((a.float() - b.float()).abs().max().item() for a, b in zip(res[1][1:4], res[2][1:4]))
    print(f"B={B} H={H} Tq={Tq} Tk={Tk}: canary hits v1 {res[1][4]} v2 {res[2][4]}  max|v1-v2| {d:.3e}  nan {any(bool(torch.isnan(t).any()) for t in res[2][1:4])}", flush=True)
