#!/usr/bin/env python3
"""In-kernel timeline of the persistent GEMM (s_memtime stamps of wave 0 of every workgroup; diagnostic build path dbg & 4)."""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16
for M, N, K, mode, cfgid in [(36928, 384, 384, 0, 1000), (36928, 1536, 384, 2, 1000), (36928, 384, 1536, 1, 1000)]:
    nb = 6
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
    Rs = [torch.randn(M, N, device="cuda").to(BF) for _ in range(nb)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    LIB.call("cxr_gemm_pk_config", 1, cfgid, 1, -100 - 4)
    for j in range(nb):
        if mode == 0: ops.gemm_nt(As[j], w, bias=bias, out=Cs[j])
        elif mode == 1: ops.gemm_nt(As[j], w, bias=bias, residual=Rs[j], out=Cs[j])
        else: ops.gemm_nt(As[j], w, bias=bias, act=1, out=Cs[j])
    torch.cuda.synchronize()
    buf = np.zeros(512 * 64, dtype=np.uint64)
    LIB.call("cxr_gemm_pk_stamps", buf.ctypes.data, buf.nbytes)
    st = buf.reshape(512, 64)
    n = st[:, 63].astype(int)
    used = np.nonzero(n)[0]
    t0 = min(int(st[w_, 0]) for w_ in used)
    print(f"== {M}x{N}x{K} mode {mode}: {len(used)} workgroups, stamps per wg {n[used[0]]}; s_memtime ticks are 100 MHz (10 ns)")
    for w_ in list(used[:3]) + [used[len(used) // 2], used[-1]]:
        ts = st[w_, :n[w_]]
        rel = [(int(t) & ~3) - (t0 & ~3) for t in ts]
        tags = [int(t) & 3 for t in ts]
        print(f" wg {w_:3d}: " + " ".join(f"{'SBEX'[g]}{r}" for g, r in zip(tags, rel)))
    ends = np.array([(int(st[w_, n[w_] - 1]) & ~3) - (t0 & ~3) for w_ in used]); starts = np.array([(int(st[w_, 0]) & ~3) - (t0 & ~3) for w_ in used])
    print(f"  start spread {starts.min()}..{starts.max()}, end {ends.min()}..{ends.max()} ticks")
