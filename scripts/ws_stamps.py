#!/usr/bin/env python3
"""In-kernel timeline of the W-stationary GEMM (s_memtime stamps of wave 0; dbg & 4)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops
from cxrmate_amd._lib import LIB
BF = torch.bfloat16
for M, N, mode, bm in [(36928, 384, 0, 32), (36928, 1536, 2, 32)]:
    K = 384
    nb = 6
    As = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nb)]
    Cs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nb)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    LIB.call("cxr_gemm_pk_config", 0, 0, -1, -1)
    LIB.call("cxr_gemm_ws_config", 1, bm, 1, 256, 4)
    for j in range(nb):
        ops.gemm_nt(As[j], w, bias=bias, out=Cs[j], act=1 if mode == 2 else 0)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 64, dtype=np.uint64)
    LIB.call("cxr_gemm_ws_stamps", buf.ctypes.data, buf.nbytes)
    st = buf.reshape(512, 64)
    n = st[:, 63].astype(int)
    used = np.nonzero(n)[0]
    print(f"== {M}x{N} mode {mode} bm {bm}: {len(used)} workgroups, {n[used[0]]} stamps each (shader cycles, relative to the workgroup's entry)")
    for w_ in list(used[:1]) + [used[len(used) // 2]]:
        ts = st[w_, :n[w_]]
        t0 = int(ts[0]) & ~3
        print(f" wg {w_:3d}: " + " ".join(f"{'SBEX'[int(t) & 3]}{(int(t) & ~3) - t0}" for t in ts))
    tot = np.array([(int(st[w_, n[w_] - 1]) & ~3) - (int(st[w_, 0]) & ~3) for w_ in used])
    print(f"  in-kernel cycles per workgroup: min {tot.min()} median {int(np.median(tot))} max {tot.max()}")
