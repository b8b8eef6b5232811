"""e4m3 GEMM (cxr_gemm_nt_fp8) vs the bf16 GEMM (cxr_gemm_nt) on the encoder's stage-3 shapes: python scripts/fp8_gemm_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops  # noqa: E402


def bench(f, n=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    for M, N, K in [(27648, 384, 384), (27648, 1536, 384), (27648, 384, 1536), (27648, 768, 384), (110592, 192, 192), (110592, 768, 192), (110592, 192, 768),
                    (442368, 64, 64), (442368, 256, 64), (442368, 64, 256),
                    (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (36864, 768, 768), (36928, 1536, 384), (36928, 384, 1536), (36928, 384, 384)]:
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda")
        a8, w8 = ops.quantize_fp8(a, 0.01), ops.quantize_fp8(w, 0.001)
        t16 = bench(lambda: ops.gemm_nt(a, w, bias=bias))
        t8 = bench(lambda: ops.gemm_nt_fp8(a8, w8, 1e-5, bias=bias))
        t88 = bench(lambda: ops.gemm_nt_fp8(a8, w8, 1e-5, bias=bias, out_scale=0.01, want_bf16=False))
        tq = bench(lambda: ops.quantize_fp8(a, 0.01))

        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: bf16 {t16:7.1f} us ({fl / t16 * 1e-6:6.1f} TF/s)  fp8->bf16 {t8:7.1f} us ({fl / t8 * 1e-6:6.1f} TF/s)  "
              f"fp8->fp8 {t88:7.1f} us  quantise A {tq:6.1f} us")
