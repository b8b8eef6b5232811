"""Does the cross-attention step kernel run faster when its K / V come from the Infinity Cache? (round 6)
Six layers' packed K / V (57 MB each, 340 MB together: more than the 256 MiB cache, as in a token-step) are visited in turn:
  cold      : cross-attention only                                          (every launch streams from HBM, as today)
  prefetched: prefetch(layer l + 1) on a side stream while `filler` latency-bound launches of the main stream run, then cross-attention(l + 1)
  warm      : the same layer again and again                                (upper bound: everything resident)
Timed region = the cross-attention launches only (events around each), medians.
The numbers in profiles/r06_cross_prefetch.txt were taken with a read-and-discard HIP kernel in the library (cxr_prefetch_bytes, `wgs` workgroups, four
16-byte loads in flight per lane); the result was negative and the entry point was removed again -- this script now reads the bytes with a torch
reduction on the side stream, which is the same hint."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cxrmate_amd import ops

Bkv, share, Tk, H, D = 16, 2, 1152, 12, 768
B = Bkv * share
q = torch.randn(B, D, device="cuda").bfloat16()
pks = [ops.pack_cross_kv(torch.randn(Bkv, Tk, D, device="cuda").bfloat16(), torch.randn(Bkv, Tk, D, device="cuda").bfloat16(), H) for _ in range(6)]
out = torch.empty(ops.dal_rows(B), D, device="cuda", dtype=torch.bfloat16)
side = torch.cuda.Stream()
w = torch.randn(768, 768, device="cuda").bfloat16(); xa = torch.randn(32, 768, device="cuda").bfloat16()


def filler(n):                    # stand-in for the layer's other kernels: short dependent launches (5 us each, little traffic)
    y = xa
    for _ in range(n):
        y = ops.gemm_nt(y, w)
    return y


def cross(l):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.attention_cross_mfma(q, pks[l], Bkv, Tk, H, 0.125, out=out, out_dal=True)
    e1.record()
    return e0, e1


def run(mode, wgs=256, nfill=4, rounds=40):
    ev = []
    main = torch.cuda.current_stream()
    for r in range(rounds):
        for l in range(6):
            if mode == "prefetched":
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    pks[l][0].view(torch.int32).sum(); pks[l][1].view(torch.int32).sum()
            filler(nfill)
            ev.append(cross(0 if mode == "warm" else l))
        main.wait_stream(side)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev[12:]]
    return statistics.median(t), min(t)


for nfill in (4, 8):
    print(f"filler launches per layer: {nfill}")
    for mode, wgs in (("cold", 0), ("warm", 0), ("prefetched", 64), ("prefetched", 128), ("prefetched", 256), ("prefetched", 512)):
        med, mn = run(mode, wgs, nfill)
        print(f"  {mode:10s} wgs={wgs:3d}: cross-attention median {med:6.2f} us  min {mn:6.2f} us")
# what the filler chain costs with and without the prefetch beside it
for mode in ("cold", "prefetched"):
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(60):
        l = r % 6
        if mode == "prefetched":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pks[l][0].view(torch.int32).sum(); pks[l][1].view(torch.int32).sum()
        filler(4)
    e1.record(); main.wait_stream(side); torch.cuda.synchronize()
    print(f"filler chain of 4 launches, {mode}: {e0.elapsed_time(e1) * 1e3 / 60:.2f} us per chain")
