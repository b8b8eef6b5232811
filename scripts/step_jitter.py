#!/usr/bin/env python3
"""Per-step GPU durations (events, no host sync inside the loop) and Python GC activity during the loop."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import SingleCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step
m = SingleCXREncoderDecoderModel(EncoderDecoderConfig(), device="cuda", seed=0); m.train()
opt = FusedAdamW(m, lr=5e-5)
px, inp, am, lab = bench.synth_batch(32, 256, 30000, "cuda", 1)
tt = m.token_ids_to_token_type_ids(inp, [3])
for _ in range(40): tf_train_step(m, opt, px, inp, am, tt, lab, 4)
torch.cuda.synchronize()
for mode in ("gc on", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.disable()
    log = []
    gc.callbacks.append(lambda phase, info: log.append((phase, info["generation"], time.perf_counter())))
    N = 60
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    t0 = time.perf_counter()
    for i in range(N):
        evs[i].record()
        tf_train_step(m, opt, px, inp, am, tt, lab, 4)
    evs[N].record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e3
    d = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
    ds = sorted(d)
    gcs = [(g, b - a) for (p1, g, a), (p2, _, b) in zip(log[::2], log[1::2])]
    print(f"{mode}: wall {wall:.2f} ms/step; per-step min {ds[0]:.2f} median {ds[N//2]:.2f} p90 {ds[int(N*0.9)]:.2f} max {ds[-1]:.2f}; "
          f"gc runs {len(gcs)} (gen2: {sum(1 for g, _ in gcs if g == 2)}), total pause {sum(t for _, t in gcs)*1e3:.1f} ms, longest {max([t for _, t in gcs] or [0])*1e3:.1f} ms")
    gc.callbacks.clear()
