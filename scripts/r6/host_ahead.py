#!/usr/bin/env python3
"""Round 6 lab: does the host run ahead of the GPU in the eager TF step?  Host time to ISSUE N steps (no synchronisation) against the wall time of the
N steps, and the host time of the step's pieces (cProfile top entries of one step).  python3 scripts/r6/host_ahead.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import MultiCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step
dev = "cuda"
m = MultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0)
m.train()
opt = FusedAdamW(m, lr=5e-5)
px, inp, am, lab = bench.synth_batch(32, 256, 30000, dev, 1000, 2)
tt = m.token_ids_to_token_type_ids(inp, [3])
step = lambda: tf_train_step(m, opt, px, inp, am, tt, lab, pad_token_id=4)
for _ in range(12):
    step()
torch.cuda.synchronize()
for rep in range(3):
    N = 10
    marks = []
    t0 = time.perf_counter()
    for _ in range(N):
        step()
        marks.append(time.perf_counter())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per = [(b - a) * 1e3 for a, b in zip([t0] + marks[:-1], marks)]
    print(f"issue {N} steps: host {1e3 * (t1 - t0) / N:.2f} ms per step, wall {1e3 * (t2 - t0) / N:.2f} ms per step; host per step:", " ".join(f"{p:.1f}" for p in per), flush=True)
# the same with a synchronisation before every step: the host's own cost of a step when nothing is queued ahead
hs = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    hs.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host ms to issue ONE step from an idle GPU:", " ".join(f"{h:.1f}" for h in hs), flush=True)
import cProfile, pstats, io
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
