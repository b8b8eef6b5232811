#!/usr/bin/env python3
"""Host enqueue time vs GPU time of the phases of an eager TF step (is any phase launch-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cxrmate_amd import ops, training
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import SingleCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step
m = SingleCXREncoderDecoderModel(EncoderDecoderConfig(), device="cuda", seed=0); m.train()
opt = FusedAdamW(m, lr=5e-5)
px, inp, am, lab = bench.synth_batch(32, 256, 30000, "cuda", 1)
tt = m.token_ids_to_token_type_ids(inp, [3])
for _ in range(30): tf_train_step(m, opt, px, inp, am, tt, lab, 4)
torch.cuda.synchronize()
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((name, time.perf_counter(), ev))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        mark(name + ":begin"); r = f(*a, **k); mark(name + ":end"); return r
    setattr(obj, attr, g)
wrap(m._enc, "forward", "enc_fwd"); wrap(m._dec, "forward", "dec_fwd"); wrap(m._dec, "backward", "dec_bwd"); wrap(m._enc, "backward", "enc_bwd"); wrap(opt, "step", "adamw")
N = 10
res = {}
for it in range(N):
    marks.clear()
    mark("step:begin"); tf_train_step(m, opt, px, inp, am, tt, lab, 4); mark("step:end")
    torch.cuda.synchronize()
    d = {n: (t, e) for n, t, e in marks}
    for ph in ("enc_fwd", "dec_fwd", "dec_bwd", "enc_bwd", "adamw", "step"):
        (t0, e0), (t1, e1) = d[ph + ":begin"], d[ph + ":end"]
        h, g = (t1 - t0) * 1e3, e0.elapsed_time(e1)
        a = res.setdefault(ph, [0.0, 0.0]); a[0] += h / N; a[1] += g / N
print("phase      host-enqueue ms   gpu ms (events on the main stream)")
for ph, (h, g) in res.items():
    print(f"{ph:10s} {h:10.2f} {g:14.2f}")
