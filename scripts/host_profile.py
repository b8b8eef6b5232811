#!/usr/bin/env python3
"""Host-side (Python) cost of one eager TF training step."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import SingleCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step
m = SingleCXREncoderDecoderModel(EncoderDecoderConfig(), device="cuda", seed=0)
m.train()
opt = FusedAdamW(m, lr=5e-5)
px, inp, am, lab = bench.synth_batch(32, 256, 30000, "cuda", 1)
tt = m.token_ids_to_token_type_ids(inp, [3])
for _ in range(3): tf_train_step(m, opt, px, inp, am, tt, lab, 4)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): tf_train_step(m, opt, px, inp, am, tt, lab, 4)
t_host = (time.perf_counter() - t) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t) / 5
print(f"host enqueue per step {t_host*1e3:.1f} ms ; with final sync {t_all*1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
tf_train_step(m, opt, px, inp, am, tt, lab, 4)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
