#!/usr/bin/env python3
"""Per-shape GEMM timing of one TF training step (bench workload): python scripts/gemm_profile.py [--batch 32]"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cxrmate_amd import ops
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import MultiCXREncoderDecoderModel, SingleCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--images", type=int, default=2); a = ap.parse_args()
cfg = EncoderDecoderConfig()
m = (MultiCXREncoderDecoderModel if a.images > 1 else SingleCXREncoderDecoderModel)(cfg, device="cuda", seed=0)
m.train()
opt = FusedAdamW(m, lr=5e-5)
px, inp, am, lab = bench.synth_batch(a.batch, 256, 30000, "cuda", 1, a.images)
tt = m.token_ids_to_token_type_ids(inp, [3])
for _ in range(2):
    tf_train_step(m, opt, px, inp, am, tt, lab, 4)
ops.GEMM_PROFILE = []
tf_train_step(m, opt, px, inp, am, tt, lab, 4)
torch.cuda.synchronize()
prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
agg = collections.OrderedDict()
for fl, e0, e1, shp, _ in prof:
    d = agg.setdefault(shp, [0, 0.0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"{'M':>8} {'N':>6} {'K':>6} {'calls':>5} {'ms':>8} {'%':>5} {'TF/s':>7}")
for shp, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(shp[0]):>8} {shp[1]:6d} {shp[2]:6d} {str(shp[3]) if len(shp) > 3 else '':>7} {n:5d} {ms:8.3f} {100*ms/tot:5.1f} {fl/ms/1e9:7.1f}")
print("total ms", tot, "TF/s", sum(v[2] for v in agg.values()) / tot / 1e9)
