#!/usr/bin/env python3
"""aten operators (not C-ABI launches) executed by one eager TF training step: what torch itself still launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import SingleCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step
m = SingleCXREncoderDecoderModel(EncoderDecoderConfig(), device="cuda", seed=0); m.train()
opt = FusedAdamW(m, lr=5e-5)
px, inp, am, lab = bench.synth_batch(32, 256, 30000, "cuda", 1)
tt = m.token_ids_to_token_type_ids(inp, [3])
for _ in range(3): tf_train_step(m, opt, px, inp, am, tt, lab, 4)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tf_train_step(m, opt, px, inp, am, tt, lab, 4)
torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.key not in ("aten::empty", "aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::reshape", "aten::_reshape_alias", "aten::unsqueeze", "aten::t", "aten::transpose", "aten::permute", "aten::stride", "aten::empty_strided", "aten::empty_like", "aten::expand", "aten::alias", "aten::detach", "aten::_unsafe_view", "aten::squeeze", "aten::narrow", "aten::unbind", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::result_type", "aten::lift_fresh", "aten::view_as")]
for e in sorted(rows, key=lambda e: -e.count)[:25]:
    print(f"{e.count:5d}  {e.key:28s}  {' <- '.join(s.split('/')[-1] for s in e.stack[:3])}")
