#!/usr/bin/env python3
"""aten operators torch itself still launches in one no-grad FORWARD of the benchmark's model (bench.py forward_only): where the ~70 device-to-device
copies per forward in the kernel trace come from."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cxrmate_amd.config import EncoderDecoderConfig
from cxrmate_amd.modelling import MultiCXREncoderDecoderModel
m = MultiCXREncoderDecoderModel(EncoderDecoderConfig(), device="cuda", seed=0); m.train()
px, inp, am, lab = bench.synth_batch(32, 256, 30000, "cuda", 77, 2)
tt = m.token_ids_to_token_type_ids(inp, [3])
with torch.no_grad():
    for _ in range(3):
        m(pixel_values=px, decoder_input_ids=inp, decoder_attention_mask=am, decoder_token_type_ids=tt).logits
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        m(pixel_values=px, decoder_input_ids=inp, decoder_attention_mask=am, decoder_token_type_ids=tt).logits
    torch.cuda.synchronize()
skip = ("aten::empty", "aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::reshape", "aten::_reshape_alias", "aten::unsqueeze", "aten::t", "aten::transpose",
        "aten::permute", "aten::stride", "aten::empty_strided", "aten::empty_like", "aten::expand", "aten::alias", "aten::detach", "aten::_unsafe_view", "aten::squeeze", "aten::narrow",
        "aten::unbind", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::result_type", "aten::lift_fresh", "aten::view_as")
rows = [e for e in prof.key_averages(group_by_stack_n=5) if e.key.startswith("aten::") and e.key not in skip]
for e in sorted(rows, key=lambda e: -e.count)[:30]:
    print(f"{e.count:5d}  {e.key:28s}  {' <- '.join(s.split('/')[-1] for s in e.stack[:4])}")
