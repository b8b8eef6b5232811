"""A/B: stream priorities of the training step's two streams (main = dX chain, side = weight gradients): python scripts/tf_priority_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cxrmate_amd import training  # noqa: E402
from cxrmate_amd.config import EncoderDecoderConfig  # noqa: E402
from cxrmate_amd.modelling import MultiCXREncoderDecoderModel  # noqa: E402
from cxrmate_amd.training import FusedAdamW, tf_train_step  # noqa: E402

dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range())
model = MultiCXREncoderDecoderModel(EncoderDecoderConfig(), device=dev, seed=0).train()
opt = FusedAdamW(model, lr=1e-5)
px, inp, am, lab = bench.synth_batch(32, 256, 30000, dev, 0, n_images=2)
tt = model.token_ids_to_token_type_ids(inp, [3])


def run(main_prio, side_prio, n=8):
    training.wgrad_overlap._stream = torch.cuda.Stream(priority=side_prio)
    ms = torch.cuda.Stream(priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
    ms.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ms):
        for _ in range(3):
            tf_train_step(model, opt, px, inp, am, tt, lab, pad_token_id=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tf_train_step(model, opt, px, inp, am, tt, lab, pad_token_id=4)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


lo, hi = torch.cuda.Stream.priority_range()        # (least, greatest) = (0, -1)
for rep in range(2):
    for mp, sp in [(None, 0), (hi, 0), (hi, lo), (None, lo), (0, hi)]:
        print(f"main priority {mp}, weight-gradient priority {sp}: {run(mp, sp):.2f} ms")
