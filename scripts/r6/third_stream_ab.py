#!/usr/bin/env python3
"""What would the gradient all-reduce cost the TF step through the memory system of ONE GPU? (round 6, review item 8)

No multi-GPU node is available to the builder, so the collective itself cannot be measured. What one GPU can tell: the step already runs two streams
(main + weight gradients); data parallel adds a THIRD one, the reducer's, on which RCCL reads and writes the gradient buckets locally while it moves them
over xGMI. This script runs the real TF step with the reducer's real schedule (dp.GradReducer.reduce_range at the decoder | encoder | per-stage cuts,
wait() in front of AdamW) and replaces each all-reduce by a stand-in kernel on the reducer's stream that moves the bucket's bytes twice (bucket -> scratch,
scratch -> bucket: 2 reads + 2 writes, what a reduce-scatter + all-gather does locally) with a fixed number of workgroups (the pace).
    wire fp32 : 449 MB of buckets per step            wire bf16 : cast + 225 MB + cast back (GradReducer(comm_dtype=bf16))
A/B inside one process: stand-in off / on, alternating. Prints ms per step for each arm."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from cxrmate_amd import dp
from cxrmate_amd.modelling import MultiCXREncoderDecoderModel
from cxrmate_amd.training import FusedAdamW, tf_train_step
import torch.distributed as dist

lab = ctypes.CDLL(os.path.join(ROOT, "scripts", "lab", "libpaced_copy.so"))
lab.lab_paced_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
STATE = {"on": False, "wgs": 32, "bytes": 0, "calls": 0}
scratch = {}


class _Done:
    def wait(self):
        return True


def standin_all_reduce(t, op=None, async_op=False):
    if STATE["on"]:
        n = t.numel() * t.element_size()
        s = scratch.get(n)
        if s is None:
            s = scratch[n] = torch.empty(n, dtype=torch.uint8, device=t.device)
        st = torch.cuda.current_stream().cuda_stream
        assert lab.lab_paced_copy(t.data_ptr(), s.data_ptr(), n, STATE["wgs"], st) == 0
        assert lab.lab_paced_copy(s.data_ptr(), t.data_ptr(), n, STATE["wgs"], st) == 0
        STATE["bytes"] += n; STATE["calls"] += 1
    return _Done()


ACTIVE = {"on": True}
dp.active = lambda: ACTIVE["on"]             # True: the reducer's (multi-rank) schedule runs; world size stays 1 (no gradient scaling changes)
dist.all_reduce = standin_all_reduce
dev = torch.device("cuda:0")
cfg = bench.bench_config()
model = MultiCXREncoderDecoderModel(cfg, device=dev, seed=0); model.train()
px, inp, am, lab_ids = bench.synth_batch(32, 256, cfg.decoder.vocab_size, dev, 1000, 2)
tt = model.token_ids_to_token_type_ids(inp, [3])


def arm(comm_dtype, on, wgs, steps=20, multi_rank_schedule=True):
    ACTIVE["on"] = multi_rank_schedule
    opt = FusedAdamW(model, lr=5e-5)
    opt.reducer = dp.GradReducer(model.gflat, opt.ranges, cuts=[opt.split] + [o for o in opt.stage_start.values() if o > 0], comm_dtype=comm_dtype)
    STATE.update(on=on, wgs=wgs, bytes=0, calls=0)
    for _ in range(4):
        tf_train_step(model, opt, px, inp, am, tt, lab_ids, pad_token_id=4)
    torch.cuda.synchronize(); STATE.update(bytes=0, calls=0)
    t0 = time.perf_counter()
    for _ in range(steps):
        tf_train_step(model, opt, px, inp, am, tt, lab_ids, pad_token_id=4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, STATE["bytes"] / max(steps, 1) / 1e6, STATE["calls"] / max(steps, 1)


for _ in range(30):                                           # warm the box
    arm(None, False, 32, steps=1)
for rep in range(2):
    ms, _, _ = arm(None, False, 32, multi_rank_schedule=False)
    print(f"rep {rep}  {'one-rank schedule (the bench.py step)':42s} {ms:7.2f} ms/step", flush=True)
    for name, cd, on, wgs in (("multi-rank schedule, no stand-in", None, False, 32), ("fp32 wire, 16 workgroups", None, True, 16), ("fp32 wire, 32 workgroups", None, True, 32),
                              ("fp32 wire, 64 workgroups", None, True, 64), ("bf16 wire, 32 workgroups", torch.bfloat16, True, 32), ("bf16 wire, 64 workgroups", torch.bfloat16, True, 64)):
        ms, mb, calls = arm(cd, on, wgs)
        print(f"rep {rep}  {name:42s} {ms:7.2f} ms/step   stand-in moved {mb:6.1f} MB x 4 accesses in {calls:4.1f} launches-pairs per step", flush=True)
# the stand-in alone on the chip: how long the reducer's stream is busy per step at each pace (the time the collective would have to hide in)
g = model.gflat
s = torch.empty(g.numel() * 4, dtype=torch.uint8, device=dev)
for wgs in (16, 32, 64):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lab.lab_paced_copy(g.data_ptr(), s.data_ptr(), g.numel() * 4, wgs, torch.cuda.current_stream().cuda_stream)
        lab.lab_paced_copy(s.data_ptr(), g.data_ptr(), g.numel() * 4, wgs, torch.cuda.current_stream().cuda_stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"alone: {g.numel() * 4 / 1e6:.0f} MB moved twice by {wgs} workgroups: {ms:.2f} ms = {g.numel() * 4 * 4 / ms / 1e9:.2f} TB/s of local traffic, {g.numel() * 4 / ms / 1e6:.0f} GB/s of gradient bytes")
