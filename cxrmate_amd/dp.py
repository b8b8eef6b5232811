"""Data parallelism over xGMI: one process per GPU, study-level sharding, RCCL collectives through torch.distributed.

The reference trains with Lightning `strategy: 'ddp'` (config/train/single_tf.yaml:8): bucketed NCCL all-reduce (mean) of the
gradients of every trainable parameter after each backward. Here the gradients already live in ONE flat fp32 buffer
(store.ParamStore), so the "buckets" are the contiguous trainable ranges of that buffer: the decoder range is reduced on a side
stream while the encoder backward is still running, the encoder range afterwards. The mean is folded into the optimiser
(`gscale = 1/world`), so the collective is a plain SUM. SCST additionally all-gathers the sampled / greedy sequences
(int64 [B, L], <= 64 KB per rank) for global reward statistics.

backend 'nccl' is RCCL on ROCm; the same code runs on 'gloo' (CPU tensors) for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = backend or os.environ.get("CXR_DIST_BACKEND") or None       # test hook: gloo lets several ranks share ONE GPU
    if os.environ.get("CXR_SINGLE_DEVICE"):                               # test hook: every rank on device 0 (multi-rank flow on a 1-GPU box)
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active() -> bool:
    """Collectives are issued: more than one rank, or a one-rank process group with CXR_DP_FORCE=1 (runs the RCCL stream / work-handle path
    of GradReducer and the SCST gather on a single GPU: the sums over one rank are the identity)."""
    return world_size() > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("CXR_DP_FORCE") == "1")


def shard_studies(n_studies: int, rank: int, world: int) -> range:
    """Contiguous study-level shard of a global batch (pure data parallel; no data-path collective)."""
    per = (n_studies + world - 1) // world
    return range(min(rank * per, n_studies), min((rank + 1) * per, n_studies))


class GradReducer:
    """SUM all-reduce of flat-gradient ranges, optionally overlapped with the remaining backward on a side stream."""

    def __init__(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]], max_bucket_elems: int = 64 << 20, cuts: Sequence[int] = (),
                 comm_dtype: torch.dtype | None = None):
        """comm_dtype = torch.bfloat16: every bucket travels as bf16 (half the bytes over xGMI: 225 instead of 449 MB for the TF step) -- cast, SUM
        all-reduce in bf16, cast back into the fp32 gradient buffer that the fused AdamW (fp32 moments, fp32 master weights) reads. The sum of W
        bf16 values rounds to 8 significant bits at every add: relative error of a reduced gradient element <= W * 2^-9 (1.6 % at 8 ranks; measured
        rel-rms against the fp32 reduction at 2 ranks: 3e-3, tests/test_cpu_host.py). Default None = fp32 on the wire (CXR_DP_BF16=1 turns it on)."""
        if comm_dtype is None and os.environ.get("CXR_DP_BF16") == "1":
            comm_dtype = torch.bfloat16
        self.comm_dtype = comm_dtype
        self._staged = []
        self.flat = flat
        self.buckets: List[Tuple[int, int]] = []
        pieces = []
        for lo, hi in ranges:                                    # never let a bucket straddle a cut (encoder | decoder boundary)
            for c in sorted(cuts):
                if lo < c < hi:
                    pieces.append((lo, c))
                    lo = c
            pieces.append((lo, hi))
        for lo, hi in pieces:
            while hi - lo > max_bucket_elems:
                self.buckets.append((lo, lo + max_bucket_elems))
                lo += max_bucket_elems
            if hi > lo:
                self.buckets.append((lo, hi))
        self._pending = []
        self._started = set()
        self._stream = torch.cuda.Stream() if flat.is_cuda else None
        self.log = None              # a list: receives (lo, hi) of every collective in launch order (every rank must issue the same sequence)

    def reduce_range(self, lo: int, hi: int, async_op: bool = True, after=None):
        """Start reducing every bucket inside [lo, hi). Safe to call while later kernels write OTHER ranges. after: an extra stream whose
        queued work (weight-gradient kernels) must finish before the collective reads the gradients."""
        if not active():
            return
        todo = [(a, b) for a, b in self.buckets if a >= lo and b <= hi and (a, b) not in self._started]
        self._started.update(todo)                                   # a later, wider reduce_range only picks up what is still missing
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            if after is not None:
                self._stream.wait_stream(after)
        elif after is not None:
            torch.cuda.current_stream().wait_stream(after)          # no private stream (gloo on device tensors): the collective runs on the current one
        def launch():
            for a, b in todo:
                if self.log is not None:
                    self.log.append((a, b))
                if self.comm_dtype is None:
                    self._pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=async_op))
                else:
                    wire = self.flat[a:b].to(self.comm_dtype)         # (on the reducer's stream: ordered before the collective)
                    self._pending.append(dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=async_op))
                    self._staged.append((a, b, wire))

        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                launch()
        else:
            launch()

    def wait(self):
        if self._staged and self._stream is not None:
            with torch.cuda.stream(self._stream):                     # the reducer's stream waits for its collectives, then widens the sums back to fp32
                for w in self._pending:
                    if w is not None:
                        w.wait()
                for a, b, wire in self._staged:
                    self.flat[a:b].copy_(wire)
        else:
            for w in self._pending:
                if w is not None:
                    w.wait()
            for a, b, wire in self._staged:
                self.flat[a:b].copy_(wire)
        self._staged = []
        self._pending = []
        self._started = set()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)


def all_gather_sequences(ids: torch.Tensor, pad_token_id: int) -> torch.Tensor:
    """ids int64 [B, L] (L may differ per rank) -> [world*B, Lmax], right-padded with pad_token_id."""
    w = world_size()
    if w == 1:
        return ids
    lens = torch.tensor([ids.shape[1]], dtype=torch.int64, device=ids.device)
    all_lens = [torch.zeros_like(lens) for _ in range(w)]
    dist.all_gather(all_lens, lens)
    lmax = int(max(int(x) for x in all_lens))
    padded = torch.full((ids.shape[0], lmax), pad_token_id, dtype=torch.int64, device=ids.device)
    padded[:, : ids.shape[1]] = ids
    out = [torch.empty_like(padded) for _ in range(w)]
    dist.all_gather(out, padded)
    return torch.cat(out, dim=0)


def all_reduce_mean_scalar(x: torch.Tensor) -> torch.Tensor:
    w = world_size()
    if w == 1:
        return x
    y = x.clone()
    dist.all_reduce(y, op=dist.ReduceOp.SUM)
    return y / w


def gather_scst_statistics(sampled: torch.Tensor, greedy: torch.Tensor, reward: torch.Tensor, baseline: torch.Tensor, pad_token_id: int,
                           max_sampled: int | None = None, max_greedy: int | None = None):
    """The SCST step's per-rank results -> what every rank needs for GLOBAL reward / baseline statistics (north_star: "all-gather of sampled /
    greedy sequences for the SCST baseline"; the reference logs per-rank means, longitudinal/scst/gt_prompt.py:135-140, and leaves the reduction
    to Lightning's sync_dist). Studies stay sharded: nothing here feeds the gradient, which is still reward - baseline per study.
    sampled / greedy int64 [B, L*] (lengths may differ per rank), reward / baseline fp32 [B] -> dict(sampled [W*B, Ls], greedy [W*B, Lg],
    reward [W*B], baseline [W*B]) in rank order. world_size 1: the inputs themselves.
    ONE collective: every rank contributes a fixed-width int64 record per study, [reward bits | baseline bits | sampled ids padded to Ls | greedy ids
    padded to Lg]. With max_sampled / max_greedy given (the caller's decode limits: scst_step passes them) the widths are known without talking to
    anybody and nothing synchronises the host; without them one MAX all-reduce of the two lengths comes first."""
    if not active():
        return {"sampled": sampled, "greedy": greedy, "reward": reward, "baseline": baseline}
    w = world_size()
    dev = sampled.device
    if max_sampled is None or max_greedy is None:
        lens = torch.tensor([sampled.shape[1], greedy.shape[1]], dtype=torch.int64, device=dev)
        dist.all_reduce(lens, op=dist.ReduceOp.MAX)
        max_sampled, max_greedy = int(lens[0]), int(lens[1])
    B = sampled.shape[0]
    rec = torch.full((B, 2 + max_sampled + max_greedy), pad_token_id, dtype=torch.int64, device=dev)
    rec[:, 0] = reward.float().contiguous().view(torch.int32).to(torch.int64)
    rec[:, 1] = baseline.float().contiguous().view(torch.int32).to(torch.int64)
    rec[:, 2: 2 + min(sampled.shape[1], max_sampled)] = sampled[:, :max_sampled]
    rec[:, 2 + max_sampled: 2 + max_sampled + min(greedy.shape[1], max_greedy)] = greedy[:, :max_greedy]
    out = torch.empty((w * B, rec.shape[1]), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, rec)
    return {"sampled": out[:, 2: 2 + max_sampled], "greedy": out[:, 2 + max_sampled:],
            "reward": out[:, 0].to(torch.int32).view(torch.float32), "baseline": out[:, 1].to(torch.int32).view(torch.float32)}
