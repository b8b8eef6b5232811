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


def shard_studies(n_studies: int, rank: int, world: int) -> range:
    """Contiguous study-level shard of a global batch (pure data parallel; no data-path collective)."""
    per = (n_studies + world - 1) // world
    return range(min(rank * per, n_studies), min((rank + 1) * per, n_studies))


class GradReducer:
    """SUM all-reduce of flat-gradient ranges, optionally overlapped with the remaining backward on a side stream."""

    def __init__(self, flat: torch.Tensor, ranges: Sequence[Tuple[int, int]], max_bucket_elems: int = 64 << 20, cuts: Sequence[int] = ()):
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

    def reduce_range(self, lo: int, hi: int, async_op: bool = True, after=None):
        """Start reducing every bucket inside [lo, hi). Safe to call while later kernels write OTHER ranges. after: an extra stream whose
        queued work (weight-gradient kernels) must finish before the collective reads the gradients."""
        if world_size() == 1:
            return
        todo = [(a, b) for a, b in self.buckets if a >= lo and b <= hi and (a, b) not in self._started]
        self._started.update(todo)                                   # a later, wider reduce_range only picks up what is still missing
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            if after is not None:
                self._stream.wait_stream(after)
        elif after is not None:
            torch.cuda.current_stream().wait_stream(after)          # no private stream (gloo on device tensors): the collective runs on the current one
        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                for a, b in todo:
                    self._pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=async_op))
        else:
            for a, b in todo:
                self._pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=async_op))

    def wait(self):
        for w in self._pending:
            if w is not None:
                w.wait()
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


def gather_scst_statistics(sampled: torch.Tensor, greedy: torch.Tensor, reward: torch.Tensor, baseline: torch.Tensor, pad_token_id: int):
    """The SCST step's per-rank results -> what every rank needs for GLOBAL reward / baseline statistics (north_star: "all-gather of sampled /
    greedy sequences for the SCST baseline"; the reference logs per-rank means, longitudinal/scst/gt_prompt.py:135-140, and leaves the reduction
    to Lightning's sync_dist). Studies stay sharded: nothing here feeds the gradient, which is still reward - baseline per study.
    sampled / greedy int64 [B, L*] (lengths may differ per rank), reward / baseline fp32 [B] -> dict(sampled [W*B, Ls], greedy [W*B, Lg],
    reward [W*B], baseline [W*B]) in rank order. world_size 1: the inputs themselves."""
    if world_size() == 1:
        return {"sampled": sampled, "greedy": greedy, "reward": reward, "baseline": baseline}
    rb = torch.stack([reward.float(), baseline.float()], dim=1).contiguous()
    out = [torch.empty_like(rb) for _ in range(world_size())]
    dist.all_gather(out, rb)
    rb = torch.cat(out, dim=0)
    return {"sampled": all_gather_sequences(sampled.contiguous(), pad_token_id), "greedy": all_gather_sequences(greedy.contiguous(), pad_token_id),
            "reward": rb[:, 0], "baseline": rb[:, 1]}
