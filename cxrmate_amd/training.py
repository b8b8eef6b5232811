"""Fused training steps on the MI355X engines (the caller-side analogue of the reference's Lightning `training_step`s).

  tf_train_step   = SingleCXR/MultiCXR/GTPrompt.training_step (modules/lightning_modules/single.py:449-475,
                    longitudinal/gt_prompt.py:186-249): forward -> cross-entropy(ignore_index=pad) -> backward -> AdamW.
Everything runs through the HIP kernels with no autograd graph: the loss kernel emits d(logits) directly, the engines'
backward passes accumulate into the flat gradient buffer, gradient all-reduce (RCCL) overlaps the encoder backward, and one
fused AdamW pass updates the fp32 master weights and their bf16 shadow.

`GraphedTFStep` captures the step's ~1700 kernel launches into three hipGraphs (forward + loss + decoder backward | encoder
backward | AdamW) that are replayed per step: the shapes are static, so the host (Python + ctypes, ~8 us per launch) leaves the
critical path. The RCCL all-reduces stay OUTSIDE the graphs, between the segments, so they still overlap the encoder backward.
"""
from __future__ import annotations

import os

import torch

from . import dp, ops


class FusedAdamW:
    """torch.optim.AdamW(params, lr) semantics (reference single.py:426-431: betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2 on
    every parameter) over the contiguous trainable ranges of the flat parameter buffer."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model, self.lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        model.enable_direct_grads()
        self.ranges = model.trainable_ranges()
        self.m = torch.zeros_like(model.flat32)
        self.v = torch.zeros_like(model.flat32)
        self.t_dev = torch.zeros((), dtype=torch.int32, device=model.device)       # step counter lives on the device (graph replay)
        self.split = model._offsets[next(k for k in model._offsets if k.startswith("decoder."))]
        # flat-buffer offsets where the parameters of each encoder stage begin (storage order: stage 0, 1, 2, projection head, then the decoder): the
        # gradients of stage s are complete when that stage's backward is, so data parallelism reduces the encoder range stage by stage under the
        # backward of the earlier stages -- what Lightning's DDP buckets do for the reference (config/train/single_tf.yaml:8). CvT-21: stage 2 + head
        # hold 29 of the encoder's 31 M parameters; stage 1 (1.8 M) goes under stage 0's backward, stage 0 (0.06 M) is the only range left for the end.
        stages = sorted({int(k.split(".")[4]) for k in model._offsets if k.startswith("encoder.cvt.encoder.stages.")})
        self.stage_start = {s_: min(o for k, o in model._offsets.items() if k.startswith(f"encoder.cvt.encoder.stages.{s_}.") and o < model._param_total)
                            for s_ in stages}
        last = stages[-1] if stages else None
        self.enc_tail = None if last is None else self.stage_start[last]
        self.enc_last_stage = last
        self.reducer = dp.GradReducer(model.gflat, self.ranges, cuts=[self.split] + [o for o in self.stage_start.values() if o > 0])

    def stage_range(self, s):
        """[lo, hi) of the flat buffer holding the parameters of encoder stage s (the last stage's range includes the projection head)."""
        starts = sorted(self.stage_start.values())
        lo = self.stage_start[s]
        later = [o for o in starts if o > lo]
        return lo, (later[0] if later else self.split)

    @property
    def t(self):
        return int(self.t_dev.item())

    def zero_grad(self):
        for lo, hi in self.ranges:
            self.model.gflat[lo:hi].zero_()

    def step(self, gscale: float = 1.0, lo_bound: int = None, hi_bound: int = None, begin: bool = True, finish: bool = True):
        """lo_bound / hi_bound: only the part of the flat buffer in [lo_bound, hi_bound) -- a step may be issued in two pieces (begin=True on the first:
        the step counter advances once; finish=True on the last), e.g. the decoder's parameters as soon as their gradients are complete."""
        mo = self.model
        if begin:
            ops.increment_(self.t_dev)
        for lo, hi in self.ranges:
            lo = lo if lo_bound is None else max(lo, lo_bound)
            hi = hi if hi_bound is None else min(hi, hi_bound)
            if lo < hi:
                ops.adamw_step(mo.flat32[lo:hi], mo.gflat[lo:hi], self.m[lo:hi], self.v[lo:hi], mo.flat16[lo:hi], self.lr, self.betas[0],
                               self.betas[1], self.eps, self.wd, 0, gscale, step_dev=self.t_dev)
        if finish:
            mo.shadow_dirty = False
            mo.shadow_version += 1          # engines re-derive their per-version weight re-layouts (BN fold, LoRA merge)


# ---------------------------------------------------------------------------------------------------- step phases
def _phase_fwd_loss_decbwd(model, opt, px, ids, am, tt, lab, pad_token_id, pos, logits_slice_from, join=True, zero=True):
    """zero grads -> encoder fwd -> decoder fwd -> fused CE (loss + dlogits) -> decoder bwd. Returns (loss, esaved, denc)."""
    dev = model.device
    if zero:
        side = ops.WGRAD_STREAM
        if side is not None and _ZERO_ON_SIDE:
            # the gradient buffer is not touched before the decoder backward, which starts by joining the weight-gradient stream: the five fill
            # kernels (0.1 ms) run there, under the forward pass
            side.wait_stream(torch.cuda.current_stream())          # the previous optimiser step has read the gradients
            with torch.cuda.stream(side):
                opt.zero_grad()
        else:
            opt.zero_grad()
    px = model._pixels(px)
    multi = px.dim() == 5
    flat = px.view(-1, *px.shape[-3:]) if multi else px
    enc_trainable = any(p.requires_grad for p in model.encoder.parameters())
    feats, esaved = model._enc.forward(flat, save=enc_trainable)
    tokens = model.config.encoder.tokens_per_image
    B = px.shape[0]
    enc = feats.view(B, -1, feats.shape[-1])
    enc_mask = ops.image_mask(px, tokens) if (multi and model.kind != "single") else None
    logits, dsaved = model._dec.forward(model._i64(ids, dev), enc, enc_mask, model._u8(am, dev), model._i64(tt, dev), model._i64(pos, dev), save=True,
                                        logits_bf16=_BF16_LOGITS)
    Bq, T, V = logits.shape
    lg = logits[:, logits_slice_from:, :]
    labels = model._i64(lab, dev).reshape(-1)
    if logits_slice_from:
        lg = lg.contiguous()
    w = ops.ce_weights(labels, pad_token_id)
    loss, _, dl = ops.softmax_ce(lg.reshape(-1, V), labels, pad_token_id, w)
    if logits_slice_from:
        full = torch.zeros((Bq, T, dl.shape[1]), dtype=dl.dtype, device=dev)
        full[:, logits_slice_from:, :] = dl.view(Bq, T - logits_slice_from, -1)
        dl = full.view(Bq * T, -1)
    denc = model._dec.backward(dsaved, dlogits=dl, need_denc=enc_trainable)
    if join:
        ops.wgrad_join()                  # decoder weight gradients complete (graph capture needs the fork joined; an all-reduce may start now)
    return loss, esaved, denc


def _phase_encbwd(model, esaved, denc, on_stage_done=None):
    if esaved is not None:
        model._enc.backward(esaved, denc.view(-1, denc.shape[-1]), on_stage_done=on_stage_done)
    ops.wgrad_join()


_ZERO_ON_SIDE = os.environ.get("CXR_ZERO_ON_SIDE", "1") != "0"              # A/B switch: 0 = gradient zeroing on the main stream
_EARLY_DEC_ADAMW = os.environ.get("CXR_EARLY_DEC_ADAMW", "1") != "0"      # A/B switch: 0 = one AdamW launch at the end of the step
_EARLY_ENC_ADAMW = os.environ.get("CXR_EARLY_ENC_ADAMW", "1") != "0"      # A/B switch: 0 = the whole encoder range updated at the end of the step
_BF16_LOGITS = os.environ.get("CXR_BF16_LOGITS", "1") != "0"      # training step: bf16 logits as under the reference's autocast (0: fp32)


class wgrad_overlap:
    """Context: run weight-gradient kernels on a side stream (see ops.WGRAD_STREAM)."""
    _stream = None

    def __enter__(self):
        if wgrad_overlap._stream is None and os.environ.get("CXR_WGRAD_OVERLAP", "1") != "0":      # 0: A/B switch, everything on one stream
            # CXR_WGRAD_PRIORITY: stream priority of the weight-gradient stream (default: torch's default = lowest; negative = higher)
            prio = int(os.environ.get("CXR_WGRAD_PRIORITY", "0"))
            wgrad_overlap._stream = torch.cuda.Stream(priority=prio)
        ops.wgrad_flush()
        self.prev, ops.WGRAD_STREAM = ops.WGRAD_STREAM, wgrad_overlap._stream
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                ops.wgrad_flush()                           # launches still collected for this stream are issued on it
                ops.wgrad_reduce()                          # ... and the split sums they left pending are added (one launch)
            else:
                ops.wgrad_discard()                         # the step failed: nothing of it is launched later by an unrelated step
        finally:
            ops.WGRAD_STREAM = self.prev
        return False


def tf_train_step(model, opt: FusedAdamW, pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids,
                  pad_token_id, decoder_position_ids=None, logits_slice_from: int = 0, accumulate=None):
    """One teacher-forcing optimisation step (eager launches); returns the (detached, device) loss tensor [1].
    accumulate = (j, k): micro-batch j of k of one optimiser step -- gradient accumulation as the reference trains (config/train/single_tf.yaml:16-17:
    mbatch_size 8, accumulated_mbatch_size 32; under DDP Lightning wraps the first k - 1 micro-steps in `no_sync()`): micro-step 0 zeroes the
    gradient buffer, every micro-step adds its gradients, ONLY the last one all-reduces (once per optimiser step, not once per micro-batch) and
    runs AdamW on the mean over ranks and micro-batches (gscale = 1 / (world * k))."""
    with wgrad_overlap():
        return _tf_train_step(model, opt, pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids,
                              pad_token_id, decoder_position_ids, logits_slice_from, accumulate)


def _tf_train_step(model, opt, pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids, pad_token_id,
                   decoder_position_ids, logits_slice_from, accumulate=None):
    j, k = accumulate if accumulate is not None else (0, 1)
    assert 0 <= j < k
    # eager launches: the main stream goes straight from the decoder backward into the encoder backward; the decoder's weight-gradient
    # kernels still queued on the side stream keep running beside it (one join at the end of the step)
    loss, esaved, denc = _phase_fwd_loss_decbwd(model, opt, pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids,
                                                label_ids, pad_token_id, decoder_position_ids, logits_slice_from, join=False, zero=j == 0)
    if j + 1 < k:                                                # no_sync micro-step: gradients stay local, no optimiser step
        _phase_encbwd(model, esaved, denc, None)
        return loss
    world = dp.world_size() if dp.active() else 1
    sync = dp.active()
    enc_trainable = esaved is not None
    if sync and enc_trainable:
        # decoder parameters sit after the encoder's in the flat buffer: reduce them while the encoder backward runs (the reducer's stream
        # waits for the weight-gradient stream, the main stream does not)
        ops.wgrad_flush()
        ops.wgrad_reduce()
        opt.reducer.reduce_range(opt.split, model._param_total, after=ops.WGRAD_STREAM)
    early = None
    if sync and enc_trainable and opt.enc_tail:
        # ... and every encoder stage as soon as its backward is through (the last stage + projection head first: 93 % of the encoder's parameters),
        # while the earlier stages are still in backward; stage 0 is picked up by the closing reduce_range below
        def early(s):
            if s != min(opt.stage_start):
                ops.wgrad_flush()
                ops.wgrad_reduce()
                opt.reducer.reduce_range(*opt.stage_range(s), after=ops.WGRAD_STREAM)
    gscale = 1.0 / (world * k)
    side = ops.WGRAD_STREAM
    dec_early = _EARLY_DEC_ADAMW and not sync and enc_trainable and side is not None
    if dec_early:
        # one rank: the decoder's gradients are final once its weight-gradient kernels are through -- its 81 M parameters (0.43 ms of AdamW at the
        # HBM rate) are updated on the weight-gradient stream, behind those kernels, while the main stream runs the encoder backward. Nothing
        # reads decoder weights before the next forward; the join at the end of the encoder backward orders the update before it.
        ops.wgrad_flush()
        ops.wgrad_reduce()                                         # the decoder's pending split sums: ONE launch, in front of the update
        side.wait_stream(torch.cuda.current_stream())              # every decoder dX kernel (they read the weights) is in front of the update
        with torch.cuda.stream(side):
            opt.step(gscale=gscale, lo_bound=opt.split, begin=True, finish=False)
    enc_hi = opt.split
    if dec_early and _EARLY_ENC_ADAMW and opt.enc_tail:
        # ... and the encoder's last stage + projection head (29 of its 31 M parameters) the same way, as soon as that stage's backward has been issued:
        # its update runs on the weight-gradient stream under the backward of the earlier stages, which read none of its weights
        last_lo, last_hi = opt.stage_range(opt.enc_last_stage)

        def early(s):
            nonlocal enc_hi
            if s == opt.enc_last_stage and last_lo > 0:
                ops.wgrad_flush()
                ops.wgrad_reduce()
                side.wait_stream(torch.cuda.current_stream())      # the stage's dX kernels (weight readers) and main-stream parameter-gradient writers
                with torch.cuda.stream(side):
                    opt.step(gscale=gscale, lo_bound=last_lo, hi_bound=last_hi, begin=False, finish=False)
                enc_hi = last_lo
    _phase_encbwd(model, esaved, denc, early)
    if sync:
        opt.reducer.reduce_range(0, opt.split if enc_trainable else model._param_total)      # whatever has not been started yet
        opt.reducer.wait()
    if dec_early:
        opt.step(gscale=gscale, hi_bound=enc_hi, begin=False, finish=True)
    else:
        opt.step(gscale=gscale)
    return loss


class GraphedTFStep:
    """hipGraph replay of tf_train_step for a fixed batch geometry. Call with new tensors of the captured shapes."""

    def __init__(self, model, opt: FusedAdamW, pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids,
                 pad_token_id, decoder_position_ids=None, warmup: int = 2):
        self.model, self.opt = model, opt
        dev = model.device
        self.static = [t.detach().to(dev).clone() if t is not None else None
                       for t in (pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids, decoder_position_ids)]
        self.world = dp.world_size() if dp.active() else 1
        self.sync = dp.active()
        px, ids, am, tt, lab, pos = self.static
        if model.training and model.static_dropout_seed is None:
            # captured kernels read the dropout seed from this device word; each replay advances it (store.next_dropout_seed)
            model.static_dropout_seed = torch.full((1,), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up off the default stream (allocator + lazy kernel loading)
            for _ in range(warmup):
                tf_train_step(model, opt, px, ids, am, tt, lab, pad_token_id, pos)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.g1, self.g2, self.g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with wgrad_overlap():
            with ops.graph_capture(self.g1, pool=pool):
                self.loss, self._esaved, self._denc = _phase_fwd_loss_decbwd(model, opt, px, ids, am, tt, lab, pad_token_id, pos, 0)
            with ops.graph_capture(self.g2, pool=pool):
                _phase_encbwd(model, self._esaved, self._denc)
        with ops.graph_capture(self.g3, pool=pool):
            opt.step(gscale=1.0 / self.world)
        self.enc_trainable = self._esaved is not None

    def __call__(self, pixel_values=None, decoder_input_ids=None, decoder_attention_mask=None, decoder_token_type_ids=None, label_ids=None,
                 decoder_position_ids=None):
        for dst, src in zip(self.static, (pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids,
                                          decoder_position_ids)):
            if src is not None and dst is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        model, opt = self.model, self.opt
        if model.training:
            model._enc._bn_version += 1       # the replay moves the BatchNorm running statistics: eval-mode folds must be re-derived
        self.g1.replay()
        if self.sync and self.enc_trainable:
            opt.reducer.reduce_range(opt.split, model._param_total)
        self.g2.replay()
        if self.sync:
            opt.reducer.reduce_range(0, opt.split if self.enc_trainable else model._param_total)
            opt.reducer.wait()
        self.g3.replay()
        # the captured AdamW rewrote master + shadow: what FusedAdamW.step does on the host side (engines key their per-version weight re-layouts,
        # LoRA merges, W^T copies and decode sessions on shadow_version)
        model.shadow_dirty = False
        model.shadow_version += 1
        return self.loss
