"""Fused training steps on the MI355X engines (the caller-side analogue of the reference's Lightning `training_step`s).

  tf_train_step   = SingleCXR/MultiCXR/GTPrompt.training_step (modules/lightning_modules/single.py:449-475,
                    longitudinal/gt_prompt.py:186-249): forward -> cross-entropy(ignore_index=pad) -> backward -> AdamW.
Everything runs through the HIP kernels with no autograd graph: the loss kernel emits d(logits) directly, the engines'
backward passes accumulate into the flat gradient buffer, gradient all-reduce (RCCL) overlaps the encoder backward, and one
fused AdamW pass updates the fp32 master weights and their bf16 shadow.
"""
from __future__ import annotations

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
        self.t = 0
        self.split = model._offsets[next(k for k in model._offsets if k.startswith("decoder."))]
        self.reducer = dp.GradReducer(model.gflat, self.ranges, cuts=[self.split])

    def zero_grad(self):
        for lo, hi in self.ranges:
            self.model.gflat[lo:hi].zero_()

    def step(self, gscale: float = 1.0):
        self.t += 1
        mo = self.model
        for lo, hi in self.ranges:
            ops.adamw_step(mo.flat32[lo:hi], mo.gflat[lo:hi], self.m[lo:hi], self.v[lo:hi], mo.flat16[lo:hi], self.lr, self.betas[0],
                           self.betas[1], self.eps, self.wd, self.t, gscale)
        mo.shadow_dirty = False
        mo.shadow_version += 1              # engines re-derive their per-version weight re-layouts (BN fold, LoRA merge)


def tf_train_step(model, opt: FusedAdamW, pixel_values, decoder_input_ids, decoder_attention_mask, decoder_token_type_ids, label_ids,
                  pad_token_id, decoder_position_ids=None, logits_slice_from: int = 0):
    """One teacher-forcing optimisation step; returns the (detached, device) loss tensor [1]."""
    dev = model.device
    opt.zero_grad()
    px = model._pixels(pixel_values)
    multi = px.dim() == 5
    flat = px.view(-1, *px.shape[-3:]) if multi else px
    enc_trainable = any(p.requires_grad for p in model.encoder.parameters())
    feats, esaved = model._enc.forward(flat, save=enc_trainable)
    tokens = model.config.encoder.tokens_per_image
    B = px.shape[0]
    enc = feats.view(B, -1, feats.shape[-1])
    enc_mask = ops.image_mask(px, tokens) if (multi and model.kind != "single") else None
    ids = model._i64(decoder_input_ids, dev)
    logits, dsaved = model._dec.forward(ids, enc, enc_mask, model._u8(decoder_attention_mask, dev), model._i64(decoder_token_type_ids, dev),
                                        model._i64(decoder_position_ids, dev), save=True)
    Bq, T, V = logits.shape
    lg = logits[:, logits_slice_from:, :]
    labels = model._i64(label_ids, dev).reshape(-1)
    if logits_slice_from:
        lg = lg.contiguous()
    w = ops.ce_weights(labels, pad_token_id)
    loss, _, dl = ops.softmax_ce(lg.reshape(-1, V), labels, pad_token_id, w)
    if logits_slice_from:
        full = torch.zeros((Bq, T, dl.shape[1]), dtype=dl.dtype, device=dev)
        full[:, logits_slice_from:, :] = dl.view(Bq, T - logits_slice_from, -1)
        dl = full.view(Bq * T, -1)
    denc = model._dec.backward(dsaved, dlogits=dl, need_denc=enc_trainable)
    ranges = opt.ranges
    world = dp.world_size()
    if world > 1 and enc_trainable:
        # decoder parameters sit after the encoder's in the flat buffer: reduce them while the encoder backward runs
        split = opt.split
        opt.reducer.reduce_range(split, model._param_total)
    if enc_trainable:
        model._enc.backward(esaved, denc.view(-1, denc.shape[-1]))
    if world > 1:
        if enc_trainable:
            opt.reducer.reduce_range(0, split)
        else:
            opt.reducer.reduce_range(0, model._param_total)
        opt.reducer.wait()
    opt.step(gscale=1.0 / world)
    return loss
