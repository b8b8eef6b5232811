"""CvT-21 + projection head on MI355X: forward and hand-written backward over the HIP kernels.

Mirrors CvtWithProjectionHead / MultiCvtWithProjectionHead of the reference
(modules/transformers/single_model/modelling_single.py:43-78, multi_model/modelling_multi.py:43-87) on top of
transformers' CvtModel (TF5 = transformers/models/cvt/modeling_cvt.py @ 5.15.0). Activations are kept token-major
[Bn, L, C] in bf16 for the whole network (the reference flips between NCHW and token-major around every conv).
BatchNorm: eval mode folds the running statistics into the depthwise taps; train mode (`store.training`, i.e. after `model.train()`,
which is how the reference runs both the TF and the SCST stage -- SURVEY.md Q7) uses the batch statistics of the launch, moves the
running statistics, and back-propagates through the statistics. DropPath (stage 3, TF5 modeling_cvt.py:297-316,404-422) is a per-image
keep/scale of both residual branches in train mode; CvT's other dropouts have rate 0 in the reference configuration.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .config import CvtConfig


_TAIL_ON_MAIN = os.environ.get("CXR_TAIL_ON_MAIN", "1") != "0"      # A/B switch: 0 = the patch embedding's weight gradient (the step's last GEMM) on the weight-gradient stream like every other
_EARLY_PATCH_COL = os.environ.get("CXR_EARLY_PATCH_COL", "1") != "0"      # A/B switch: 0 = the pixel im2col matrix of the backward is built at the end of the step, 1 = during the forward pass


class CvtEncoderEngine:
    def __init__(self, store, cfg: CvtConfig, prefix: str = "encoder."):
        self.s, self.cfg, self.p = store, cfg, prefix
        self._prep_version = -1
        self._prep = {}
        self._wt_ready = False
        self._bn_version = 0                        # bumped whenever a train-mode forward moves the running statistics
        self._fold_cache = {}
        self._taps, self._taps_version, self._taps_index = None, -1, {}
        self._wt_event = None
        self._dp_rate, self._dp_factors = {}, {}
        self._fused_proj = os.environ.get("CXR_DWPROJ", "1") != "0"      # A/B switch: 0 = per-projection kernels of conv.hip
        self._embed_buf = {}                        # persistent conv-as-GEMM weight re-layouts (their addresses feed the batched transpose table)
        self._bt, self._bt_sig, self._bt_keys = None, None, None
        self.fp8 = None                             # {"w": {key: (e4m3 weight, scale)}, "a": {key: scale}} after enable_fp8(); None = bf16
        self._bwd_stats_from_y = os.environ.get("CXR_DW3_STATS_FROM_Y", "1") != "0"      # A/B switch: 0 = backward BatchNorm statistics recompute the convolution
        self._q8_fused = os.environ.get("CXR_FP8_FUSED", "1") != "0"      # A/B switch: 0 = separate bf16 -> e4m3 passes in front of the e4m3 GEMMs
        self._implicit_embed = os.environ.get("CXR_IMPLICIT_EMBED", "1") != "0"      # A/B switch: 0 = im2col + GEMM for the stage-2 / stage-3 embeddings
        self._patch_fused = os.environ.get("CXR_PATCH_EMBED_FUSED", "1") != "0"      # A/B switch: 0 = im2col + GEMM + LayerNorm for the stage-1 embedding
        self._amax = None                           # calibration pass: {key: running max |activation|}

    # ------------------------------------------------------------------------------------------ fp8 (e4m3) linear layers of the frozen encoder
    def enable_fp8(self, px, margin: float = 2.0, train: bool | None = None):
        """Post-training quantisation of the encoder's linear layers for gradient-free forwards (BASELINE.json configs[4], the reference's
        gen-prompt SCST: modules/lightning_modules/longitudinal/scst/gen_prompt.py:174-259): every Linear of the 21 blocks (q/k/v projections,
        attention output, both MLP layers: 99.5 % of the encoder's multiply-adds) runs as an e4m3 (OCP) GEMM with ONE scale per weight tensor
        (max|W| / 448) and ONE static scale per GEMM input (margin * max|activation| / 448 over the calibration images `px`, saturating). The
        patch embeddings (K = 147 / 576 / 1728 im2col rows), depthwise convolutions, LayerNorms, attention and the projection head stay bf16.
        Forwards that save activations for a backward pass ignore the switch. Call again after the weights change."""
        st = self.s
        self.fp8 = None
        self._amax = {}
        with torch.no_grad():
            self.forward(px, save=False, train=train)
        amax, self._amax = self._amax, None
        self.fp8 = {"w": {}, "a": {k: max(float(v), 1e-12) * margin / ops.FP8_MAX for k, v in amax.items()}, "version": st.shadow_version}
        self._quantise_weights(sorted({key[0] for key in amax}))
        return self

    def _quantise_weights(self, names):
        st = self.s
        for name in list(names):
            w = st.w16(name)
            sw = max(float(w.float().abs().max()), 1e-12) / ops.FP8_MAX
            old = self.fp8["w"].get(name)
            self.fp8["w"][name] = (ops.quantize_fp8(w, sw, out=None if old is None else old[0]), sw)

    def _linear(self, x, wname, bias=None, residual=None, act=0, row_scale=None, x8=None, out8_for=None):
        """x [M,K] bf16 (or its e4m3 form x8 from the producing layer) -> Linear. bf16 GEMM, or the e4m3 GEMM once enable_fp8() ran; during the
        calibration pass the input's max magnitude is recorded. out8_for = weight name of the ONLY consumer: the output is produced as e4m3 with
        that layer's input scale instead of bf16 (no separate quantisation pass) -> returns (None, e4m3)."""
        st = self.s
        if self._amax is not None:
            key = (wname, "in")
            m = x.float().abs().max()
            self._amax[key] = m if key not in self._amax else torch.maximum(self._amax[key], m)
        if self.fp8 is None or self._saving:
            return ops.gemm_nt(x, st.w16(wname), bias=None if bias is None else st.f32(bias), residual=residual, act=act, row_scale=row_scale), None
        w8, sw = self.fp8["w"][wname]
        sa = self.fp8["a"][(wname, "in")]
        if x8 is None:
            x8 = ops.quantize_fp8(x, sa)
        so = self.fp8["a"][(out8_for, "in")] if out8_for is not None else None
        return ops.gemm_nt_fp8(x8, w8, sa * sw, bias=None if bias is None else st.f32(bias), residual=residual, act=act, out_scale=so,
                               want_bf16=out8_for is None, row_scale=row_scale)

    # ------------------------------------------------------------------------------------------ weight preparation
    def _stage(self, s):
        return f"{self.p}cvt.encoder.stages.{s}."

    def prepare(self):
        """Per-weight-version re-layouts: conv taps as GEMM operands, BatchNorm folded into the depthwise taps."""
        st = self.s
        st.refresh_shadow()
        if self._prep_version == st.shadow_version:
            return self._prep
        cfg, prep = self.cfg, {}
        for s in range(len(cfg.depth)):
            sp = self._stage(s)
            w = st.w16(sp + "embedding.convolution_embeddings.projection.weight")
            co = w.shape[0]
            wp = self._embed_buf.get(s)
            if s == 0:
                k = w.shape[1] * w.shape[2] * w.shape[3]
                kpad = ((k + 63) // 64) * 64
                if wp is None or wp.device != w.device:
                    wp = self._embed_buf[s] = torch.zeros((co, kpad), dtype=torch.bfloat16, device=w.device)
                wp[:, :k] = w.reshape(co, k)                         # K order (c, ky, kx) = weight.view(Cout, -1)
                if self._fused_patch_embed(w.shape):
                    # the one-launch stage-1 embedding (csrc/conv.hip patch_embed_s1_kernel) reads the weights in its own packed order
                    pk = self._embed_buf.get("pk0")
                    if pk is None or pk.device != w.device:
                        pk = self._embed_buf["pk0"] = torch.empty((64, 192), dtype=torch.bfloat16, device=w.device)
                    ops.patch_embed_pack(st.f32(sp + "embedding.convolution_embeddings.projection.weight").contiguous(), out=pk)
                    prep[("embed_pk", 0)] = pk
            else:
                if wp is None or wp.device != w.device:
                    wp = self._embed_buf[s] = torch.empty((co, w.shape[1] * w.shape[2] * w.shape[3]), dtype=torch.bfloat16, device=w.device)
                wp.view(co, w.shape[2], w.shape[3], w.shape[1]).copy_(w.permute(0, 2, 3, 1))      # K order (ky, kx, c): channel-contiguous gathers
            prep[("embed", s)] = wp
        self._prep, self._prep_version = prep, st.shadow_version
        self._wt_ready = False
        self._fold_cache = {}
        return prep

    def _fused_patch_embed(self, wshape):
        """The stage-1 embedding runs as ONE kernel (direct 7 x 7 / stride 4 convolution on the matrix cores + bias + LayerNorm) for the reference's
        geometry: Conv2d(3, 64, 7, 4, 2). CXR_PATCH_EMBED_FUSED=0: im2col + GEMM + LayerNorm (A/B switch)."""
        cfg = self.cfg
        return (self._patch_fused and tuple(wshape) == (64, 3, 7, 7) and cfg.patch_stride[0] == 4 and cfg.patch_padding[0] == 2 and not cfg.cls_token[0])

    def _conv_prefix(self, s, l, name):
        return self._stage(s) + f"layers.{l}.attention.attention.convolution_projection_{name}.convolution_projection."

    def _fold_eval(self, s, l, name):
        """Depthwise taps with the RUNNING statistics folded in (eval-mode BatchNorm): cached per weight / running-stat version."""
        key = (s, l, name)
        hit = self._fold_cache.get(key)
        if hit is None or hit[0] != self._bn_version:
            st, cp = self.s, self._conv_prefix(s, l, name)
            fold = ops.bn_fold(st.f32(cp + "convolution.weight"), st.f32(cp + "normalization.weight"), st.f32(cp + "normalization.bias"),
                               st.f32(cp + "normalization.running_mean"), st.f32(cp + "normalization.running_var"), self.cfg.bn_eps)
            hit = self._fold_cache[key] = (self._bn_version, fold)
        return hit[1]

    def _raw_taps(self, s, l, name):
        """Raw depthwise taps as [9, C] fp32 (the layout the conv kernels read): persistent copies of the [C,1,3,3] parameters, all 3 x 21
        projections refreshed by ONE launch per weight version (layout plumbing)."""
        st = self.s
        if self._taps is None or self._taps_version != st.shadow_version:
            names = [(s_, l_, n_) for s_ in range(len(self.cfg.depth)) for l_ in range(self.cfg.depth[s_]) for n_ in ("query", "key", "value")]
            srcs = [st.f32(self._conv_prefix(*k) + "convolution.weight") for k in names]
            srcs = [w.view(w.shape[0], 9) for w in srcs]
            if self._taps is None or self._taps.sig != tuple(w.data_ptr() for w in srcs):       # first use, or the store was re-packed
                self._taps = ops.TapsLayout(srcs)
                self._taps_index = {k: i for i, k in enumerate(names)}
            self._taps.run()
            self._taps_version = st.shadow_version
        return self._taps.outs[self._taps_index[(s, l, name)]]

    def _dwproj_params(self, s, l):
        """Static half of the cxr_dwproj descriptors of layer (s, l): raw taps in conv layout + the BatchNorm parameters (views of the flat store)."""
        key = ("dwp", s, l)
        hit = self._fold_cache.get(key)
        if hit is None:
            cfg, st = self.cfg, self.s
            hit = []
            for n, sd in (("query", cfg.stride_q[s]), ("key", cfg.stride_kv[s]), ("value", cfg.stride_kv[s])):
                cp = self._conv_prefix(s, l, n)
                w = st.f32(cp + "convolution.weight")
                hit.append(dict(stride=sd, taps=self._raw_taps(s, l, n), w=w.view(w.shape[0], 9), gamma=st.f32(cp + "normalization.weight"),
                                beta=st.f32(cp + "normalization.bias"), run_mean=st.f32(cp + "normalization.running_mean"),
                                run_var=st.f32(cp + "normalization.running_var")))
            self._fold_cache[key] = hit
        return hit

    def _fold_train(self, h1, H, W, stride, tok0, s, l, names):
        """Train-mode BatchNorm of one (query) or two (key, value) depthwise projections of h1: batch statistics in one pass over h1,
        running statistics moved in place (two launches per call). -> folds for dwconv_fwd, {name: (mean, rstd, count)} for backward."""
        cfg, st = self.cfg, self.s
        projs = []
        for n in names:
            cp = self._conv_prefix(s, l, n)
            projs.append(dict(wt=self._raw_taps(s, l, n), w=st.f32(cp + "convolution.weight"), g=st.f32(cp + "normalization.weight"),
                              b=st.f32(cp + "normalization.bias"), run_mean=st.f32(cp + "normalization.running_mean"),
                              run_var=st.f32(cp + "normalization.running_var")))
        outs, count = ops.dwconv_bn_train_fwd_stats(h1, H, W, stride, tok0, cfg.bn_eps, cfg.bn_momentum, projs)
        return [o[0] for o in outs], {n: (o[1], o[2], count) for n, o in zip(names, outs)}

    def _prepare_transposes(self):
        """W^T (the K-contiguous operand of every dX GEMM) for the current weight version: ONE batched launch over all ~130 matrices, issued on
        the weight-gradient side stream at the start of a training forward (off the critical path); joined at the start of backward."""
        st, cfg, prep = self.s, self.cfg, self._prep
        if not self._wt_ready:
            keys, srcs = [], []
            for s in range(len(cfg.depth)):
                sp = self._stage(s)
                if s > 0:
                    keys.append(("embed", s)); srcs.append(prep[("embed", s)])
                for l in range(cfg.depth[s]):
                    lp = sp + f"layers.{l}."
                    for name in ("attention.attention.projection_query", "attention.attention.projection_key", "attention.attention.projection_value",
                                 "attention.output.dense", "intermediate.dense", "output.dense"):
                        keys.append(lp + name + ".weight"); srcs.append(st.w16(lp + name + ".weight"))
            keys.append(self.p + "projection_head.projection.weight"); srcs.append(st.w16(keys[-1]))
            sig = tuple(x.data_ptr() for x in srcs)
            if self._bt is None or self._bt_sig != sig:               # first use, or the store was re-packed (.to() / .cuda())
                self._bt, self._bt_sig, self._bt_keys = ops.BatchedTranspose(srcs), sig, keys
            with ops._on_wgrad_stream():
                self._bt.run()
            self._wt_event = ops.wgrad_mark()
            self._wt_ready = True
        for k, o in zip(self._bt_keys, self._bt.outs):
            prep[("wt", k)] = o

    def _wt(self, key):
        return self._prep[("wt", key)]

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, px: torch.Tensor, save: bool = False, train: bool | None = None, stage_outputs: list | None = None):
        """px [Bn,3,H,W] fp32 (contiguous) -> feats [Bn*tokens, projection_size] bf16, saved-activation dict | None.
        train (default: the store's nn.Module flag): batch-statistics BatchNorm + DropPath.
        stage_outputs: a list that receives each stage's output as (tokens bf16 [Bn, H*W, C] without the class token, H, W) -- CvtModel's
        `hidden_states` (TF5:cvt:430-447) in token-major form, for parity tests."""
        cfg, st = self.cfg, self.s
        train = bool(st.training) if train is None else bool(train)
        prep = self.prepare()
        if save:
            self._prepare_transposes()
        Bn = px.shape[0]
        saved = {"Bn": Bn, "stages": [], "train": train} if save else None
        self._train = train
        self._saving = bool(save)
        if self.fp8 is not None and self.fp8["version"] != st.shadow_version and not save:
            # the bf16 shadow was refreshed (train-mode BatchNorm statistics, an optimiser step on OTHER parameters ...): the e4m3 weights are
            # re-quantised only when an encoder Linear can have moved, i.e. when one of them is trainable (a frozen encoder keeps its snapshot;
            # load_state_dict() drops the fp8 state, see ParamStore.load_state_dict)
            if any(st.param(name).requires_grad for name in self.fp8["w"]):
                self._quantise_weights(self.fp8["w"].keys())
            self.fp8["version"] = st.shadow_version
        self._seed = None
        self._dp_factors = {}
        if train:
            self._bn_version += 1
            st.num_batches_tracked.add_(1)
            if any(self._drop_path_rate(s) > 0.0 for s in range(len(cfg.depth))):
                self._seed = st.next_dropout_seed()
        x = None            # [Bn, H*W, C] output of the previous stage (no class token)
        H = W = None
        for s in range(len(cfg.depth)):
            sp = self._stage(s)
            C = cfg.embed_dim[s]
            ep = sp + "embedding.convolution_embeddings."
            xin, Hin, Win = x, H, W
            col_ev = None
            fused0 = s == 0 and ("embed_pk", 0) in prep and px.shape[2] % 4 == 0 and px.shape[3] % 4 == 0 and px.shape[3] <= 384
            if fused0:
                # projection + bias + LayerNorm in one launch; e (the LayerNorm's input) and the statistics only when a backward pass follows
                col, xin = None, px
                xs = torch.empty((Bn, (px.shape[2] // 4) * (px.shape[3] // 4), C), dtype=torch.bfloat16, device=px.device)
                _, e, estats, Ho, Wo = ops.patch_embed_s1(px, prep[("embed_pk", 0)], st.f32(ep + "projection.bias"), st.f32(ep + "normalization.weight"),
                                                          st.f32(ep + "normalization.bias"), cfg.inner_layer_norm_eps, need_e=save, out=xs.view(-1, C))
                if save and _EARLY_PATCH_COL and ops.WGRAD_STREAM is not None:
                    # the im2col matrix of the PIXELS feeds the very last weight-gradient GEMM of the step; built where that GEMM is issued, its 100 us
                    # sit in the tail of the step with nothing on the main stream to hide them. It depends on the input alone: built now, on the
                    # weight-gradient stream, which has nothing else to do during the forward pass
                    with ops._on_wgrad_stream(px):
                        col = ops.im2col_pixels(px, cfg.patch_sizes[0], cfg.patch_stride[0], cfg.patch_padding[0], prep[("embed", 0)].shape[1])[0]
                        col_ev = torch.cuda.Event()
                        col_ev.record()                   # (the backward's last GEMM may read it from the main stream: see backward)
            elif s == 0:
                col, Ho, Wo = ops.im2col_pixels(px, cfg.patch_sizes[0], cfg.patch_stride[0], cfg.patch_padding[0], prep[("embed", 0)].shape[1])
                e = ops.gemm_nt(col, prep[("embed", s)], bias=st.f32(ep + "projection.bias"))
            elif self._implicit_embed and C % 4 == 0 and x.shape[2] % 64 == 0 and cfg.patch_sizes[s] == 3:
                # stage 2 / 3 embedding as an implicit GEMM: the [tokens, 9 * Cin] im2col matrix is never written (the backward pass builds it on the
                # weight-gradient stream for the one product that needs it, dW = de^T . col)
                col = None
                e, Ho, Wo = ops.gemm_nt_conv(x, H, W, cfg.patch_stride[s], cfg.patch_padding[s], prep[("embed", s)], bias=st.f32(ep + "projection.bias"))
            else:
                col, Ho, Wo = ops.im2col_tokens(x, H, W, cfg.patch_stride[s], cfg.patch_padding[s])
                e = ops.gemm_nt(col, prep[("embed", s)], bias=st.f32(ep + "projection.bias"))
            tok0 = 1 if cfg.cls_token[s] else 0
            L = tok0 + Ho * Wo
            if not fused0:
                xs = torch.empty((Bn, L, C), dtype=torch.bfloat16, device=px.device)
            if fused0:
                pass
            elif tok0:
                y, estats = ops.layernorm(e, st.f32(ep + "normalization.weight"), st.f32(ep + "normalization.bias"), cfg.inner_layer_norm_eps, need_stats=save)
                ops.copy_rows(y.view(Bn, Ho * Wo, C), xs[:, 1:, :])
                ops.bcast_row(st.f32(sp + "cls_token"), xs)
            else:
                _, estats = ops.layernorm(e, st.f32(ep + "normalization.weight"), st.f32(ep + "normalization.bias"), cfg.inner_layer_norm_eps,
                                          need_stats=save, out=xs.view(Bn * L, C))
            H, W = Ho, Wo
            ssave = {"col": col, "e": e, "estats": estats, "H": H, "W": W, "layers": [], "xin": xin if col is None else None, "Hin": Hin, "Win": Win,
                     "col_ev": col_ev} if save else None
            cur = xs
            for l in range(cfg.depth[s]):
                cur, lsave = self._layer_fwd(cur, s, l, H, W, tok0, prep, save)
                if save:
                    ssave["layers"].append(lsave)
            if save:
                saved["stages"].append(ssave)
            if tok0:
                x = torch.empty((Bn, H * W, C), dtype=torch.bfloat16, device=px.device)
                ops.copy_rows(cur[:, 1:, :], x)
            else:
                x = cur
            if stage_outputs is not None:
                stage_outputs.append((x, H, W))
        # projection head: LayerNorm(eps = config.layer_norm_eps) -> Linear(384 -> 768, no bias)
        hp = self.p + "projection_head."
        C = cfg.embed_dim[-1]
        hn, hstats = ops.layernorm(x.view(-1, C), st.f32(hp + "layer_norm.weight"), st.f32(hp + "layer_norm.bias"), cfg.layer_norm_eps, need_stats=save)
        feats = ops.gemm_nt(hn, st.w16(hp + "projection.weight"))
        if save:
            saved.update(x_last=x, hn=hn, hstats=hstats)
        return feats, saved

    def _drop_path_rate(self, s):
        """TF5 modeling_cvt.py:404-422: every layer of stage s gets linspace(0, drop_path_rate[s], depth[s])[s] -- indexed by the STAGE
        (quirk Q2; CvT-21: 0, 0, 0.1 * 2/15)."""
        hit = self._dp_rate.get(s)
        if hit is not None:
            return hit
        rate, depth = self.cfg.drop_path_rate[s], self.cfg.depth[s]
        if rate <= 0.0:
            hit = 0.0
        else:
            if depth <= s:
                raise IndexError("list index out of range")          # what the reference raises for such a depth (quirk Q2)
            hit = float(torch.linspace(0, rate, depth)[s])
        self._dp_rate[s] = hit
        return hit

    def _drop_path_scales(self, s, l, Bn):
        """Per-image factors (0 or 1/keep_prob) of the two DropPath calls of a layer, or (None, None). Sites 1000 + 2*(global layer) + {0, 1}; the
        factors of all layers of a stage come from one launch per forward (every layer of a stage has the same rate, quirk Q2)."""
        rate = self._drop_path_rate(s) if self._train else 0.0
        if rate <= 0.0:
            return None, None
        f = self._dp_factors.get(s)
        if f is None:
            gl0 = sum(self.cfg.depth[:s])
            f = self._dp_factors[s] = ops.dropout_site_factors(2 * self.cfg.depth[s], Bn, rate, self._seed, 1000 + 2 * gl0)
        return f[2 * l], f[2 * l + 1]

    def _layer_fwd(self, x, s, l, H, W, tok0, prep, save):
        cfg, st = self.cfg, self.s
        lp = self._stage(s) + f"layers.{l}."
        ap = lp + "attention.attention."
        Bn, L, C = x.shape
        nh = cfg.num_heads[s]
        x2d = x.view(Bn * L, C)
        h1, st1 = ops.layernorm(x2d, st.f32(lp + "layernorm_before.weight"), st.f32(lp + "layernorm_before.bias"), cfg.inner_layer_norm_eps, need_stats=save)
        h1 = h1.view(Bn, L, C)
        bn = None
        qc8 = kc8 = vc8 = None
        names = ("query", "key", "value")
        strides = (cfg.stride_q[s], cfg.stride_kv[s], cfg.stride_kv[s])
        if self._fused_proj and C % 64 == 0:
            # all three convolutional projections from ONE LDS-staged pass over h1 (csrc/dwproj.hip)
            if self._train:
                folds = ops.dwproj_bn_train_stats(h1, H, W, tok0, cfg.bn_eps, cfg.bn_momentum, self._dwproj_params(s, l))
                bn = {n: (f["mean"], f["rstd"], f["count"]) for n, f in zip(names, folds)}
            else:
                folds = [dict(stride=sd, taps=f[0], shift=f[1]) for sd, f in zip(strides, (self._fold_eval(s, l, n) for n in names))]
            if self.fp8 is not None and self._q8_fused and not save:
                # frozen e4m3 encoder: the three projections leave the pass as e4m3 with the input scales of the Linear layers that consume them
                qc8, kc8, vc8 = (t_.view(-1, C) for t_ in ops.dwproj_apply_q8(h1, H, W, tok0, folds, [self.fp8["a"][(ap + f"projection_{n}.weight", "in")] for n in names]))
                qc = kc = vc = None
                Lk_q8 = kc8.shape[0] // Bn
            else:
                qc, kc, vc = ops.dwproj_apply(h1, H, W, tok0, folds)
        else:
            if self._train:
                (fq,), bn = self._fold_train(h1, H, W, cfg.stride_q[s], tok0, s, l, ("query",))
                (fk, fv), bkv = self._fold_train(h1, H, W, cfg.stride_kv[s], tok0, s, l, ("key", "value"))
                bn.update(bkv)
            else:
                fq, fk, fv = (self._fold_eval(s, l, n) for n in names)
            qc, _ = ops.dwconv_bn(h1, H, W, cfg.stride_q[s], tok0, fq)
            kc, vc = ops.dwconv_bn(h1, H, W, cfg.stride_kv[s], tok0, fk, fv)
        Lk = kc.shape[1] if kc is not None else Lk_q8
        dp1, dp2 = self._drop_path_scales(s, l, Bn)
        Ch = st.w16(lp + "intermediate.dense.weight").shape[0]
        if (self.fp8 is not None or self._amax is not None) and not save:
            # e4m3 GEMMs (or their calibration pass); the MLP's hidden activation goes from the first GEMM's epilogue to the second as e4m3
            fused8 = self.fp8 is not None and self._q8_fused
            q, k, v = (self._linear(None if t_ is None else t_.view(-1, C), ap + f"projection_{n}.weight", ap + f"projection_{n}.bias", x8=t8)[0]
                       for n, t_, t8 in (("query", qc, qc8), ("key", kc, kc8), ("value", vc, vc8)))
            q, k, v = q.view(Bn, L, C), k.view(Bn, Lk, C), v.view(Bn, Lk, C)
            wo = lp + "attention.output.dense.weight"
            if fused8:      # the context leaves the attention kernel as e4m3 with the output projection's input scale
                ctx, ctx8 = None, ops.attention_q8(q, k, v, nh, C ** -0.5, self.fp8["a"][(wo, "in")]).view(-1, C)
            else:
                ctx, ctx8 = ops.attention(q, k, v, nh, C ** -0.5, need_lse=False)[0].view(-1, C), None
            x2, _ = self._linear(ctx, wo, lp + "attention.output.dense.bias", residual=x2d, row_scale=None if dp1 is None else (dp1, L, False), x8=ctx8)
            fc1 = lp + "intermediate.dense.weight"
            if fused8:      # ... and so does the second LayerNorm (bf16 copy not needed: the residual of the MLP is x2)
                h2, h28 = None, ops.layernorm_q8(x2, st.f32(lp + "layernorm_after.weight"), st.f32(lp + "layernorm_after.bias"), cfg.inner_layer_norm_eps,
                                                 self.fp8["a"][(fc1, "in")])
            else:
                h2, h28 = ops.layernorm(x2, st.f32(lp + "layernorm_after.weight"), st.f32(lp + "layernorm_after.bias"), cfg.inner_layer_norm_eps)[0], None
            fc2 = lp + "output.dense.weight"
            g, g8 = self._linear(h2, fc1, lp + "intermediate.dense.bias", act=1, out8_for=fc2 if self.fp8 is not None else None, x8=h28)
            x3, _ = self._linear(g, fc2, lp + "output.dense.bias", residual=x2, row_scale=None if dp2 is None else (dp2, L, True), x8=g8)
            return x3.view(Bn, L, C), None
        # the three linear projections in one grouped launch (the key / value GEMMs alone would leave most CUs idle)
        q, k, v = ops.gemm_nt_group([(t_.view(-1, C), st.w16(ap + f"projection_{n}.weight"), st.f32(ap + f"projection_{n}.bias"))
                                     for n, t_ in (("query", qc), ("key", kc), ("value", vc))])
        q, k, v = q.view(Bn, L, C), k.view(Bn, Lk, C), v.view(Bn, Lk, C)
        ctx, lse = ops.attention(q, k, v, nh, C ** -0.5, need_lse=save)                  # scale = embed_dim^-0.5 (quirk Q1)
        x2 = ops.gemm_nt(ctx.view(-1, C), st.w16(lp + "attention.output.dense.weight"), bias=st.f32(lp + "attention.output.dense.bias"), residual=x2d,
                         row_scale=None if dp1 is None else (dp1, L, False))              # first CvtDropPath: scales the attention branch
        h2, st2 = ops.layernorm(x2, st.f32(lp + "layernorm_after.weight"), st.f32(lp + "layernorm_after.bias"), cfg.inner_layer_norm_eps, need_stats=save)
        u = torch.empty((Bn * L, Ch), dtype=torch.bfloat16, device=x.device) if save else None
        g = ops.gemm_nt(h2, st.w16(lp + "intermediate.dense.weight"), bias=st.f32(lp + "intermediate.dense.bias"), act=1, aux=u)
        # the second CvtDropPath scales the WHOLE layer output, residual included (TF5:cvt:382-383, Q12)
        x3 = ops.gemm_nt(g, st.w16(lp + "output.dense.weight"), bias=st.f32(lp + "output.dense.bias"), residual=x2,
                         row_scale=None if dp2 is None else (dp2, L, True))
        lsave = None
        if save:
            lsave = dict(x=x, st1=st1, h1=h1, qc=qc, kc=kc, vc=vc, q=q, k=k, v=v, ctx=ctx, lse=lse, x2=x2, st2=st2, h2=h2, u=u, g=g, bn=bn, dp=(dp1, dp2))
        return x3.view(Bn, L, C), lsave

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, saved, dfeats: torch.Tensor, on_stage_done=None):
        """dfeats [Bn*tokens, projection_size] bf16. Parameter gradients are ACCUMULATED into the store's flat gradient buffer.
        on_stage_done(s): called when every gradient launch of stage s (and of the later stages and the projection head) has been issued --
        data parallelism starts the all-reduce of that stage's parameters while the earlier stages are still in backward."""
        cfg, st = self.cfg, self.s
        ops.wgrad_begin()
        prep = self.prepare()
        self._prepare_transposes()
        ops.wgrad_wait(self._wt_event)                                  # transposed weights are ready (NOT a join: the decoder's weight-gradient backlog keeps running)
        st.ensure_grads()
        Bn = saved["Bn"]
        hp = self.p + "projection_head."
        C = cfg.embed_dim[-1]
        ops.linear_bwd_weight(dfeats, saved["hn"], st.grad(hp + "projection.weight"))
        dhn = ops.gemm_nt(dfeats, self._wt(hp + "projection.weight"))
        dx = ops.layernorm_bwd(saved["x_last"].view(-1, C), dhn, st.f32(hp + "layer_norm.weight"), saved["hstats"],
                               st.grad(hp + "layer_norm.weight"), st.grad(hp + "layer_norm.bias"))
        dx = dx.view(Bn, -1, C)                                     # grad wrt stage output tokens (no class token)
        for s in reversed(range(len(cfg.depth))):
            sp = self._stage(s)
            C = cfg.embed_dim[s]
            ss = saved["stages"][s]
            H, W = ss["H"], ss["W"]
            tok0 = 1 if cfg.cls_token[s] else 0
            if tok0:
                full = torch.zeros((Bn, 1 + H * W, C), dtype=torch.bfloat16, device=dx.device)   # class-token output is discarded -> zero grad
                ops.copy_rows(dx, full[:, 1:, :])
                dx = full
            for l in reversed(range(cfg.depth[s])):
                # the layer below scales its incoming gradient by ITS second DropPath factor first thing: let this layer's last kernel write it scaled
                below = ss["layers"][l - 1]["dp"][1] if l > 0 else None
                dx = self._layer_bwd(dx, ss["layers"][l], s, l, H, W, tok0, prep, prescaled=l + 1 < cfg.depth[s] and ss["layers"][l]["dp"][1] is not None,
                                     out_scale=below)
            if tok0:
                ops.sum_row0_into(dx, st.grad(sp + "cls_token").view(-1))
                dsp = torch.empty((Bn, H * W, C), dtype=torch.bfloat16, device=dx.device)
                ops.copy_rows(dx[:, 1:, :], dsp)
                dx = dsp
            ep = sp + "embedding.convolution_embeddings."
            de = ops.layernorm_bwd(ss["e"], dx.view(-1, C), st.f32(ep + "normalization.weight"), ss["estats"],
                                   st.grad(ep + "normalization.weight"), st.grad(ep + "normalization.bias"))
            wp = prep[("embed", s)]
            dwp = torch.zeros(wp.shape, dtype=torch.float32, device=dx.device)
            col = ss["col"]
            if col is None:                                   # implicit-GEMM forward: the im2col matrix is built here, on the weight-gradient stream
                with ops._on_wgrad_stream(ss["xin"]):
                    if s == 0:
                        col = ops.im2col_pixels(ss["xin"], cfg.patch_sizes[0], cfg.patch_stride[0], cfg.patch_padding[0], wp.shape[1])[0]
                    else:
                        col = ops.im2col_tokens(ss["xin"], ss["Hin"], ss["Win"], cfg.patch_stride[s], cfg.patch_padding[s])[0]
            gw = st.grad(ep + "projection.weight")
            if (s == 0 and _TAIL_ON_MAIN and ss.get("col_ev") is not None and ops.WGRAD_STREAM is not None and not ops._WGRAD_SKIP
                    and not torch.cuda.is_current_stream_capturing()):
                # The step's LAST weight gradient (64 x 192 x 589824 at the benchmark batch, ~0.15 ms): its operand `de` is the main stream's last
                # product, and on the weight-gradient stream it queues behind that stream's backlog of the stage-1 layer -- the main stream, which has
                # nothing left to do, runs it instead, beside that backlog (profiles/r06_step_tail.txt: 0.33 ms of exposed tail per step before)
                ops.wgrad_flush()
                torch.cuda.current_stream().wait_event(ss["col_ev"])
                col.record_stream(torch.cuda.current_stream())
                ops.gemm_tn(de, col, dwp, dbias=st.grad(ep + "projection.bias"))
                gw.add_(dwp[:, :gw[0].numel()].view(gw.shape))
            else:
                ops.linear_bwd_weight(de, col, dwp, st.grad(ep + "projection.bias"))
                with ops._on_wgrad_stream(dwp):                  # same stream as the weight-gradient GEMM that fills dwp (ordered after it)
                    if s == 0:
                        gw.add_(dwp[:, :gw[0].numel()].view(gw.shape))                       # layout plumbing back to [Co,Ci,kh,kw]
                    else:
                        gw.add_(dwp.view(gw.shape[0], gw.shape[2], gw.shape[3], gw.shape[1]).permute(0, 3, 1, 2))
            if s > 0:
                dcol = ops.gemm_nt(de, self._wt(("embed", s)))
                Hp = Wp = cfg.grid(s - 1)
                dx = ops.col2im_tokens(dcol, Bn, cfg.embed_dim[s - 1], Hp, Wp, cfg.patch_stride[s], cfg.patch_padding[s])
            if on_stage_done is not None:
                on_stage_done(s)
        return None

    def _layer_bwd(self, dy, sv, s, l, H, W, tok0, prep, prescaled=False, out_scale=None):
        """prescaled: dy already carries this layer's second DropPath factor (the layer above wrote it so). out_scale: per-image factors the returned
        gradient is multiplied with (the second DropPath of the layer below), applied by the last LayerNorm-backward kernel instead of a separate pass."""
        cfg, st = self.cfg, self.s
        lp = self._stage(s) + f"layers.{l}."
        ap = lp + "attention.attention."
        Bn, L, C = dy.shape
        nh = cfg.num_heads[s]
        dy2 = dy.reshape(Bn * L, C)
        g = st.grad
        # MLP:  x3 = droppath2(x2 + W2 gelu(W1 h2 + b1) + b2)
        dp1, dp2 = sv["dp"]
        if dp2 is not None and not prescaled:
            dy2 = ops.dropout_add(dy2, None, 0.0, None, 0, L, row_scale=dp2)
        ops.linear_bwd_weight(dy2, sv["g"], g(lp + "output.dense.weight"), g(lp + "output.dense.bias"))
        du = ops.gemm_nt(dy2, self._wt(lp + "output.dense.weight"), act=2, aux=sv["u"])
        ops.linear_bwd_weight(du, sv["h2"], g(lp + "intermediate.dense.weight"), g(lp + "intermediate.dense.bias"))
        dh2 = ops.gemm_nt(du, self._wt(lp + "intermediate.dense.weight"))
        dx2 = ops.layernorm_bwd(sv["x2"], dh2, st.f32(lp + "layernorm_after.weight"), sv["st2"], g(lp + "layernorm_after.weight"),
                                g(lp + "layernorm_after.bias"), add=dy2, row_scale=None if dp1 is None else (dp1, L))
        dx2, da = dx2 if dp1 is not None else (dx2, dx2)
        # attention output projection: x2 = x + droppath(Wo ctx + bo)
        ops.linear_bwd_weight(da, sv["ctx"].view(-1, C), g(lp + "attention.output.dense.weight"), g(lp + "attention.output.dense.bias"))
        dctx = ops.gemm_nt(da, self._wt(lp + "attention.output.dense.weight")).view(Bn, L, C)
        dq, dk, dv = ops.attention_bwd(sv["q"], sv["k"], sv["v"], sv["ctx"], dctx, sv["lse"], nh, C ** -0.5)
        for name, d, inp in (("query", dq, sv["qc"]), ("key", dk, sv["kc"]), ("value", dv, sv["vc"])):
            ops.linear_bwd_weight(d.view(-1, C), inp.view(-1, C), g(ap + f"projection_{name}.weight"), g(ap + f"projection_{name}.bias"))
        # gradients wrt the three BatchNorm outputs, one grouped launch
        outs = ops.gemm_nt_group([(d.view(-1, C), self._wt(ap + f"projection_{name}.weight"), None) for name, d in (("query", dq), ("key", dk), ("value", dv))])
        dcs = {name: o.view(d.shape) for (name, d), o in zip((("query", dq), ("key", dk), ("value", dv)), outs)}
        strides = {"query": cfg.stride_q[s], "key": cfg.stride_kv[s], "value": cfg.stride_kv[s]}
        projs = []
        h1 = sv["h1"]
        fused = self._fused_proj and C % 64 == 0
        if sv["bn"] is not None and fused:
            # batch-statistics BatchNorm, all three projections per pass: (sum dy, sum dy*c) -> coefficients, dgamma, dbeta -> dc in place + raw-tap
            # gradient (tap sums ride along with the dc pass) -> dx
            names = ("query", "key", "value")
            base = self._dwproj_params(s, l)
            bp = []
            for n, b in zip(names, base):
                cp = self._conv_prefix(s, l, n)
                mean, rstd, _ = sv["bn"][n]
                bp.append(dict(stride=b["stride"], taps=b["taps"], y=dcs[n], gamma=b["gamma"], mean=mean, rstd=rstd, beta=b["beta"],
                               yf=sv[{"query": "qc", "key": "kc", "value": "vc"}[n]] if self._bwd_stats_from_y else None,
                               dgamma=g(cp + "normalization.weight"), dbeta=g(cp + "normalization.bias"), dw=g(cp + "convolution.weight").view(C, 9)))
            coefs = ops.dwproj_bn_train_bwd_stats(h1, H, W, tok0, bp)
            for b, cf in zip(bp, coefs):
                b["coef"] = cf
            ops.dwproj_dc_taps_(h1, H, W, tok0, bp)
            dh1 = ops.dwproj_dx(bp, Bn, C, H, W, tok0)
        elif sv["bn"] is not None:
            # batch-statistics BatchNorm: (sum dy, sum dy*c) per channel -> dc rewritten in place as the gradient of the RAW conv output ->
            # the ordinary dx / tap-sum kernels with the raw taps
            arena = torch.empty((30, C), dtype=torch.float32, device=dy.device)     # 3 x tap-sum outputs
            raws = {n: self._raw_taps(s, l, n) for n in strides}

            def bproj(name):
                cp = self._conv_prefix(s, l, name)
                mean, rstd, _ = sv["bn"][name]
                return dict(wt=raws[name], dy=dcs[name], g=st.f32(cp + "normalization.weight"), mean=mean, rstd=rstd,
                            dg=g(cp + "normalization.weight"), db=g(cp + "normalization.bias"))
            coefs = dict(zip(("query",), ops.dwconv_bn_train_bwd_stats(h1, H, W, strides["query"], tok0, [bproj("query")])))
            coefs.update(zip(("key", "value"), ops.dwconv_bn_train_bwd_stats(h1, H, W, strides["key"], tok0, [bproj("key"), bproj("value")])))
            for i, name in enumerate(("query", "key", "value")):
                cp = self._conv_prefix(s, l, name)
                ops.dwconv_bn_train_dc_(h1, raws[name], coefs[name], dcs[name], H, W, strides[name], tok0)
                projs.append((dcs[name], raws[name], strides[name]))
                with ops._on_wgrad_stream(h1, dcs[name], arena):
                    G2, _ = ops.dwconv_bn_bwd_w(h1, dcs[name], H, W, strides[name], tok0, out=arena[10 * i:10 * i + 10])
                    ops.tap_grad_accum(G2, g(cp + "convolution.weight"))
            dh1 = ops.dwconv_bn_bwd_dx(projs, Bn, C, H, W, tok0)
        else:
            arena = torch.empty((30, C), dtype=torch.float32, device=dy.device)
            for i, name in enumerate(("query", "key", "value")):
                cp = self._conv_prefix(s, l, name)
                wf, _ = self._fold_eval(s, l, name)
                projs.append((dcs[name], wf, strides[name]))
                with ops._on_wgrad_stream(h1, dcs[name], arena):
                    G, S = ops.dwconv_bn_bwd_w(h1, dcs[name], H, W, strides[name], tok0, out=arena[10 * i:10 * i + 10])
                    ops.bn_fold_bwd(st.f32(cp + "convolution.weight"), st.f32(cp + "normalization.weight"), st.f32(cp + "normalization.running_mean"),
                                    st.f32(cp + "normalization.running_var"), cfg.bn_eps, G, S, g(cp + "convolution.weight"),
                                    g(cp + "normalization.weight"), g(cp + "normalization.bias"))
            if fused:
                dh1 = ops.dwproj_dx([dict(stride=sd, taps=wf, y=d) for d, wf, sd in projs], Bn, C, H, W, tok0)
            else:
                dh1 = ops.dwconv_bn_bwd_dx(projs, Bn, C, H, W, tok0)
        if out_scale is not None:
            _, dx = ops.layernorm_bwd(sv["x"].view(-1, C), dh1.view(-1, C), st.f32(lp + "layernorm_before.weight"), sv["st1"],
                                      g(lp + "layernorm_before.weight"), g(lp + "layernorm_before.bias"), add=dx2, row_scale=(out_scale, L), main=False)
        else:
            dx = ops.layernorm_bwd(sv["x"].view(-1, C), dh1.view(-1, C), st.f32(lp + "layernorm_before.weight"), sv["st1"],
                                   g(lp + "layernorm_before.weight"), g(lp + "layernorm_before.bias"), add=dx2)
        return dx.view(Bn, L, C)
