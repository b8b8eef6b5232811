"""BERT decoder with cross-attention (BertLMHeadModel) on MI355X: teacher-forced forward, hand-written backward and
KV-cached decode steps over the HIP kernels. Also runs the bidirectional CXR-BERT stand-in (no cross-attention, no LM head).

Mirrors transformers' BertLMHeadModel as driven by the reference forward()
(modules/transformers/longitudinal_model/modelling_longitudinal.py:212-224; TF5 = transformers/models/bert/modeling_bert.py
@ 5.15.0: embeddings :70-108, eager attention :111-136, layers :374-411, LM head :466-496). LoRA on self-attention query/key
(modelling_longitudinal.py:163-170) is merged into the effective weight W + (alpha/r) B A before the GEMM in eval mode; under
model.train() its lora_dropout (0.1 on the branch input) forbids the merge and the rank-8 branch runs beside the base GEMM (csrc/lora.hip).

Train mode (`store.training`, i.e. after model.train()): nn.Dropout(hidden_dropout_prob) after the embedding LayerNorm and after the
attention-output / cross-attention-output / FFN-output dense layers, nn.Dropout(attention_probs_dropout_prob) on the attention
probabilities (TF5 :106,131,298,464). Masks come from the counter-based hash of csrc/common.h keyed by (seed, site, sequence, absolute
position, column): backward -- and the teacher-forced re-scoring of a sequence sampled with the KV cache -- regenerate them.
"""
from __future__ import annotations

import os

import torch

from . import ops, weights
from .config import BertConfig

BF16 = torch.bfloat16

SITE_EMBED = 1


_CROSS_KV_FUSED = os.environ.get("CXR_CROSS_KV_FUSED", "1") != "0"      # A/B switch: 0 = one K / V GEMM per layer
_LORA_IN_KERNEL = os.environ.get("CXR_LORA_IN_KERNEL", "1") != "0"        # A/B switch: 0 = separate LoRA down-projection launch per decode layer
_CROSS_KV_SHARED = os.environ.get("CXR_CROSS_KV_SHARED", "1") != "0"      # A/B switch: 0 = per-layer cross K / V projections at prefill, projected again by the re-scoring pass
_LORA_MULTI = os.environ.get("CXR_LORA_MULTI", "1") != "0"                # A/B switch: 0 = one launch per LoRA contraction in teacher-forced passes
_SELF_QKV_FUSED = os.environ.get("CXR_SELF_QKV_FUSED", "1") != "0"      # A/B switch: 0 = separate query / key / value GEMMs
_CROSS_WG_KEYS = int(os.environ.get("CXR_CROSS_WG_KEYS", "0"))          # cached cross-attention geometry (ops.attention_decode wg_keys)
_CROSS_Q_FUSED = os.environ.get("CXR_CROSS_Q_FUSED", "1") != "0"      # A/B switch: 0 = separate query GEMM launch in front of the cached cross-attention
_CROSS_MFMA = os.environ.get("CXR_CROSS_MFMA", "1") != "0"              # A/B switch: 0 = VALU decode kernel for the cached cross-attention


def _site(layer, k):
    """Dropout site ids of decoder layer `layer`: k = 0 self-attention probabilities, 1 self-attention output, 2 cross-attention
    probabilities, 3 cross-attention output, 4 FFN output, 5 / 6 LoRA input of the self-attention query / key."""
    return 16 + 8 * layer + k


class SharedCrossKV:
    """The cross-attention K / V of all layers that a decode session projected at its prefill, lent to a teacher-forced pass (the SCST
    re-scoring forward + its backward read them instead of projecting the encoder output again). The buffer is the session's own static
    storage: every later prefill of that session overwrites it. `fills` is the session's prefill count when the loan was made; `check()`
    raises once the session has decoded something else since -- the forward refuses a stale buffer, the backward refuses to differentiate
    through K / V that are no longer the ones the forward read."""

    def __init__(self, tensor, session, fills):
        self.tensor, self.session, self.fills = tensor, session, fills

    def valid(self):
        return self.session is None or self.session.fills == self.fills

    def check(self, where):
        if not self.valid():
            raise RuntimeError(f"{where}: the decode session whose cross-attention K / V this teacher-forced pass shares has run another prefill since "
                               f"(fill {self.session.fills} != {self.fills}); run backward() before the next generate() on the same geometry, or pass "
                               "cross_kv=None to project K / V again")


class KVCache:
    """Per-layer self-attention K/V [B, Tmax, D] (appended in place) and cross-attention K/V [B, S, D] (projected once)."""

    def __init__(self, layers, B, Tmax, D, device):
        self.k = [torch.empty((B, Tmax, D), dtype=BF16, device=device) for _ in range(layers)]
        self.v = [torch.empty((B, Tmax, D), dtype=BF16, device=device) for _ in range(layers)]
        self.k2 = [torch.empty((B, Tmax, D), dtype=BF16, device=device) for _ in range(layers)]   # reorder targets (beam search)
        self.v2 = [torch.empty((B, Tmax, D), dtype=BF16, device=device) for _ in range(layers)]
        self.ck = [None] * layers
        self.cv = [None] * layers
        self.kv_all = None               # optional [B, S, 2 L D]: ck / cv of all layers as its column blocks (one projection GEMM at prefill)
        self.cpk = [None] * layers       # fragment-ordered cross-attention (K, V) of the studies (ops.pack_cross_kv): the MFMA cross-attention of the cached steps
        self.cross_ready = False         # ck/cv may be pre-allocated static buffers (graph replay): filled at prefill
        self.enc_bits = None             # optional: the encoder key-padding mask as bit words (ops.pack_mask_bits), read by the cached steps
        self.len = 0
        self.Tmax = Tmax

    def reorder(self, idx):
        """self K/V <- K/V[idx] (TF5 generation/utils.py:3468-3478)."""
        for l in range(len(self.k)):
            ops.gather_batch(self.k[l], idx, self.len, self.k2[l])
            ops.gather_batch(self.v[l], idx, self.len, self.v2[l])
            self.k[l], self.k2[l] = self.k2[l], self.k[l]
            self.v[l], self.v2[l] = self.v2[l], self.v[l]


class BertEngine:
    def __init__(self, store, cfg: BertConfig, prefix: str = "decoder."):
        self.s, self.cfg = store, cfg
        self.p = prefix + ("base_model.model." if cfg.lora_r else "")
        self._prep_version = -1
        self._prep = {}
        self._wt_ready = False
        self._wtb_ready = False
        self._bt, self._bt_last = {}, None
        self._qkv_cache = {}
        self._packs, self._pack_key = {}, None

    # ------------------------------------------------------------------------------------------ parameters
    def _lin(self, base):
        """-> (weight bf16 [N,K], bias fp32 [N]) of an nn.Linear or a LoRA-wrapped Linear (merged)."""
        st = self.s
        if st.has(base + ".weight"):
            return st.w16(base + ".weight"), st.f32(base + ".bias")
        return self.prepare()[("lora", base)], st.f32(base + ".base_layer.bias")

    def prepare(self):
        st = self.s
        st.refresh_shadow()
        if self._prep_version == st.shadow_version:
            return self._prep
        prep = self._prep                      # buffers are allocated once and refreshed in place (hipGraph-captured pointers stay valid)
        cfg = self.cfg
        if cfg.lora_r:
            scale = cfg.lora_alpha / cfg.lora_r
            for l in range(cfg.num_hidden_layers):
                for name in ("query", "key"):
                    base = self.p + f"bert.encoder.layer.{l}.attention.self.{name}"
                    d = cfg.hidden_size
                    b_pad = torch.zeros((d, 32), dtype=BF16, device=st.device)
                    ops.copy_rows(st.w16(base + ".lora_B.default.weight").view(1, d, cfg.lora_r), b_pad[:, :cfg.lora_r].unsqueeze(0))
                    a_t = ops.transpose(st.w16(base + ".lora_A.default.weight"), 32)          # [d, 32]: A^T zero padded
                    w = st.f32(base + ".base_layer.weight").clone()
                    ops.gemm_nt(b_pad, a_t, out=w, out_f32=True, accumulate=True, alpha=scale)   # W + (alpha/r) B A
                    prep[("lora", base)] = ops.cast_to_bf16(w, prep.get(("lora", base)))
        self._prep, self._prep_version = prep, st.shadow_version
        self._wt_ready = False
        self._wtb_ready = False
        return prep

    def _cross_kv_all(self):
        """(W bf16 [layers*2*d, d], bias fp32 [layers*2*d], keys) of the cross-attention key / value projections of ALL layers as one matrix (they
        are stored back to back: weights.bert_param_shapes), or None (no cross-attention / CXR_CROSS_KV_FUSED=0)."""
        cfg, st = self.cfg, self.s
        if not cfg.add_cross_attention or not _CROSS_KV_FUSED:
            return None
        prefix = self.p[:-len("base_model.model.")] if cfg.lora_r else self.p
        wk, bk = weights.cross_kv_keys(cfg, prefix, ".weight"), weights.cross_kv_keys(cfg, prefix, ".bias")
        w, b = st.span(wk, "w16"), st.span(bk, "f32")
        if w is None or b is None:
            return None
        d = cfg.hidden_size
        return w.view(len(wk) * d, d), b, wk, bk

    def _self_qkv(self, l):
        """(W bf16 [3*d, d], bias fp32 [3*d], weight keys, bias keys) of layer l's self-attention query / key / value as one matrix, or None
        (LoRA-wrapped projections, parameters not adjacent, CXR_SELF_QKV_FUSED=0)."""
        cfg, st = self.cfg, self.s
        if cfg.lora_r or not _SELF_QKV_FUSED:
            return None
        hit = self._qkv_cache.get(l)
        if hit is None or hit[0] is not st.flat16:                    # (re-packed store: new flat buffers)
            wk, bk = weights.self_qkv_keys(cfg, l, self.p, ".weight"), weights.self_qkv_keys(cfg, l, self.p, ".bias")
            w, b = (st.span(wk, "w16"), st.span(bk, "f32")) if all(st.has(k) for k in wk + bk) else (None, None)
            d = cfg.hidden_size
            hit = self._qkv_cache[l] = (st.flat16, None if w is None or b is None else (w.view(3 * d, d), b, wk, bk))
        return hit[1]

    def _linear_names(self):
        cfg, p = self.cfg, self.p
        names = []
        for l in range(cfg.num_hidden_layers):
            lp = p + f"bert.encoder.layer.{l}."
            blocks = ["attention"] + (["crossattention"] if cfg.add_cross_attention else [])
            for blk in blocks:
                names += [lp + f"{blk}.self.query", lp + f"{blk}.self.key", lp + f"{blk}.self.value", lp + f"{blk}.output.dense"]
            names += [lp + "intermediate.dense", lp + "output.dense"]
        return names

    def _prepare_transposes(self, lora_tr=False):
        """W^T of every Linear (+ the padded word-embedding transpose for the LM-head dX) for the current weight version: ONE batched launch
        on the weight-gradient side stream during the training forward (off the critical path); backward joins it. lora_tr: the LoRA-wrapped
        projections additionally need the transpose of their BASE weight (train mode does not merge), kept under ("wtb", base)."""
        st, prep, p = self.s, self._prep, self.p
        mode = bool(lora_tr)
        if not (self._wt_ready and (self._wtb_ready or not mode)):
            keys, srcs, pads = [], [], []
            for base in self._linear_names():
                keys.append(("wt", base)); srcs.append(self._lin(base)[0]); pads.append(1)
                if mode and not st.has(base + ".weight"):
                    keys.append(("wtb", base)); srcs.append(st.w16(base + ".base_layer.weight")); pads.append(1)
            kv_all = self._cross_kv_all()
            if kv_all is not None:
                keys.append(("wt", "cross_kv_all")); srcs.append(kv_all[0]); pads.append(1)
            for l in range(self.cfg.num_hidden_layers):
                qkv = self._self_qkv(l)
                if qkv is not None:
                    keys.append(("wt", ("self_qkv", l))); srcs.append(qkv[0]); pads.append(1)
            if not self.cfg.cls_projection_size:
                keys.append(("wt", p + "cls.predictions.transform.dense")); srcs.append(st.w16(p + "cls.predictions.transform.dense.weight")); pads.append(1)
                keys.append(("wt", p + "bert.embeddings.word_embeddings.weight")); srcs.append(st.w16(keys[-1][1])); pads.append(64)
            sig = (mode, tuple(x.data_ptr() for x in srcs))
            bt = self._bt.get(mode)
            if bt is None or bt[0] != sig:
                bt = self._bt[mode] = (sig, ops.BatchedTranspose(srcs, pads), keys)
            with ops._on_wgrad_stream():
                bt[1].run()
            self._wt_ready = True
            self._wtb_ready = mode or self._wtb_ready
            self._bt_last = bt
        for k, o in zip(self._bt_last[2], self._bt_last[1].outs):
            prep[k] = o

    # ------------------------------------------------------------------------------------------ teacher-forced forward
    def _lora_parts(self, base):
        """(base weight bf16, bias fp32, lora_A bf16 [r,K], lora_B bf16 [N,r]) of a LoRA-wrapped Linear"""
        st = self.s
        return (st.w16(base + ".base_layer.weight"), st.f32(base + ".base_layer.bias"), st.w16(base + ".lora_A.default.weight"),
                st.w16(base + ".lora_B.default.weight"))

    def _lora_train(self, train):
        train = bool(self.s.training) if train is None else bool(train)
        return bool(self.cfg.lora_r) and train and self.cfg.lora_dropout > 0.0

    def _dropout_cfg(self, train, seed):
        """-> (p_hidden, p_attn, seed tensor) of this pass; zeros in eval mode."""
        train = bool(self.s.training) if train is None else bool(train)
        ph = float(self.cfg.hidden_dropout_prob) if train else 0.0
        pa = float(self.cfg.attention_probs_dropout_prob) if train else 0.0
        if (ph > 0.0 or pa > 0.0 or self._lora_train(train)) and seed is None:
            seed = self.s.next_dropout_seed()
        return ph, pa, seed

    def cross_kv_fused(self):
        """True when the cross-attention key / value projections of all layers are one [2 L D, D] matrix in storage order (CXR_CROSS_KV_SHARED=0: off)."""
        return _CROSS_KV_SHARED and bool(self.cfg.add_cross_attention) and self._cross_kv_all() is not None

    def forward(self, ids, enc=None, enc_mask=None, attn_mask=None, token_type_ids=None, position_ids=None, save=False, causal=True,
                lm_head=True, train=None, seed=None, logits_bf16=False, inputs_embeds=None, cross_kv=None, logit_from=0):
        """ids int64 [B,T]; enc bf16 [B,S,D] | None; masks uint8 (1 = attend). -> logits fp32 [B,T,V] (or hidden bf16 [B,T,D]), saved.
        train (default: the store's nn.Module flag) enables dropout; seed: device int32 [1] to REPRODUCE the masks of an earlier pass.
        cross_kv: [B, S, 2 L D] bf16, the cross-attention K / V of all layers already projected from THIS enc with the current weights (a decode
        session's prefill): the projection GEMM is skipped; the backward is unchanged.
        logit_from = t0 > 0: the LM head runs on positions t0 .. T-1 only -> logits [B, T - t0, V], contiguous (the SCST re-scoring pass scores the
        sampled positions, not the prompt's; backward(dlogits=[B (T - t0), V]) then)."""
        cfg, st, p = self.cfg, self.s, self.p
        self.prepare()
        lora_tr = self._lora_train(train)
        if save:
            self._prepare_transposes(lora_tr)
        word = None
        if inputs_embeds is not None:
            # BertEmbeddings with inputs_embeds (TF5:bert:84-108): the given vectors replace the word-embedding lookup; position / token-type
            # embeddings, LayerNorm and dropout are unchanged. Same kernel: the vectors are the "table", row r looks up row r.
            assert ids is None and inputs_embeds.dim() == 3, "specify exactly one of decoder_input_ids / decoder_inputs_embeds"
            word = inputs_embeds.to(BF16).contiguous().view(-1, inputs_embeds.shape[-1])
            ids = torch.arange(word.shape[0], device=word.device, dtype=torch.int64).view(inputs_embeds.shape[0], inputs_embeds.shape[1])
        B, T = ids.shape
        D, nh = cfg.hidden_size, cfg.num_attention_heads
        R = B * T
        ph, pa, seed = self._dropout_cfg(train, seed)
        pl, ls = float(cfg.lora_dropout), (cfg.lora_alpha / cfg.lora_r if cfg.lora_r else 0.0)
        e = p + "bert.embeddings."
        h, esum, estats = ops.bert_embed(ids, token_type_ids, position_ids, st.w16(e + "word_embeddings.weight") if word is None else word,
                                         st.w16(e + "token_type_embeddings.weight"), st.w16(e + "position_embeddings.weight"),
                                         st.f32(e + "LayerNorm.weight"), st.f32(e + "LayerNorm.bias"), cfg.layer_norm_eps, T, need_sum=save,
                                         drop=(ph, seed, SITE_EMBED) if ph else None)
        saved = dict(B=B, T=T, ids=ids, tt=token_type_ids, pos=position_ids, esum=esum, estats=estats, attn_mask=attn_mask, enc=enc,
                     enc_mask=enc_mask, causal=causal, layers=[], ph=ph, pa=pa, seed=seed, lora_tr=lora_tr, from_embeds=word is not None) if save else None
        scale = cfg.head_dim ** -0.5
        kv_all = None
        if cfg.add_cross_attention and enc is not None:
            kva = self._cross_kv_all()
            if kva is not None:
                # cross-attention K and V of every layer in one GEMM: [B*S, d] x [d, layers*2*d] (all layers project the same encoder output)
                S_ = enc.shape[1]
                if cross_kv is not None:
                    if isinstance(cross_kv, SharedCrossKV):
                        cross_kv.check("BertEngine.forward")
                        if save:
                            saved["cross_kv_loan"] = cross_kv
                        cross_kv = cross_kv.tensor
                    assert cross_kv.shape == (B, S_, kva[0].shape[0]) and cross_kv.dtype == BF16 and cross_kv.is_contiguous()
                    kv_all = cross_kv
                else:
                    kv_all = ops.gemm_nt(enc.reshape(B * S_, D), kva[0], bias=kva[1]).view(B, S_, -1)
                if save:
                    saved["kv_all"] = True

        def out_proj(x, w, b, resid, site):
            """dense -> dropout -> + residual (BertSelfOutput / BertOutput before their LayerNorm)"""
            return ops.gemm_nt(x, w, bias=b, residual=resid, drop=(ph, seed, site, T, 0) if ph else None)

        for l in range(cfg.num_hidden_layers):
            lp = p + f"bert.encoder.layer.{l}."
            sv = {}
            wv, bv = self._lin(lp + "attention.self.value")
            if lora_tr:
                # peft Linear under train(): base(x) + (alpha/r) * B(A(dropout(x))): base GEMM + rank-8 branch (one launch for q and k)
                wq, bq, aq, bq_l = self._lora_parts(lp + "attention.self.query"); wk, bk, ak, bk_l = self._lora_parts(lp + "attention.self.key")
                if self._lora_multi(R, D):
                    # both adapters per launch: down projections on the matrix cores, the two rank-8 updates as one launch
                    tq, tk = ops.lora_down_multi([dict(x=h, W=aq, drop=(pl, _site(l, 5))), dict(x=h, W=ak, drop=(pl, _site(l, 6)))], rows_per_b=T, seed=seed, scale=ls)
                    q, k = ops.gemm_nt(h, wq, bias=bq), ops.gemm_nt(h, wk, bias=bk)
                    ops.lora_up_add_multi_([dict(y=q, t=tq, W=bq_l, w_is_b=True), dict(y=k, t=tk, W=bk_l, w_is_b=True)])
                    q, k = q.view(B, T, D), k.view(B, T, D)
                else:
                    tq, tk = ops.lora_down(h, aq, drop0=(pl, _site(l, 5)), W1=ak, drop1=(pl, _site(l, 6)), rows_per_b=T, seed=seed, scale=ls)
                    q = ops.lora_up_add_(ops.gemm_nt(h, wq, bias=bq), tq, bq_l, True).view(B, T, D)
                    k = ops.lora_up_add_(ops.gemm_nt(h, wk, bias=bk), tk, bk_l, True).view(B, T, D)
                if save:
                    sv.update(tq=tq, tk=tk)
                v = ops.gemm_nt(h, wv, bias=bv).view(B, T, D)
            else:
                qkv_w = self._self_qkv(l)
                if qkv_w is not None:
                    # query, key and value of the layer in one GEMM: [R, d] x [d, 3d]; q / k / v are its three column blocks
                    qkv = ops.gemm_nt(h, qkv_w[0], bias=qkv_w[1]).view(B, T, 3 * D)
                    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
                    if save:
                        sv["qkv_fused"] = True
                else:
                    wq, bq = self._lin(lp + "attention.self.query"); wk, bk = self._lin(lp + "attention.self.key")
                    q = ops.gemm_nt(h, wq, bias=bq).view(B, T, D)
                    k = ops.gemm_nt(h, wk, bias=bk).view(B, T, D)
                    v = ops.gemm_nt(h, wv, bias=bv).view(B, T, D)
            ctx, lse = ops.attention(q, k, v, nh, scale, kpm=attn_mask, causal=causal, need_lse=save, drop=(pa, seed, _site(l, 0), 0))
            wo, bo = self._lin(lp + "attention.output.dense")
            a1 = out_proj(ctx.view(R, D), wo, bo, h, _site(l, 1))
            h1, s1 = ops.layernorm(a1, st.f32(lp + "attention.output.LayerNorm.weight"), st.f32(lp + "attention.output.LayerNorm.bias"),
                                   cfg.layer_norm_eps, need_stats=save)
            if save:
                sv.update(h=h, q=q, k=k, v=v, ctx=ctx, lse=lse, a1=a1, s1=s1, h1=h1)
            if cfg.add_cross_attention and enc is not None:
                S = enc.shape[1]
                cq, cbq = self._lin(lp + "crossattention.self.query"); ck, cbk = self._lin(lp + "crossattention.self.key")
                cv, cbv = self._lin(lp + "crossattention.self.value"); co, cbo = self._lin(lp + "crossattention.output.dense")
                q2 = ops.gemm_nt(h1, cq, bias=cbq).view(B, T, D)
                if kv_all is not None:
                    k2, v2 = kv_all[:, :, 2 * l * D:(2 * l + 1) * D], kv_all[:, :, (2 * l + 1) * D:(2 * l + 2) * D]
                else:
                    k2 = ops.gemm_nt(enc.view(B * S, D), ck, bias=cbk).view(B, S, D)
                    v2 = ops.gemm_nt(enc.view(B * S, D), cv, bias=cbv).view(B, S, D)
                ctx2, lse2 = ops.attention(q2, k2, v2, nh, scale, kpm=enc_mask, need_lse=save, drop=(pa, seed, _site(l, 2), 0))
                a2 = out_proj(ctx2.view(R, D), co, cbo, h1, _site(l, 3))
                h2, s2 = ops.layernorm(a2, st.f32(lp + "crossattention.output.LayerNorm.weight"), st.f32(lp + "crossattention.output.LayerNorm.bias"),
                                       cfg.layer_norm_eps, need_stats=save)
                if save:
                    sv.update(q2=q2, k2=k2, v2=v2, ctx2=ctx2, lse2=lse2, a2=a2, s2=s2, h2=h2)
            else:
                h2 = h1
            w1, b1 = self._lin(lp + "intermediate.dense"); w2, b2 = self._lin(lp + "output.dense")
            u = torch.empty((R, cfg.intermediate_size), dtype=BF16, device=ids.device) if save else None
            f = ops.gemm_nt(h2, w1, bias=b1, act=1, aux=u)
            a3 = out_proj(f, w2, b2, h2, _site(l, 4))
            h, s3 = ops.layernorm(a3, st.f32(lp + "output.LayerNorm.weight"), st.f32(lp + "output.LayerNorm.bias"), cfg.layer_norm_eps, need_stats=save)
            if save:
                sv.update(u=u, f=f, a3=a3, s3=s3, h2in=h2)
                saved["layers"].append(sv)
        if not lm_head:
            if save:
                saved["h_out"] = h
            return h.view(B, T, D), saved
        t0 = int(logit_from)
        if t0 > 0:
            assert t0 < T
            hl = torch.empty((B, T - t0, D), dtype=BF16, device=h.device)
            ops.copy_rows(h.view(B, T, D)[:, t0:, :], hl)
            hl = hl.view(B * (T - t0), D)
        else:
            hl = h
        logits, hs = self._lm_head(hl, save, logits_bf16)
        if save:
            saved.update(h_out=hl, logit_from=t0, **hs)
        return logits.view(B, T - t0, -1), saved

    def _lm_head(self, h, save, logits_bf16=False):
        """BertLMPredictionHead (TF5:bert:466-496): dense -> GELU -> LayerNorm -> tied projection + bias; logits fp32 (API default) or bf16
        (the training step: the dtype the reference's bf16 autocast gives them; the loss kernel upcasts)."""
        cfg, st, p = self.cfg, self.s, self.p
        c = p + "cls.predictions."
        tu = torch.empty((h.shape[0], cfg.hidden_size), dtype=BF16, device=h.device) if save else None
        t = ops.gemm_nt(h, st.w16(c + "transform.dense.weight"), bias=st.f32(c + "transform.dense.bias"), act=1, aux=tu)
        tn, ts = ops.layernorm(t, st.f32(c + "transform.LayerNorm.weight"), st.f32(c + "transform.LayerNorm.bias"), cfg.layer_norm_eps, need_stats=save)
        logits = ops.gemm_nt(tn, st.w16(p + "bert.embeddings.word_embeddings.weight"), bias=st.f32(c + "bias"), out_f32=not logits_bf16)
        return logits, dict(tu=tu, t=t, ts=ts, tn=tn)

    # ------------------------------------------------------------------------------------------ backward
    def _wgrad(self, base, dy, x):
        """Accumulate dW/db of a (possibly LoRA-wrapped) Linear."""
        st = self.s
        if st.has(base + ".weight"):
            ops.linear_bwd_weight(dy, x, st.grad(base + ".weight"), st.grad(base + ".bias"))
            return
        cfg = self.cfg
        scale = cfg.lora_alpha / cfg.lora_r
        dw = torch.zeros(st.f32(base + ".base_layer.weight").shape, dtype=torch.float32, device=dy.device)
        ops.linear_bwd_weight(dy, x, dw, st.grad(base + ".base_layer.bias"))
        with ops._on_wgrad_stream(dw):                       # consumers of dw: same stream as the weight-gradient GEMM that fills it
            st.grad(base + ".base_layer.weight").add_(dw)
            dw16 = ops.cast_to_bf16(dw)
            # dB[n,r] = s * sum_k dW[n,k] A[r,k] ;  dA[r,k] = s * sum_n B[n,r] dW[n,k]
            ops.gemm_nt(dw16, st.w16(base + ".lora_A.default.weight"), out=st.grad(base + ".lora_B.default.weight"), out_f32=True,
                        accumulate=True, alpha=scale)
            bt = ops.transpose(st.w16(base + ".lora_B.default.weight"))             # [r, d]
            ops.gemm_nt(bt, ops.transpose(dw16), out=st.grad(base + ".lora_A.default.weight"), out_f32=True, accumulate=True, alpha=scale)

    def _wt(self, base, lora_tr=False):
        if lora_tr and ("wtb", base) in self._prep:
            return self._prep[("wtb", base)]
        return self._prep[("wt", base)]

    def _wgrad_lora_train(self, base, dy, x, t, site, T, seed):
        """Train-mode LoRA Linear y = base(x) + t B^T, t = s * dropout(x) A^T: parameter gradients; returns dt = s * dy B  [R, 8]."""
        st, cfg = self.s, self.cfg
        s_ = cfg.lora_alpha / cfg.lora_r
        ops.linear_bwd_weight(dy, x, st.grad(base + ".base_layer.weight"), st.grad(base + ".base_layer.bias"))
        B_l, A_l = st.w16(base + ".lora_B.default.weight"), st.w16(base + ".lora_A.default.weight")
        dt = ops.lora_down(dy, B_l, w_is_b=True, scale=s_)
        with ops._on_wgrad_stream(dy, x, t, dt):
            ops.lora_outer_into(dy, t, st.grad(base + ".lora_B.default.weight"), cfg.lora_r, 1)                       # dB[n,r] += sum dy[m,n] t[m,r]
            ops.lora_outer_into(x, dt, st.grad(base + ".lora_A.default.weight"), 1, x.shape[1],                       # dA[r,k] += sum drop(x)[m,k] dt[m,r]
                                drop=(cfg.lora_dropout, site), rows_per_b=T, seed=seed)
        return dt

    @staticmethod
    def _lora_multi(R, D):
        """Teacher-forced passes take the several-problems-per-launch LoRA kernels (csrc/lora.hip, second half); CXR_LORA_MULTI=0: the per-problem ones."""
        return _LORA_MULTI and R > 256 and D % 32 == 0 and D <= 1024

    def _wgrad_lora_train_pair(self, lp, dq, dk, x, tq, tk, l, T, seed):
        """_wgrad_lora_train for the query and key adapters of a layer together: dt of both in one launch, dB / dA of both in one launch."""
        st, cfg = self.s, self.cfg
        s_ = cfg.lora_alpha / cfg.lora_r
        bq, bk = lp + "attention.self.query", lp + "attention.self.key"
        ops.linear_bwd_weight(dq, x, st.grad(bq + ".base_layer.weight"), st.grad(bq + ".base_layer.bias"))
        ops.linear_bwd_weight(dk, x, st.grad(bk + ".base_layer.weight"), st.grad(bk + ".base_layer.bias"))
        dtq, dtk = ops.lora_down_multi([dict(x=dq, W=st.w16(bq + ".lora_B.default.weight"), w_is_b=True),
                                        dict(x=dk, W=st.w16(bk + ".lora_B.default.weight"), w_is_b=True)], scale=s_)
        K, r, pl = x.shape[1], cfg.lora_r, cfg.lora_dropout
        with ops._on_wgrad_stream(dq, dk, x, tq, tk, dtq, dtk):
            ops.lora_outer_multi_into([dict(a=dq, t=tq, G=st.grad(bq + ".lora_B.default.weight"), g_ks=r, g_rs=1),                               # dB[n,r] += sum dy[m,n] t[m,r]
                                       dict(a=x, t=dtq, G=st.grad(bq + ".lora_A.default.weight"), g_ks=1, g_rs=K, drop=(pl, _site(l, 5))),        # dA[r,k] += sum drop(x)[m,k] dt[m,r]
                                       dict(a=dk, t=tk, G=st.grad(bk + ".lora_B.default.weight"), g_ks=r, g_rs=1),
                                       dict(a=x, t=dtk, G=st.grad(bk + ".lora_A.default.weight"), g_ks=1, g_rs=K, drop=(pl, _site(l, 6)))],
                                      rows_per_b=T, seed=seed)
        return dtq, dtk

    def backward(self, saved, dlogits=None, dhidden=None, need_denc=False):
        """dlogits bf16 [R, V] (row stride may be padded to a multiple of 64) or dhidden bf16 [R, D]. Accumulates parameter
        gradients into the store; returns d(enc) bf16 [B,S,D] when need_denc."""
        cfg, st, p = self.cfg, self.s, self.p
        loan = saved.get("cross_kv_loan")
        if loan is not None:
            loan.check("BertEngine.backward")                           # k2 / v2 saved below are VIEWS of a decode session's buffer
        self.prepare()
        lora_tr = bool(saved.get("lora_tr"))
        self._prepare_transposes(lora_tr)
        ops.wgrad_join()                                                # transposed weights (side stream) are ready
        ops.wgrad_begin()                                               # from here on weight-gradient kernels run beside the main stream
        st.ensure_grads()
        B, T = saved["B"], saved["T"]
        R, D, nh = B * T, cfg.hidden_size, cfg.num_attention_heads
        g = st.grad
        if dlogits is not None:
            c = p + "cls.predictions."
            V = cfg.vocab_size
            Vp = ((V + 63) // 64) * 64
            t0 = int(saved.get("logit_from", 0))
            Rl = B * (T - t0)                                                    # rows the LM head ran on (forward(logit_from=))
            assert dlogits.shape[0] == Rl, (dlogits.shape, Rl)
            if dlogits.stride(0) >= Vp and dlogits.stride(0) % 8 == 0:
                dlp = dlogits.as_strided((Rl, Vp), (dlogits.stride(0), 1))       # padded columns are zero by construction (ops.softmax_ce)
            else:
                dlp = torch.zeros((Rl, Vp), dtype=BF16, device=dlogits.device)
                ops.copy_rows(dlogits.unsqueeze(0), dlp[:, :V].unsqueeze(0))
            word = st.w16(p + "bert.embeddings.word_embeddings.weight")
            ops.gemm_tn(dlp[:, :V], saved["tn"], g(p + "bert.embeddings.word_embeddings.weight"), dbias=g(c + "bias"))
            dtn = ops.gemm_nt(dlp, self._prep[("wt", p + "bert.embeddings.word_embeddings.weight")])       # K = Vp
            dt = ops.layernorm_bwd(saved["t"], dtn, st.f32(c + "transform.LayerNorm.weight"), saved["ts"], g(c + "transform.LayerNorm.weight"),
                                   g(c + "transform.LayerNorm.bias"))
            dtu = ops.gelu_bwd(dt, saved["tu"])
            ops.linear_bwd_weight(dtu, saved["h_out"], g(c + "transform.dense.weight"), g(c + "transform.dense.bias"))
            dh = ops.gemm_nt(dtu, self._prep[("wt", c + "transform.dense")])
            if t0 > 0:                                                           # positions in front of t0 fed no logit: zero gradient there
                dfull = torch.zeros((B, T, D), dtype=BF16, device=dh.device)
                ops.copy_rows(dh.view(B, T - t0, D), dfull[:, t0:, :])
                dh = dfull.view(R, D)
        else:
            dh = dhidden
        denc = None
        enc = saved["enc"]
        if need_denc and enc is not None:
            denc = torch.zeros((enc.shape[0] * enc.shape[1], D), dtype=torch.float32, device=dh.device)
        scale = cfg.head_dim ** -0.5
        ph, pa, seed = saved.get("ph", 0.0), saved.get("pa", 0.0), saved.get("seed")
        dkv_all = None
        if saved.get("kv_all") and enc is not None:
            # dK / dV of every layer's cross-attention land in ONE [B*S, layers*2*d] matrix: one GEMM each for d(enc) and for the weight gradients
            dkv_all = torch.empty((enc.shape[0], enc.shape[1], 2 * cfg.num_hidden_layers * D), dtype=BF16, device=dh.device)
            denc = None

        def dspec(site):
            """second LayerNorm-backward output = the forward dropout mask of `site` re-applied to dx (None in eval mode)"""
            return (ph, seed, site, T, 0) if ph else None

        def undrop(d, site):
            """gradient of the dense output under `dense -> dropout -> + residual`: the forward mask re-applied to the sum's gradient"""
            return ops.dropout_add(d, None, ph, seed, site, T) if ph else d

        for l in reversed(range(cfg.num_hidden_layers)):
            lp = p + f"bert.encoder.layer.{l}."
            sv = saved["layers"][l]
            da3 = ops.layernorm_bwd(sv["a3"], dh, st.f32(lp + "output.LayerNorm.weight"), sv["s3"], g(lp + "output.LayerNorm.weight"),
                                    g(lp + "output.LayerNorm.bias"), drop=dspec(_site(l, 4)))
            da3, dd3 = da3 if ph else (da3, da3)
            self._wgrad(lp + "output.dense", dd3, sv["f"])
            du = ops.gemm_nt(dd3, self._wt(lp + "output.dense"), act=2, aux=sv["u"])
            self._wgrad(lp + "intermediate.dense", du, sv["h2in"])
            dh2 = ops.gemm_nt(du, self._wt(lp + "intermediate.dense"), residual=da3)
            if "a2" in sv:
                S = enc.shape[1]
                da2 = ops.layernorm_bwd(sv["a2"], dh2, st.f32(lp + "crossattention.output.LayerNorm.weight"), sv["s2"],
                                        g(lp + "crossattention.output.LayerNorm.weight"), g(lp + "crossattention.output.LayerNorm.bias"),
                                        drop=dspec(_site(l, 3)))
                da2, dd2 = da2 if ph else (da2, da2)
                self._wgrad(lp + "crossattention.output.dense", dd2, sv["ctx2"].view(R, D))
                dctx2 = ops.gemm_nt(dd2, self._wt(lp + "crossattention.output.dense")).view(B, T, D)
                if dkv_all is not None:
                    dq2, _, _ = ops.attention_bwd(sv["q2"], sv["k2"], sv["v2"], sv["ctx2"], dctx2, sv["lse2"], nh, scale, kpm=saved["enc_mask"],
                                                  drop=(pa, seed, _site(l, 2), 0), dk_out=dkv_all[:, :, 2 * l * D:(2 * l + 1) * D],
                                                  dv_out=dkv_all[:, :, (2 * l + 1) * D:(2 * l + 2) * D])
                    self._wgrad(lp + "crossattention.self.query", dq2.view(R, D), sv["h1"])
                else:
                    dq2, dk2, dv2 = ops.attention_bwd(sv["q2"], sv["k2"], sv["v2"], sv["ctx2"], dctx2, sv["lse2"], nh, scale, kpm=saved["enc_mask"],
                                                      drop=(pa, seed, _site(l, 2), 0))
                    self._wgrad(lp + "crossattention.self.query", dq2.view(R, D), sv["h1"])
                    self._wgrad(lp + "crossattention.self.key", dk2.view(B * S, D), enc.view(B * S, D))
                    self._wgrad(lp + "crossattention.self.value", dv2.view(B * S, D), enc.view(B * S, D))
                    if denc is not None:
                        ops.gemm_nt(dk2.view(B * S, D), self._wt(lp + "crossattention.self.key"), out=denc, out_f32=True, accumulate=True)
                        ops.gemm_nt(dv2.view(B * S, D), self._wt(lp + "crossattention.self.value"), out=denc, out_f32=True, accumulate=True)
                dh1 = ops.gemm_nt(dq2.view(R, D), self._wt(lp + "crossattention.self.query"), residual=da2)
            else:
                dh1 = dh2
            da1 = ops.layernorm_bwd(sv["a1"], dh1, st.f32(lp + "attention.output.LayerNorm.weight"), sv["s1"],
                                    g(lp + "attention.output.LayerNorm.weight"), g(lp + "attention.output.LayerNorm.bias"), drop=dspec(_site(l, 1)))
            da1, dd1 = da1 if ph else (da1, da1)
            self._wgrad(lp + "attention.output.dense", dd1, sv["ctx"].view(R, D))
            dctx = ops.gemm_nt(dd1, self._wt(lp + "attention.output.dense")).view(B, T, D)
            if sv.get("qkv_fused"):
                qkv_w = self._self_qkv(l)
                dqkv = torch.empty((B, T, 3 * D), dtype=BF16, device=dctx.device)
                ops.attention_bwd(sv["q"], sv["k"], sv["v"], sv["ctx"], dctx, sv["lse"], nh, scale, kpm=saved["attn_mask"], causal=saved["causal"],
                                  drop=(pa, seed, _site(l, 0), 0), dq_out=dqkv[:, :, :D], dk_out=dqkv[:, :, D:2 * D], dv_out=dqkv[:, :, 2 * D:])
                d2 = dqkv.view(R, 3 * D)
                ops.linear_bwd_weight(d2, sv["h"], st.span(qkv_w[2], "grad").view(3 * D, D), st.span(qkv_w[3], "grad"))
                dh = ops.gemm_nt(d2, self._prep[("wt", ("self_qkv", l))], residual=da1)
                continue
            dq, dk, dv = ops.attention_bwd(sv["q"], sv["k"], sv["v"], sv["ctx"], dctx, sv["lse"], nh, scale, kpm=saved["attn_mask"],
                                           causal=saved["causal"], drop=(pa, seed, _site(l, 0), 0))
            if lora_tr and self._lora_multi(R, D):
                dtq, dtk = self._wgrad_lora_train_pair(lp, dq.view(R, D), dk.view(R, D), sv["h"], sv["tq"], sv["tk"], l, T, seed)
            elif lora_tr:
                dtq = self._wgrad_lora_train(lp + "attention.self.query", dq.view(R, D), sv["h"], sv["tq"], _site(l, 5), T, seed)
                dtk = self._wgrad_lora_train(lp + "attention.self.key", dk.view(R, D), sv["h"], sv["tk"], _site(l, 6), T, seed)
            else:
                self._wgrad(lp + "attention.self.query", dq.view(R, D), sv["h"])
                self._wgrad(lp + "attention.self.key", dk.view(R, D), sv["h"])
            self._wgrad(lp + "attention.self.value", dv.view(R, D), sv["h"])
            t1 = ops.gemm_nt(dq.view(R, D), self._wt(lp + "attention.self.query", lora_tr), residual=da1)
            t2 = ops.gemm_nt(dk.view(R, D), self._wt(lp + "attention.self.key", lora_tr), residual=t1)
            dh = ops.gemm_nt(dv.view(R, D), self._wt(lp + "attention.self.value"), residual=t2)
            if lora_tr and self._lora_multi(R, D):                       # dx += dropout-mask * (dt A) of both LoRA branches: one read-modify-write
                pl = cfg.lora_dropout
                ops.lora_up_add_multi_([dict(y=dh, t=dtq, W=st.w16(lp + "attention.self.query.lora_A.default.weight"), drop=(pl, _site(l, 5))),
                                        dict(y=dh, t=dtk, W=st.w16(lp + "attention.self.key.lora_A.default.weight"), drop=(pl, _site(l, 6)))],
                                       rows_per_b=T, seed=seed)
            elif lora_tr:
                pl = cfg.lora_dropout
                ops.lora_up_add_(dh, dtq, st.w16(lp + "attention.self.query.lora_A.default.weight"), False, drop=(pl, _site(l, 5)), rows_per_b=T, seed=seed)
                ops.lora_up_add_(dh, dtk, st.w16(lp + "attention.self.key.lora_A.default.weight"), False, drop=(pl, _site(l, 6)), rows_per_b=T, seed=seed)
        e = p + "bert.embeddings."
        dh = undrop(dh, SITE_EMBED)
        dsum = ops.layernorm_bwd(saved["esum"], dh, st.f32(e + "LayerNorm.weight"), saved["estats"], g(e + "LayerNorm.weight"), g(e + "LayerNorm.bias"))
        if saved.get("from_embeds"):
            # the "word table" was the caller's inputs_embeds: its gradient is d(inputs_embeds) (kept in saved["d_embeds"], fp32 [B*T, D]);
            # the word-embedding parameter only receives its tied LM-head gradient
            d_emb = torch.zeros((R, D), dtype=torch.float32, device=dsum.device)
            ops.bert_embed_bwd(dsum, saved["ids"], saved["tt"], saved["pos"], d_emb, g(e + "token_type_embeddings.weight"),
                               g(e + "position_embeddings.weight"), T, 0, -1)
            saved["d_embeds"] = d_emb
        else:
            ops.bert_embed_bwd(dsum, saved["ids"], saved["tt"], saved["pos"], g(e + "word_embeddings.weight"), g(e + "token_type_embeddings.weight"),
                               g(e + "position_embeddings.weight"), T, 0, cfg.pad_token_id)
        if dkv_all is not None:
            kva = self._cross_kv_all()
            BS = enc.shape[0] * enc.shape[1]
            d2 = dkv_all.view(BS, -1)
            ops.linear_bwd_weight(d2, enc.reshape(BS, D), st.span(kva[2], "grad").view(d2.shape[1], D), st.span(kva[3], "grad"))
            return ops.gemm_nt(d2, self._prep[("wt", "cross_kv_all")]).view(enc.shape) if need_denc else None
        if denc is not None:
            return ops.cast_to_bf16(denc).view(enc.shape)
        return None

    # ------------------------------------------------------------------------------------------ cached decode
    def new_cache(self, B, Tmax, device):
        return KVCache(self.cfg.num_hidden_layers, B, Tmax, self.cfg.hidden_size, device)

    def decode(self, cache: KVCache, ids_new, enc, enc_mask, attn_mask_full, token_type_ids, position_ids, train=None, seed=None):
        """One cached step. ids_new [B,Tn] (Tn = prompt length at prefill, 1 afterwards); attn_mask_full uint8 [B, len+Tn] | None.
        Returns fp32 logits of the LAST position [B, V] (TF5 generation/utils.py:2894). In train mode (the reference decodes under
        model.train() inside its SCST training_step) the dropout masks are keyed by (sequence, ABSOLUTE position), so a later
        teacher-forced pass with the same `seed` reproduces exactly the network that sampled."""
        cfg, st, p = self.cfg, self.s, self.p
        self.prepare()
        B, Tn = ids_new.shape
        D, nh = cfg.hidden_size, cfg.num_attention_heads
        past = cache.len
        R = B * Tn
        ph, pa, seed = self._dropout_cfg(train, seed)
        e = p + "bert.embeddings."
        single = Tn == 1 and B <= 64          # one new token per row: weight-streaming GEMMs + single-query attention kernels
        lora_tr = self._lora_train(train)
        fused = single and self.fused_step_ok(cache, B, enc)
        h, _, _ = ops.bert_embed(ids_new, token_type_ids, position_ids, st.w16(e + "word_embeddings.weight"),
                                 st.w16(e + "token_type_embeddings.weight"), st.w16(e + "position_embeddings.weight"),
                                 st.f32(e + "LayerNorm.weight"), st.f32(e + "LayerNorm.bias"), cfg.layer_norm_eps, Tn, pos_offset=past,
                                 drop=(ph, seed, SITE_EMBED) if ph else None, out_dal=fused)
        scale = cfg.head_dim ** -0.5
        pl, ls = float(cfg.lora_dropout), (cfg.lora_alpha / cfg.lora_r if cfg.lora_r else 0.0)
        if fused:
            return self._decode_single_fused(cache, h, B, enc_mask, attn_mask_full, past, ph, pa, seed, lora_tr)
        lin = (lambda x, w, **kw: ops.gemm_skinny(x, w, **kw)) if single else (lambda x, w, **kw: ops.gemm_nt(x, w, **kw))

        def out_lin(x, w, b, resid, site):
            """dense -> dropout -> + residual"""
            if not ph:
                return lin(x, w, bias=b, residual=resid)
            if single:
                return ops.gemm_skinny(x, w, bias=b, residual=resid, drop=(ph, seed, site, past))
            return ops.gemm_nt(x, w, bias=b, residual=resid, drop=(ph, seed, site, Tn, past))

        for l in range(cfg.num_hidden_layers):
            lp = p + f"bert.encoder.layer.{l}."
            wv, bv = self._lin(lp + "attention.self.value")
            lq = lk = None
            if lora_tr:                                                       # train mode: base weights + rank-8 branch on dropout(h)
                wq, bq, aq, bq_l = self._lora_parts(lp + "attention.self.query"); wk, bk, ak, bk_l = self._lora_parts(lp + "attention.self.key")
                tq, tk = ops.lora_down(h, aq, drop0=(pl, _site(l, 5)), W1=ak, drop1=(pl, _site(l, 6)), rows_per_b=Tn, tpos0=past, seed=seed, scale=ls)
                lq, lk = (tq, bq_l), (tk, bk_l)
            else:
                wq, bq = self._lin(lp + "attention.self.query"); wk, bk = self._lin(lp + "attention.self.key")
            if single:                                                        # q, k, v in one launch; k/v straight into their cache rows
                q = torch.empty((B, 1, D), dtype=BF16, device=h.device)
                ops.gemm_skinny3(h, wq, bq, q.view(B, D), wk, bk, cache.k[l][:, past, :], wv, bv, cache.v[l][:, past, :], lora0=lq, lora1=lk)
            elif lora_tr:
                q = ops.lora_up_add_(ops.gemm_nt(h, wq, bias=bq), tq, bq_l, True).view(B, Tn, D)
                kf = ops.lora_up_add_(ops.gemm_nt(h, wk, bias=bk), tk, bk_l, True)
                ops.copy_rows(kf.view(B, Tn, D), cache.k[l][:, past:past + Tn, :])
                ops.copy_rows(ops.gemm_nt(h, wv, bias=bv).view(B, Tn, D), cache.v[l][:, past:past + Tn, :])
            elif Tn == 1:
                q = lin(h, wq, bias=bq).view(B, Tn, D)
                lin(h, wk, bias=bk, out=cache.k[l][:, past, :])
                lin(h, wv, bias=bv, out=cache.v[l][:, past, :])
            else:
                q = lin(h, wq, bias=bq).view(B, Tn, D)
                ops.copy_rows(ops.gemm_nt(h, wk, bias=bk).view(B, Tn, D), cache.k[l][:, past:past + Tn, :])
                ops.copy_rows(ops.gemm_nt(h, wv, bias=bv).view(B, Tn, D), cache.v[l][:, past:past + Tn, :])
            kk, vv = cache.k[l][:, :past + Tn, :], cache.v[l][:, :past + Tn, :]
            if single:
                ctx = ops.attention_decode(q, kk, vv, nh, scale, kpm=attn_mask_full, drop=(pa, seed, _site(l, 0), past))
            else:
                ctx, _ = ops.attention(q, kk, vv, nh, scale, kpm=attn_mask_full, causal=True, causal_shift=past, drop=(pa, seed, _site(l, 0), past))
            wo, bo = self._lin(lp + "attention.output.dense")
            a1 = out_lin(ctx.view(R, D), wo, bo, h, _site(l, 1))
            h1, _ = ops.layernorm(a1, st.f32(lp + "attention.output.LayerNorm.weight"), st.f32(lp + "attention.output.LayerNorm.bias"), cfg.layer_norm_eps)
            if cfg.add_cross_attention and enc is not None:
                Be, S = enc.shape[0], enc.shape[1]          # Be == B, or B/2 when two decodes of the same studies share the encoder rows
                if cache.ck[l] is None or (not cache.cross_ready and past == 0):
                    if cache.kv_all is not None:
                        if l == 0:                                       # K / V of every layer in one GEMM; ck / cv are its column blocks
                            kva = self._cross_kv_all()
                            ops.gemm_nt(enc.reshape(Be * S, D), kva[0], bias=kva[1], out=cache.kv_all.view(Be * S, -1))
                    else:
                        ck, cbk = self._lin(lp + "crossattention.self.key"); cv, cbv = self._lin(lp + "crossattention.self.value")
                        okb = cache.ck[l].view(Be * S, D) if cache.ck[l] is not None else None
                        ovb = cache.cv[l].view(Be * S, D) if cache.cv[l] is not None else None
                        cache.ck[l] = ops.gemm_nt(enc.reshape(Be * S, D), ck, bias=cbk, out=okb).view(Be, S, D)
                        cache.cv[l] = ops.gemm_nt(enc.reshape(Be * S, D), cv, bias=cbv, out=ovb).view(Be, S, D)
                    if _CROSS_MFMA and ops.attention_cross_mfma_ok(B, Be, S):      # the cached steps read fragment-ordered copies (one launch per decode and layer)
                        keep = cache.cpk[l] if (cache.cpk[l] is not None and cache.cpk[l][0].numel() == Be * S * D) else None
                        cache.cpk[l] = ops.pack_cross_kv(cache.ck[l], cache.cv[l], nh, out=keep)
                    else:
                        cache.cpk[l] = None
                cq, cbq = self._lin(lp + "crossattention.self.query"); co, cbo = self._lin(lp + "crossattention.output.dense")
                q2 = lin(h1, cq, bias=cbq).view(B, Tn, D)
                if single:
                    ctx2 = ops.attention_decode(q2, cache.ck[l], cache.cv[l], nh, scale, kpm=enc_mask, drop=(pa, seed, _site(l, 2), past))
                elif enc.shape[0] == B:
                    ctx2, _ = ops.attention(q2, cache.ck[l], cache.cv[l], nh, scale, kpm=enc_mask, drop=(pa, seed, _site(l, 2), past))
                else:                                  # shared encoder rows (sample + greedy halves of one SCST step): one pass per half
                    ctx2 = torch.empty((B, Tn, D), dtype=BF16, device=h.device)
                    Bk = enc.shape[0]
                    for g0 in range(0, B, Bk):
                        ops.attention(q2[g0:g0 + Bk], cache.ck[l], cache.cv[l], nh, scale, kpm=enc_mask, out=ctx2[g0:g0 + Bk],
                                      drop=(pa, seed, _site(l, 2), past))
                a2 = out_lin(ctx2.view(R, D), co, cbo, h1, _site(l, 3))
                h2, _ = ops.layernorm(a2, st.f32(lp + "crossattention.output.LayerNorm.weight"), st.f32(lp + "crossattention.output.LayerNorm.bias"),
                                      cfg.layer_norm_eps)
            else:
                h2 = h1
            w1, b1 = self._lin(lp + "intermediate.dense"); w2, b2 = self._lin(lp + "output.dense")
            f = lin(h2, w1, bias=b1, act=1)
            a3 = out_lin(f, w2, b2, h2, _site(l, 4))
            h, _ = ops.layernorm(a3, st.f32(lp + "output.LayerNorm.weight"), st.f32(lp + "output.LayerNorm.bias"), cfg.layer_norm_eps)
        cache.len = past + Tn
        if past == 0:
            cache.cross_ready = True
        last = h.view(B, Tn, D)[:, -1, :]                                      # [B, D] view, row stride Tn*D
        if B <= 64:
            c = p + "cls.predictions."
            t = ops.gemm_skinny(last, st.w16(c + "transform.dense.weight"), bias=st.f32(c + "transform.dense.bias"), act=1)
            tn, _ = ops.layernorm(t, st.f32(c + "transform.LayerNorm.weight"), st.f32(c + "transform.LayerNorm.bias"), cfg.layer_norm_eps)
            return ops.gemm_skinny(tn, st.w16(p + "bert.embeddings.word_embeddings.weight"), bias=st.f32(c + "bias"), out_f32=True)
        logits, _ = self._lm_head(last, False)
        return logits

    fuse_decode_layernorm = True

    def _decode_packs(self, lora_tr):
        """Per weight version: every Linear a cached decode step streams, re-laid out in MFMA-fragment order with the LayerNorm that FEEDS it
        folded in (csrc/decode_gemm.hip), the (gamma, beta) pairs of the LayerNorms that are applied to a residual operand, and the LoRA
        down-projections [A diag(gamma); A diag(beta)]. Buffers are refreshed in place: hipGraph-captured pointers stay valid."""
        st, cfg, p = self.s, self.cfg, self.p
        self.prepare()
        key = (st.shadow_version, bool(lora_tr), st.flat16.data_ptr())
        if self._pack_key == key:
            return self._packs
        if self._pack_key is not None and self._pack_key[2] != key[2]:
            self._packs = {}                                     # re-packed store (.to()/.cuda())
        pk = self._packs

        def ln(base):
            return (None, None) if base is None else (st.f32(base + ".weight"), st.f32(base + ".bias"))

        def pack(name, w, bias, fold):
            g, b = ln(fold)
            pk[name] = ops.dec_pack_weight(w, g, b, bias, out=pk.get(name))

        def rgb(base):
            t = pk.get(("rgb", base))
            if t is None:
                t = pk[("rgb", base)] = torch.empty((cfg.hidden_size, 2), dtype=torch.float32, device=st.device)
            t[:, 0].copy_(st.f32(base + ".weight")); t[:, 1].copy_(st.f32(base + ".bias"))

        L = cfg.num_hidden_layers
        for l in range(L):
            lp = p + f"bert.encoder.layer.{l}."
            prev = None if l == 0 else p + f"bert.encoder.layer.{l - 1}.output.LayerNorm"
            for nm in ("query", "key", "value"):
                base = lp + "attention.self." + nm
                if lora_tr and not st.has(base + ".weight"):         # train mode: base weight here, rank-8 branch beside it
                    w, b_ = st.w16(base + ".base_layer.weight"), st.f32(base + ".base_layer.bias")
                    g, bt = ln(prev)
                    pk[(l, "lora_" + nm)] = ops.dec_pack_lora(st.w16(base + ".lora_A.default.weight"), g, bt, out=pk.get((l, "lora_" + nm)))
                else:
                    w, b_ = self._lin(base)
                pack((l, nm), w, b_, prev)
            pack((l, "attn_out"), *self._lin(lp + "attention.output.dense"), None)
            pack((l, "cq"), *self._lin(lp + "crossattention.self.query"), lp + "attention.output.LayerNorm")
            pack((l, "cout"), *self._lin(lp + "crossattention.output.dense"), None)
            pack((l, "ffn1"), *self._lin(lp + "intermediate.dense"), lp + "crossattention.output.LayerNorm")
            pack((l, "ffn2"), *self._lin(lp + "output.dense"), None)
            for nm in ("attention.output.LayerNorm", "crossattention.output.LayerNorm", "output.LayerNorm"):
                rgb(lp + nm)
        c = p + "cls.predictions."
        pack("transform", st.w16(c + "transform.dense.weight"), st.f32(c + "transform.dense.bias"), p + f"bert.encoder.layer.{L - 1}.output.LayerNorm")
        pack("lm_head", st.w16(p + "bert.embeddings.word_embeddings.weight"), st.f32(c + "bias"), c + "transform.LayerNorm")
        self._pack_key = key
        return pk

    def fused_step_ok(self, cache, B, enc):
        """One-token steps run on the decode-step kernels of csrc/decode_gemm.hip (instantiated for BERT-base widths) once the cross-attention
        K/V exist."""
        cfg = self.cfg
        return (B <= 64 and cfg.hidden_size == 768 and cfg.intermediate_size == 3072 and self.fuse_decode_layernorm and cfg.add_cross_attention
                and enc is not None and cache.cross_ready and cfg.vocab_size % 2 == 0)

    def embed_tables(self):
        """(word, token-type, position tables bf16, LayerNorm gamma, beta fp32, eps) of BertEmbeddings"""
        st, e = self.s, self.p + "bert.embeddings."
        return (st.w16(e + "word_embeddings.weight"), st.w16(e + "token_type_embeddings.weight"), st.w16(e + "position_embeddings.weight"),
                st.f32(e + "LayerNorm.weight"), st.f32(e + "LayerNorm.bias"), self.cfg.layer_norm_eps)

    def decode_embedded(self, cache, x0, B, enc_mask, attn_mask_full, train=None, seed=None):
        """A cached one-token step whose embedding output x0 (decode activation layout) already exists (ops.decode_step_embed: the step-input
        assembly and the embeddings are one launch). Same result as decode()."""
        self.prepare()
        ph, pa, seed = self._dropout_cfg(train, seed)
        return self._decode_single_fused(cache, x0, B, enc_mask, attn_mask_full, cache.len, ph, pa, seed, self._lora_train(train))

    def refresh_decode_packs(self, train=None):
        """Bring the packed decode weights up to the current weight version / train mode (in place). Graph-replayed decode steps do not run
        the Python that would otherwise notice a stale version: DecodeSession.reset calls this before the first step of every decode."""
        cfg = self.cfg
        if self.fuse_decode_layernorm and cfg.hidden_size == 768 and cfg.intermediate_size == 3072 and cfg.add_cross_attention and cfg.vocab_size % 2 == 0:
            self._decode_packs(self._lora_train(train))

    def _decode_single_fused(self, cache, x0, B, enc_mask, attn_mask_full, past, ph, pa, seed, lora_tr=False):
        """One new token per row on the decode-step kernels of csrc/decode_gemm.hip: activations stay in the decode activation layout between
        kernels, every LayerNorm is folded into the weights of the Linear it feeds (its row statistics are published by the GEMM that produced
        its input) or applied to the residual operand in an epilogue, the LM head's transform LayerNorm included: 8 launches per layer + 2.
        x0: embedding output (already LayerNorm'ed + dropped) in the decode activation layout. Returns fp32 logits [B, V]."""
        cfg, st, p = self.cfg, self.s, self.p
        D, F, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        nh, eps, scale = cfg.num_attention_heads, cfg.layer_norm_eps, cfg.head_dim ** -0.5
        dev = x0.device
        pk = self._decode_packs(lora_tr)
        cur, cur_st, cur_ln = x0, None, None           # hidden state = cur when cur_ln is None, else LayerNorm_{cur_ln}(cur) with row statistics cur_st

        def prob(name, N, out=None, lora=None):
            wp, bc = pk[name]
            return dict(wp=wp, bc=bc, N=N, fold=cur_ln is not None and name[1] not in ("attn_out", "cout", "ffn2"), out=out, lora=lora)

        def drop(site):
            return (ph, seed, site, past) if ph else None

        def res_kw():
            return dict(residual=cur, stats=cur_st, rgb=pk[("rgb", cur_ln)], eps=eps) if cur_ln is not None else dict(residual=cur)

        lora_kw = dict(lora=(float(cfg.lora_dropout), seed, cfg.lora_alpha / cfg.lora_r, past)) if lora_tr else {}
        for l in range(cfg.num_hidden_layers):
            lp = p + f"bert.encoder.layer.{l}."
            q = torch.empty((B, D), dtype=BF16, device=dev)
            lq = lk = None
            if lora_tr:
                lq = (pk[(l, "lora_query")], st.w16(lp + "attention.self.query.lora_B.default.weight"), _site(l, 5))
                lk = (pk[(l, "lora_key")], st.w16(lp + "attention.self.key.lora_B.default.weight"), _site(l, 6))
            # q / k / v in one launch; k and v land in their KV-cache rows
            ops.dec_gemm(cur, B, D, [prob((l, "query"), D, q, lq), prob((l, "key"), D, cache.k[l][:, past, :], lk),
                                     prob((l, "value"), D, cache.v[l][:, past, :])], stats=cur_st, eps=eps, **lora_kw)
            # more than 256 cached tokens (configs[4]: a 128-token prompt): the workgroup loops over 256-key passes instead of splitting the range over
            # workgroups + a merge launch (4.8 us per layer; the looping pass costs ~2)
            ctx = ops.attention_decode(q, cache.k[l][:, :past + 1, :], cache.v[l][:, :past + 1, :], nh, scale, kpm=attn_mask_full,
                                       drop=(pa, seed, _site(l, 0), past), wg_keys=256 if past + 1 <= 256 else -256, out_dal=True)
            (a1,), st1 = ops.dec_gemm(ctx, B, D, [prob((l, "attn_out"), D)], out_stats=True, drop=drop(_site(l, 1)), **res_kw())
            cur, cur_st, cur_ln = a1, st1, lp + "attention.output.LayerNorm"
            Bkv, Tk = cache.ck[l].shape[0], cache.ck[l].shape[1]
            if (_CROSS_Q_FUSED and cache.cpk[l] is not None and (enc_mask is None or cache.enc_bits is not None) and D == 768 and nh * 64 == D and B // Bkv <= 4
                    and Tk <= 1920 and Tk % 32 == 0 and B <= 64 and cur_st is not None and cur_ln is not None):
                # the query projection runs inside the cross-attention kernel (one launch instead of two per layer: csrc/decode.hip QPROJ)
                wp, bc = pk[(l, "cq")]
                ctx2 = ops.attention_cross_mfma_q(cur, B, cur_st, eps, wp, bc, cache.cpk[l], Bkv, Tk, nh, scale,
                                                  kpm_bits=cache.enc_bits if enc_mask is not None else None, drop=(pa, seed, _site(l, 2), past), out_dal=True)
                (a2,), st2 = ops.dec_gemm(ctx2, B, D, [prob((l, "cout"), D)], out_stats=True, drop=drop(_site(l, 3)), **res_kw())
                cur, cur_st, cur_ln = a2, st2, lp + "crossattention.output.LayerNorm"
                (f,), _ = ops.dec_gemm(cur, B, D, [prob((l, "ffn1"), F)], act=1, stats=cur_st, eps=eps)
                (a3,), st3 = ops.dec_gemm(f, B, F, [prob((l, "ffn2"), D)], out_stats=True, drop=drop(_site(l, 4)), **res_kw())
                cur, cur_st, cur_ln = a3, st3, lp + "output.LayerNorm"
                continue
            q2 = torch.empty((B, D), dtype=BF16, device=dev)
            ops.dec_gemm(cur, B, D, [prob((l, "cq"), D, q2)], stats=cur_st, eps=eps)
            if cache.cpk[l] is not None and (enc_mask is None or cache.enc_bits is not None):
                # matrix-core kernel on the fragment-ordered K / V copies (Tk <= 1152, <= 4 rows per study)
                ctx2 = ops.attention_cross_mfma(q2, cache.cpk[l], cache.ck[l].shape[0], cache.ck[l].shape[1], nh, scale,
                                                kpm_bits=cache.enc_bits if enc_mask is not None else None, drop=(pa, seed, _site(l, 2), past), out_dal=True)
            else:
                ctx2 = ops.attention_decode(q2, cache.ck[l], cache.cv[l], nh, scale, kpm=enc_mask, drop=(pa, seed, _site(l, 2), past),
                                            wg_keys=_CROSS_WG_KEYS, out_dal=True, kpm_bits=cache.enc_bits if enc_mask is not None else None)
            (a2,), st2 = ops.dec_gemm(ctx2, B, D, [prob((l, "cout"), D)], out_stats=True, drop=drop(_site(l, 3)), **res_kw())
            cur, cur_st, cur_ln = a2, st2, lp + "crossattention.output.LayerNorm"
            (f,), _ = ops.dec_gemm(cur, B, D, [prob((l, "ffn1"), F)], act=1, stats=cur_st, eps=eps)
            (a3,), st3 = ops.dec_gemm(f, B, F, [prob((l, "ffn2"), D)], out_stats=True, drop=drop(_site(l, 4)), **res_kw())
            cur, cur_st, cur_ln = a3, st3, lp + "output.LayerNorm"
        cache.len = past + 1
        wp, bc = pk["transform"]
        (t,), stt = ops.dec_gemm(cur, B, D, [dict(wp=wp, bc=bc, N=D, fold=True)], act=1, stats=cur_st, eps=eps, out_stats=True)
        wp, bc = pk["lm_head"]
        (logits,), _ = ops.dec_gemm(t, B, D, [dict(wp=wp, bc=bc, N=V, fold=True)], out_f32=True, stats=stt, eps=eps)
        return logits

    # ------------------------------------------------------------------------------------------ CXR-BERT stand-in head
    def cls_projection(self, hidden, prefix=""):
        """last_hidden_state[:,0] -> Linear(768,128) -> GELU -> LayerNorm(128) -> Linear(128,128)  (assumption)."""
        st, cfg = self.s, self.cfg
        c = prefix + "cls_projection_head."
        B, T, D = hidden.shape
        cls = hidden[:, 0, :]
        x = ops.gemm_nt(cls, st.w16(c + "dense_to_hidden.weight"), bias=st.f32(c + "dense_to_hidden.bias"), act=1)
        x, _ = ops.layernorm(x, st.f32(c + "LayerNorm.weight"), st.f32(c + "LayerNorm.bias"), cfg.layer_norm_eps)
        return ops.gemm_nt(x, st.w16(c + "dense_to_output.weight"), bias=st.f32(c + "dense_to_output.bias"), out_f32=True)
