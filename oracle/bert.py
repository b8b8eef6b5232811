"""fp32 restatement of the BERT decoder (BertLMHeadModel with cross-attention) and of the CXR-BERT stand-in.
TEST INFRASTRUCTURE; see oracle/__init__.py.   TF5:bert = transformers/models/bert/modeling_bert.py @ 5.15.0.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NEG = torch.finfo(torch.float32).min     # eager attention adds finfo.min where masked (TF5 masking_utils, eager path)


def _ln(x, sd, key, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def linear(x, sd, base, lora_scale=None, lora_drop=None):
    """nn.Linear, or peft LoRA Linear: base(x) + B(A(dropout(x))) * (alpha / r)  (modelling_longitudinal.py:163-170); lora_drop = the
    factor keep/(1-p) of lora_dropout in train mode (None = eval)."""
    if base + ".weight" in sd:
        return F.linear(x, sd[base + ".weight"], sd[base + ".bias"])
    y = F.linear(x, sd[base + ".base_layer.weight"], sd[base + ".base_layer.bias"])
    xd = x if lora_drop is None else x * lora_drop
    return y + F.linear(F.linear(xd, sd[base + ".lora_A.default.weight"]), sd[base + ".lora_B.default.weight"]) * lora_scale


def _drop(x, dropout, site):
    """nn.Dropout in train mode with the mask given as data: x * factor, factor = keep / (1 - p) (dropout[site]); identity when absent."""
    if dropout is None or site not in dropout:
        return x
    return x * dropout[site]


def embeddings(ids, token_type_ids, position_ids, sd, p, eps, dropout=None):
    """TF5:bert:70-108  word + token_type + position -> LayerNorm -> dropout (site "embed")."""
    b, t = ids.shape
    if position_ids is None:
        position_ids = torch.arange(t).unsqueeze(0).expand(b, t)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(ids)
    e = sd[p + "word_embeddings.weight"][ids] + sd[p + "token_type_embeddings.weight"][token_type_ids]
    e = e + sd[p + "position_embeddings.weight"][position_ids]
    return _drop(_ln(e, sd, p + "LayerNorm", eps), dropout, "embed")


def attention(xq, xkv, sd, p, nh, add_mask, lora_scale=None, dropout=None, site=None):
    """TF5:bert:111-136,164-203,230-279  eager softmax(QK^T/sqrt(d) + mask) -> dropout (:131) -> .V"""
    b, tq, d = xq.shape
    hd = d // nh
    layer = site[0] if site is not None else None
    q = linear(xq, sd, p + "self.query", lora_scale, None if dropout is None else dropout.get((layer, "lora_q"))).view(b, tq, nh, hd).transpose(1, 2)
    k = linear(xkv, sd, p + "self.key", lora_scale, None if dropout is None else dropout.get((layer, "lora_k"))).view(b, -1, nh, hd).transpose(1, 2)
    v = linear(xkv, sd, p + "self.value", lora_scale).view(b, -1, nh, hd).transpose(1, 2)
    w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    if add_mask is not None:
        w = w + add_mask
    w = _drop(torch.softmax(w, dim=-1), dropout, site)
    o = torch.matmul(w, v).transpose(1, 2).reshape(b, tq, d)
    return o


def bert_layers(h, sd, p, cfg, self_mask, enc=None, enc_mask=None, eps=1e-12, dropout=None):
    """dropout: {site: factor} with sites (layer, "self_probs" | "self_out" | "cross_probs" | "cross_out" | "ffn_out") -- the nn.Dropout
    calls of BertSelfAttention / BertSelfOutput / BertCrossAttention / BertOutput (TF5:bert:131,298,464) in train mode."""
    lora_scale = (cfg.lora_alpha / cfg.lora_r) if cfg.lora_r else None
    for l in range(cfg.num_hidden_layers):
        lp = p + f"encoder.layer.{l}."
        a = attention(h, h, sd, lp + "attention.", cfg.num_attention_heads, self_mask, lora_scale, dropout, (l, "self_probs"))
        h = _ln(_drop(linear(a, sd, lp + "attention.output.dense"), dropout, (l, "self_out")) + h, sd, lp + "attention.output.LayerNorm", eps)
        if enc is not None:
            c = attention(h, enc, sd, lp + "crossattention.", cfg.num_attention_heads, enc_mask, None, dropout, (l, "cross_probs"))
            h = _ln(_drop(linear(c, sd, lp + "crossattention.output.dense"), dropout, (l, "cross_out")) + h, sd,
                    lp + "crossattention.output.LayerNorm", eps)
        f = F.gelu(linear(h, sd, lp + "intermediate.dense"))
        h = _ln(_drop(linear(f, sd, lp + "output.dense"), dropout, (l, "ffn_out")) + h, sd, lp + "output.LayerNorm", eps)
    return h


def decoder_forward(ids, sd, cfg, encoder_hidden_states, encoder_attention_mask=None, attention_mask=None,
                    token_type_ids=None, position_ids=None, prefix="decoder.", dropout=None):
    """Teacher-forced / no-cache forward of BertLMHeadModel (TF5:bert:851-905): logits [B,T,V] fp32.

    attention_mask [B,T] (1 = keep) is combined with the causal mask; encoder_attention_mask [B,S] likewise for
    cross-attention (reference modelling_multi.py:189-199)."""
    p = prefix + ("base_model.model." if cfg.lora_r else "")
    b, t = ids.shape
    h = embeddings(ids, token_type_ids, position_ids, sd, p + "bert.embeddings.", cfg.layer_norm_eps, dropout)
    keep = torch.tril(torch.ones(t, t, dtype=torch.bool)).view(1, 1, t, t)
    if attention_mask is not None:
        keep = keep & attention_mask.bool().view(b, 1, 1, t)
    self_mask = torch.zeros(keep.shape, dtype=torch.float32).masked_fill(~keep, NEG)
    enc_mask = None
    if encoder_attention_mask is not None:
        enc_mask = torch.zeros(b, 1, 1, encoder_attention_mask.shape[1]).masked_fill(
            ~encoder_attention_mask.bool().view(b, 1, 1, -1), NEG)
    h = bert_layers(h, sd, p + "bert.", cfg, self_mask, encoder_hidden_states.float(), enc_mask, cfg.layer_norm_eps, dropout)
    c = p + "cls.predictions."
    h = F.gelu(linear(h, sd, c + "transform.dense"))                                  # TF5:bert:466-481
    h = _ln(h, sd, c + "transform.LayerNorm", cfg.layer_norm_eps)
    return F.linear(h, sd[p + "bert.embeddings.word_embeddings.weight"], sd[c + "bias"])   # tied projection (:484-496)


def reward_embed(ids, attention_mask, sd, cfg, prefix=""):
    """CXR-BERT stand-in: bidirectional BERT-base -> last_hidden_state[:,0] -> projection head -> [B,128]
    (call site tools/rewards/cxrbert.py:42-47 takes tuple element [2] = cls_projected_embedding, quirk Q10)."""
    b, t = ids.shape
    h = embeddings(ids, None, None, sd, prefix + "bert.embeddings.", cfg.layer_norm_eps)
    mask = torch.zeros(b, 1, 1, t).masked_fill(~attention_mask.bool().view(b, 1, 1, t), NEG)
    h = bert_layers(h, sd, prefix + "bert.", cfg, mask, None, None, cfg.layer_norm_eps)
    c = prefix + "cls_projection_head."
    x = F.gelu(linear(h[:, 0], sd, c + "dense_to_hidden"))
    x = _ln(x, sd, c + "LayerNorm", cfg.layer_norm_eps)
    return linear(x, sd, c + "dense_to_output")


def reward_cosine(pred_ids, pred_mask, label_ids, label_mask, sd, cfg):
    """tools/rewards/cxrbert.py:66-71."""
    return F.cosine_similarity(reward_embed(pred_ids, pred_mask, sd, cfg), reward_embed(label_ids, label_mask, sd, cfg))
