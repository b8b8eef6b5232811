"""fp32 restatement of CvtWithProjectionHead / MultiCvtWithProjectionHead (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows reference modules/transformers/single_model/modelling_single.py:53-78 and
modules/transformers/multi_model/modelling_multi.py:53-87 on top of transformers' CvtModel
(TF5:cvt = transformers/models/cvt/modeling_cvt.py @ 5.15.0).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _ln(x, sd, key, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def conv_embed(x, sd, p, stride, padding, eps):
    """TF5:cvt:77-90  Conv2d -> 'b c h w -> b (h w) c' -> LayerNorm(eps=1e-5) -> back to b c h w."""
    x = F.conv2d(x, sd[p + "projection.weight"], sd[p + "projection.bias"], stride=stride, padding=padding)
    b, c, h, w = x.shape
    t = x.view(b, c, h * w).permute(0, 2, 1)
    t = _ln(t, sd, p + "normalization", eps)
    return t.permute(0, 2, 1).view(b, c, h, w)


def dw_bn(x, sd, p, stride, padding, bn_eps, bn_train=False, bn_momentum=0.0):
    """TF5:cvt:93-110 depthwise Conv2d(groups=C, bias=False) -> BatchNorm2d -> 'b c h w -> b (h w) c' (:113-119).
    bn_train: batch statistics (model.train()); bn_momentum > 0 additionally moves sd's running statistics IN PLACE (nn.BatchNorm2d
    default 0.1) and counts the batch -- pass a cloned state dict."""
    c = x.shape[1]
    y = F.conv2d(x, sd[p + "convolution.weight"], None, stride=stride, padding=padding, groups=c)
    y = F.batch_norm(y, sd[p + "normalization.running_mean"], sd[p + "normalization.running_var"],
                     sd[p + "normalization.weight"], sd[p + "normalization.bias"],
                     training=bn_train, momentum=bn_momentum, eps=bn_eps)
    if bn_train and bn_momentum > 0.0 and p + "normalization.num_batches_tracked" in sd:
        sd[p + "normalization.num_batches_tracked"] += 1
    b, c, h, w = y.shape
    return y.view(b, c, h * w).permute(0, 2, 1)


def cvt_attention(h, sd, p, cfg, s, height, width, bn_train=False, bn_momentum=0.0):
    """TF5:cvt:186-217. scale = embed_dim ** -0.5 (quirk Q1, :152)."""
    c, nh = cfg.embed_dim[s], cfg.num_heads[s]
    cls = None
    if cfg.cls_token[s]:
        cls, h = torch.split(h, [1, height * width], 1)
    b = h.shape[0]
    sp = h.permute(0, 2, 1).reshape(b, c, height, width)
    ap = p + "attention.attention."
    k = dw_bn(sp, sd, ap + "convolution_projection_key.convolution_projection.", cfg.stride_kv[s], cfg.padding_kv[s], cfg.bn_eps, bn_train,
              bn_momentum)
    q = dw_bn(sp, sd, ap + "convolution_projection_query.convolution_projection.", cfg.stride_q[s], cfg.padding_q[s], cfg.bn_eps, bn_train,
              bn_momentum)
    v = dw_bn(sp, sd, ap + "convolution_projection_value.convolution_projection.", cfg.stride_kv[s], cfg.padding_kv[s], cfg.bn_eps, bn_train,
              bn_momentum)
    if cls is not None:
        q, k, v = (torch.cat((cls, t), dim=1) for t in (q, k, v))
    hd = c // nh

    def heads(t, name):
        t = F.linear(t, sd[ap + f"projection_{name}.weight"], sd[ap + f"projection_{name}.bias"])
        return t.view(b, t.shape[1], nh, hd).permute(0, 2, 1, 3)

    q, k, v = heads(q, "query"), heads(k, "key"), heads(v, "value")
    score = torch.einsum("bhlk,bhtk->bhlt", q, k) * (c ** -0.5)
    prob = torch.softmax(score, dim=-1)
    ctx = torch.einsum("bhlt,bhtv->bhlv", prob, v)
    ctx = ctx.permute(0, 2, 1, 3).reshape(b, -1, c)
    return F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])


def cvt_layer(x, sd, p, cfg, s, height, width, bn_train=False, bn_momentum=0.0, drop_path=None):
    """TF5:cvt:365-384 (pre-LN block). drop_path = (f_attn [Bn], f_out [Bn]) per-image factors 0 | 1/keep_prob of the two CvtDropPath calls
    (TF5:cvt:297-316), None = identity (eval / rate 0). The first scales the attention branch (:372); the second is applied AFTER the
    second residual connection (:382-383), i.e. it scales the layer's whole output, residual stream included (quirk Q12)."""
    eps = cfg.inner_layer_norm_eps
    a = cvt_attention(_ln(x, sd, p + "layernorm_before", eps), sd, p, cfg, s, height, width, bn_train, bn_momentum)
    if drop_path is not None:
        a = a * drop_path[0].view(-1, 1, 1)
    x = a + x
    h = _ln(x, sd, p + "layernorm_after", eps)
    h = F.gelu(F.linear(h, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    h = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]) + x
    if drop_path is not None:
        h = h * drop_path[1].view(-1, 1, 1)
    return h


def cvt_stage(x, sd, cfg, s, prefix, bn_train=False, bn_momentum=0.0, drop_path=None):
    """TF5:cvt:430-447."""
    p = f"{prefix}cvt.encoder.stages.{s}."
    x = conv_embed(x, sd, p + "embedding.convolution_embeddings.", cfg.patch_stride[s], cfg.patch_padding[s],
                   cfg.inner_layer_norm_eps)
    b, c, height, width = x.shape
    t = x.view(b, c, height * width).permute(0, 2, 1)
    if cfg.cls_token[s]:
        t = torch.cat((sd[p + "cls_token"].expand(b, -1, -1), t), dim=1)
    for l in range(cfg.depth[s]):
        t = cvt_layer(t, sd, p + f"layers.{l}.", cfg, s, height, width, bn_train, bn_momentum,
                      None if drop_path is None else drop_path.get((s, l)))
    if cfg.cls_token[s]:
        _, t = torch.split(t, [1, height * width], 1)
    return t.permute(0, 2, 1).reshape(b, c, height, width)


def encoder_forward(pixel_values, sd, cfg, prefix="encoder.", bn_train=False, return_stages=False, bn_momentum=0.0, drop_path=None):
    """pixel_values [B,3,H,W] (single) or [B,N,3,H,W] (multi). Train mode = bn_train (+ bn_momentum to move sd's running statistics) and
    drop_path {(stage, layer): (f_attn [Bn], f_mlp [Bn])}.

    Returns (last_hidden_state [B, N*576, 768], attention_mask [B, N*576] bool or None[, per-stage NCHW activations]).
    Mask rule = reference modelling_multi.py:80: first pixel of the image != 0 (quirk Q3)."""
    multi = pixel_values.dim() == 5
    x = pixel_values.reshape(-1, *pixel_values.shape[-3:]) if multi else pixel_values
    x = x.float()
    stages = []
    for s in range(len(cfg.depth)):
        x = cvt_stage(x, sd, cfg, s, prefix, bn_train, bn_momentum, drop_path)
        stages.append(x)
    t = torch.flatten(x, 2).permute(0, 2, 1)                                       # modelling_single.py:70
    t = _ln(t, sd, prefix + "projection_head.layer_norm", cfg.layer_norm_eps)      # :29 eps=config.layer_norm_eps
    t = F.linear(t, sd[prefix + "projection_head.projection.weight"])              # :32 bias=False
    mask = None
    if multi:
        t = t.reshape(pixel_values.shape[0], -1, t.shape[-1])                       # modelling_multi.py:77
        mask = (pixel_values[:, :, 0, 0, 0] != 0.0).repeat_interleave(x.shape[-1] * x.shape[-2], dim=1)
    if return_stages:
        return t, mask, stages
    return t, mask
