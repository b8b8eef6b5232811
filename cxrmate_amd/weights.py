"""Parameter naming (HuggingFace state-dict keys) and deterministic random initialisation.

Key names follow what the reference's checkpoints contain (`*_to_hub.ipynb` cell 6
"<All keys matched successfully>"): `encoder.cvt.encoder.stages.N...`, `encoder.projection_head.{layer_norm,projection}`,
`decoder.bert...`, `decoder.cls.predictions...`; with LoRA the decoder keys gain the peft nesting
`decoder.base_model.model...` and `query|key` become `{base_layer, lora_A.default, lora_B.default}`
(reference modules/transformers/longitudinal_model/modelling_longitudinal.py:163-171).

`init_state_dict` is a *seeded, platform-independent* initialiser (CPU torch.Generator): the golden fixtures under
tests/golden were produced by loading exactly these tensors into the reference model, so the same call on the GPU box
reproduces the reference's weights without shipping them.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

from .config import BertConfig, CvtConfig, EncoderDecoderConfig


# ----------------------------------------------------------------------------------------------- shapes
def cvt_param_shapes(cfg: CvtConfig, prefix: str = "encoder.") -> "OrderedDict[str, Tuple[int, ...]]":
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for s, depth in enumerate(cfg.depth):
        c = cfg.embed_dim[s]
        cin = cfg.num_channels if s == 0 else cfg.embed_dim[s - 1]
        k = cfg.patch_sizes[s]
        st = f"{prefix}cvt.encoder.stages.{s}."
        if cfg.cls_token[s]:
            out[st + "cls_token"] = (1, 1, cfg.embed_dim[-1])
        out[st + "embedding.convolution_embeddings.projection.weight"] = (c, cin, k, k)
        out[st + "embedding.convolution_embeddings.projection.bias"] = (c,)
        out[st + "embedding.convolution_embeddings.normalization.weight"] = (c,)
        out[st + "embedding.convolution_embeddings.normalization.bias"] = (c,)
        for l in range(depth):
            lp = st + f"layers.{l}."
            for name in ("query", "key", "value"):
                cp = lp + f"attention.attention.convolution_projection_{name}.convolution_projection."
                out[cp + "convolution.weight"] = (c, 1, cfg.kernel_qkv[s], cfg.kernel_qkv[s])
                out[cp + "normalization.weight"] = (c,)
                out[cp + "normalization.bias"] = (c,)
                out[cp + "normalization.running_mean"] = (c,)
                out[cp + "normalization.running_var"] = (c,)
                out[cp + "normalization.num_batches_tracked"] = ()
            for name in ("query", "key", "value"):
                out[lp + f"attention.attention.projection_{name}.weight"] = (c, c)
                out[lp + f"attention.attention.projection_{name}.bias"] = (c,)
            out[lp + "attention.output.dense.weight"] = (c, c)
            out[lp + "attention.output.dense.bias"] = (c,)
            h = int(c * cfg.mlp_ratio[s])
            out[lp + "intermediate.dense.weight"] = (h, c)
            out[lp + "intermediate.dense.bias"] = (h,)
            out[lp + "output.dense.weight"] = (c, h)
            out[lp + "output.dense.bias"] = (c,)
            out[lp + "layernorm_before.weight"] = (c,)
            out[lp + "layernorm_before.bias"] = (c,)
            out[lp + "layernorm_after.weight"] = (c,)
            out[lp + "layernorm_after.bias"] = (c,)
    out[prefix + "projection_head.layer_norm.weight"] = (cfg.embed_dim[-1],)
    out[prefix + "projection_head.layer_norm.bias"] = (cfg.embed_dim[-1],)
    out[prefix + "projection_head.projection.weight"] = (cfg.projection_size, cfg.embed_dim[-1])
    return out


def _lin(out, base, n, k, lora_r=0):
    if lora_r:
        out[base + ".base_layer.weight"] = (n, k)
        out[base + ".base_layer.bias"] = (n,)
        out[base + ".lora_A.default.weight"] = (lora_r, k)
        out[base + ".lora_B.default.weight"] = (n, lora_r)
    else:
        out[base + ".weight"] = (n, k)
        out[base + ".bias"] = (n,)


def bert_param_shapes(cfg: BertConfig, prefix: str = "decoder.", storage_order: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """BertLMHeadModel (decoder) or the CXR-BERT stand-in (cls_projection_size > 0, no LM head)."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    p = prefix + ("base_model.model." if cfg.lora_r else "")
    d, f = cfg.hidden_size, cfg.intermediate_size
    e = p + "bert.embeddings."
    out[e + "word_embeddings.weight"] = (cfg.vocab_size, d)
    out[e + "position_embeddings.weight"] = (cfg.max_position_embeddings, d)
    out[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, d)
    out[e + "LayerNorm.weight"] = (d,)
    out[e + "LayerNorm.bias"] = (d,)
    for l in range(cfg.num_hidden_layers):
        lp = p + f"bert.encoder.layer.{l}."
        blocks = ["attention"] + (["crossattention"] if cfg.add_cross_attention else [])
        for blk in blocks:
            for name in ("query", "key", "value"):
                r = cfg.lora_r if (blk == "attention" and name in ("query", "key")) else 0
                _lin(out, lp + f"{blk}.self.{name}", d, d, r)
            _lin(out, lp + f"{blk}.output.dense", d, d)
            out[lp + f"{blk}.output.LayerNorm.weight"] = (d,)
            out[lp + f"{blk}.output.LayerNorm.bias"] = (d,)
        _lin(out, lp + "intermediate.dense", f, d)
        _lin(out, lp + "output.dense", d, f)
        out[lp + "output.LayerNorm.weight"] = (d,)
        out[lp + "output.LayerNorm.bias"] = (d,)
    if cfg.cls_projection_size:
        c = prefix + "cls_projection_head."
        out[c + "dense_to_hidden.weight"] = (cfg.cls_projection_size, d)
        out[c + "dense_to_hidden.bias"] = (cfg.cls_projection_size,)
        out[c + "LayerNorm.weight"] = (cfg.cls_projection_size,)
        out[c + "LayerNorm.bias"] = (cfg.cls_projection_size,)
        out[c + "dense_to_output.weight"] = (cfg.cls_projection_size, cfg.cls_projection_size)
        out[c + "dense_to_output.bias"] = (cfg.cls_projection_size,)
    else:
        c = p + "cls.predictions."
        out[c + "bias"] = (cfg.vocab_size,)
        out[c + "transform.dense.weight"] = (d, d)
        out[c + "transform.dense.bias"] = (d,)
        out[c + "transform.LayerNorm.weight"] = (d,)
        out[c + "transform.LayerNorm.bias"] = (d,)
        # tied copies present in HF state dicts (TF5 models/bert/modeling_bert.py:778-781):
        out[c + "decoder.weight"] = (cfg.vocab_size, d)
        out[c + "decoder.bias"] = (cfg.vocab_size,)
    if storage_order and (cfg.add_cross_attention or not cfg.lora_r):       # (random initialisation iterates the HF order: storage_order=False)
        # storage order only (names are unchanged): the cross-attention key / value projections of ALL layers sit back to back -- K0 V0 K1 V1 ...,
        # weights then biases -- so that the flat parameter / shadow / gradient buffers expose them as ONE [layers*2*d, d] matrix: every layer
        # projects the same encoder output, which makes them one GEMM forward, one for the encoder-output gradient and one for the weight gradient
        front = (cross_kv_keys(cfg, prefix, ".weight") + cross_kv_keys(cfg, prefix, ".bias")) if cfg.add_cross_attention else []
        if not cfg.lora_r:
            # ... and the self-attention query / key / value of a layer (weights, then biases): one [3*d, d] projection of the layer input
            for l in range(cfg.num_hidden_layers):
                front += self_qkv_keys(cfg, l, prefix, ".weight") + self_qkv_keys(cfg, l, prefix, ".bias")
        re = OrderedDict((k, out[k]) for k in front)
        for k, v in out.items():
            if k not in re:
                re[k] = v
        out = re
    return out


def self_qkv_keys(cfg: BertConfig, layer: int, prefix: str = "decoder.", suffix: str = ".weight"):
    p = prefix + ("base_model.model." if cfg.lora_r else "")
    return [p + f"bert.encoder.layer.{layer}.attention.self.{n}{suffix}" for n in ("query", "key", "value")]


def cross_kv_keys(cfg: BertConfig, prefix: str = "decoder.", suffix: str = ".weight"):
    p = prefix + ("base_model.model." if cfg.lora_r else "")
    return [p + f"bert.encoder.layer.{l}.crossattention.self.{n}{suffix}" for l in range(cfg.num_hidden_layers) for n in ("key", "value")]


def tied_aliases(cfg: BertConfig, prefix: str = "decoder.") -> Dict[str, str]:
    """alias key -> canonical key (the LM projection shares storage with the word embeddings)."""
    if cfg.cls_projection_size:
        return {}
    p = prefix + ("base_model.model." if cfg.lora_r else "")
    return {
        p + "cls.predictions.decoder.weight": p + "bert.embeddings.word_embeddings.weight",
        p + "cls.predictions.decoder.bias": p + "cls.predictions.bias",
    }


def encoder_decoder_param_shapes(cfg: EncoderDecoderConfig, storage_order: bool = True):
    out = cvt_param_shapes(cfg.encoder)
    out.update(bert_param_shapes(cfg.decoder, storage_order=storage_order))
    return out


BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def is_buffer(key: str) -> bool:
    return key.endswith(BUFFER_SUFFIXES)


# ----------------------------------------------------------------------------------------------- init
def _fill(key: str, shape, g: torch.Generator, std: float, perturb: float) -> torch.Tensor:
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)

    def noise(scale):
        return torch.empty(shape, dtype=torch.float32).normal_(0.0, 1.0, generator=g).clamp_(-2.0, 2.0) * scale

    leaf = key.rsplit(".", 1)[-1]
    is_norm = any(t in key for t in ("LayerNorm", "layer_norm", "layernorm_", "normalization"))
    if leaf == "running_mean":
        return noise(perturb)
    if leaf == "running_var":
        return 1.0 + noise(perturb).abs()
    if is_norm and leaf == "weight":
        return 1.0 + noise(perturb)
    if is_norm and leaf == "bias":
        return noise(perturb)
    if leaf == "bias":
        return noise(perturb)            # HF zero-inits biases; `perturb` exercises the bias path in parity tests
    if "lora_B" in key:
        return noise(perturb)            # peft zero-inits B
    if key.endswith("convolution.weight"):
        return noise(max(std, 0.2 if perturb else std))   # depthwise 3x3: keep a visible signal under BN
    return noise(std)                    # Linear / Conv2d / Embedding / cls_token / lora_A


def init_state_dict(shapes, seed: int = 0, std: float = 0.02, perturb: float = 0.0,
                    aliases: Dict[str, str] | None = None, pad_row_zero: List[str] | None = None):
    """Deterministic fp32 CPU state dict for `shapes` (ordered). `perturb` > 0 randomises norm scales/biases/BN stats
    so that parity tests exercise every term (HF init leaves them at identity)."""
    aliases = aliases or {}
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in shapes.items():
        if key in aliases:
            continue
        sd[key] = _fill(key, tuple(shape), g, std, perturb)
    for key in (pad_row_zero or []):
        sd[key][0].zero_()
    for alias, canon in aliases.items():
        if alias in shapes:
            sd[alias] = sd[canon]
    return sd


def init_encoder_decoder(cfg: EncoderDecoderConfig, seed: int = 0, perturb: float = 0.0):
    shapes = encoder_decoder_param_shapes(cfg, storage_order=False)
    aliases = tied_aliases(cfg.decoder)
    p = "decoder." + ("base_model.model." if cfg.decoder.lora_r else "")
    return init_state_dict(shapes, seed=seed, std=cfg.decoder.initializer_range, perturb=perturb, aliases=aliases,
                           pad_row_zero=[p + "bert.embeddings.word_embeddings.weight"])


def init_reward(cfg: BertConfig, seed: int = 1, perturb: float = 0.0):
    shapes = bert_param_shapes(cfg, prefix="", storage_order=False)
    return init_state_dict(shapes, seed=seed, std=cfg.initializer_range, perturb=perturb,
                           pad_row_zero=["bert.embeddings.word_embeddings.weight"])
