"""Static shape description of the CXRMate hot path.

The numbers mirror what the reference builds at run time:
  * encoder  = CvT-21 @ 384x384 ('microsoft/cvt-21-384-22k', reference modules/lightning_modules/single.py:214)
    wrapped by CvtWithProjectionHead (modules/transformers/single_model/modelling_single.py:43-78)
  * decoder  = BertLMHeadModel, 6 layers, vocab 30000, type_vocab 2 (single.py:207-213,
    modules/transformers/multi_tf_model_to_hub.ipynb cell 5)
  * reward   = CXR-BERT stand-in (BERT-base + CLS projection head)

There is no dependency on `transformers` here: the product path only needs the shapes.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass
class CvtConfig:
    num_channels: int = 3
    image_size: int = 384
    patch_sizes: Tuple[int, ...] = (7, 3, 3)
    patch_stride: Tuple[int, ...] = (4, 2, 2)
    patch_padding: Tuple[int, ...] = (2, 1, 1)
    embed_dim: Tuple[int, ...] = (64, 192, 384)
    num_heads: Tuple[int, ...] = (1, 3, 6)
    depth: Tuple[int, ...] = (1, 4, 16)            # CvT-21
    mlp_ratio: Tuple[float, ...] = (4.0, 4.0, 4.0)
    cls_token: Tuple[bool, ...] = (False, False, True)
    kernel_qkv: Tuple[int, ...] = (3, 3, 3)
    padding_kv: Tuple[int, ...] = (1, 1, 1)
    stride_kv: Tuple[int, ...] = (2, 2, 2)
    padding_q: Tuple[int, ...] = (1, 1, 1)
    stride_q: Tuple[int, ...] = (1, 1, 1)
    initializer_range: float = 0.02
    layer_norm_eps: float = 1e-12                  # used ONLY by the projection head (modelling_single.py:29)
    inner_layer_norm_eps: float = 1e-5             # nn.LayerNorm default inside CvtModel (TF5 modeling_cvt.py:79,363-364)
    bn_eps: float = 1e-5                           # nn.BatchNorm2d default (TF5 modeling_cvt.py:105)
    bn_momentum: float = 0.1                       # nn.BatchNorm2d default: running-stat update in train mode
    # train-mode stochastic regularisers, CvtConfig defaults of TF5 configuration_cvt.py (drop_rate / attention_drop_rate are 0 everywhere)
    drop_path_rate: Tuple[float, ...] = (0.0, 0.0, 0.1)   # per stage: layer l uses linspace(0, rate, depth)[l] (TF5 modeling_cvt.py:416-418)
    projection_size: int = 768

    def grid(self, stage: int) -> int:
        """Side length of the token grid after `stage` (0-based) conv embeddings."""
        s = self.image_size
        for i in range(stage + 1):
            s = (s + 2 * self.patch_padding[i] - self.patch_sizes[i]) // self.patch_stride[i] + 1
        return s

    def kv_grid(self, stage: int) -> int:
        s = self.grid(stage)
        return (s + 2 * self.padding_kv[stage] - self.kernel_qkv[stage]) // self.stride_kv[stage] + 1

    @property
    def tokens_per_image(self) -> int:
        g = self.grid(len(self.depth) - 1)
        return g * g


@dataclass
class BertConfig:
    vocab_size: int = 30000
    hidden_size: int = 768
    num_hidden_layers: int = 6
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 0                          # BertConfig default -> nn.Embedding(padding_idx=0)
    initializer_range: float = 0.02
    hidden_dropout_prob: float = 0.1               # BertConfig defaults; active under model.train() only
    attention_probs_dropout_prob: float = 0.1
    is_decoder: bool = True
    add_cross_attention: bool = True
    # LoRA (longitudinal model only; modelling_longitudinal.py:163-170)
    lora_r: int = 0
    lora_alpha: int = 32
    lora_dropout: float = 0.1
    # CXR-BERT stand-in only: CLS projection head 768 -> 128
    cls_projection_size: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class EncoderDecoderConfig:
    encoder: CvtConfig = field(default_factory=CvtConfig)
    decoder: BertConfig = field(default_factory=BertConfig)

    def to_dict(self):
        return asdict(self)


def _get(obj, name, default):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def from_hf_config(config) -> "EncoderDecoderConfig":
    """An EncoderDecoderConfig from whatever the reference hands its model classes: an HF `VisionEncoderDecoderConfig` (duck-typed: anything
    with `.encoder` / `.decoder` sub-configs, or their dicts), as built by the Lightning modules (reference modules/lightning_modules/
    single.py:205-216: BertConfig(vocab_size, num_hidden_layers, type_vocab_size) + CvtWithProjectionHeadConfig(projection_size) on the
    'microsoft/cvt-21-384-22k' defaults). Fields this build does not vary must hold the values it is instantiated for."""
    if isinstance(config, EncoderDecoderConfig):
        return config
    enc, dec = _get(config, "encoder", None), _get(config, "decoder", None)
    if enc is None or dec is None:
        raise ValueError(f"Config: {config} has to be of type {EncoderDecoderConfig} (or an HF VisionEncoderDecoderConfig)")
    ours_e, ours_d = CvtConfig(), BertConfig()
    e = CvtConfig(**{f: (tuple(v) if isinstance(v, (list, tuple)) else v)
                     for f in ("num_channels", "patch_sizes", "patch_stride", "patch_padding", "embed_dim", "num_heads", "depth", "mlp_ratio",
                               "cls_token", "kernel_qkv", "padding_kv", "stride_kv", "padding_q", "stride_q", "initializer_range", "layer_norm_eps",
                               "drop_path_rate", "projection_size")
                     for v in [_get(enc, f, getattr(ours_e, f))]})
    e.image_size = int(_get(enc, "image_size", ours_e.image_size) or ours_e.image_size)
    d = BertConfig(**{f: _get(dec, f, getattr(ours_d, f))
                      for f in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "max_position_embeddings",
                                "type_vocab_size", "layer_norm_eps", "pad_token_id", "initializer_range", "hidden_dropout_prob",
                                "attention_probs_dropout_prob")})
    d.is_decoder = bool(_get(dec, "is_decoder", True))
    d.add_cross_attention = bool(_get(dec, "add_cross_attention", True))
    if _get(dec, "hidden_act", "gelu") != "gelu" or _get(dec, "position_embedding_type", "absolute") != "absolute":
        raise ValueError("only BERT with GELU and absolute position embeddings is implemented (the reference's decoder)")
    if d.hidden_size != 768 or d.num_attention_heads * 64 != d.hidden_size or any(c % 64 for c in e.embed_dim):
        raise ValueError("the MI355X kernels are instantiated for head_dim 64 (CvT-13/21, BERT-base widths)")
    return EncoderDecoderConfig(encoder=e, decoder=d)


def reward_config() -> BertConfig:
    """CXR-BERT-specialized stand-in: BERT-base, vocab 30522, CLS projection 128 (assumption)."""
    return BertConfig(vocab_size=30522, num_hidden_layers=12, is_decoder=False, add_cross_attention=False,
                      cls_projection_size=128)


def tiny_config(vocab_size: int = 1000, decoder_layers: int = 2, depth=(1, 2, 3), image_size: int = 96,
                lora_r: int = 0) -> EncoderDecoderConfig:
    """Full-width (head_dim 64 everywhere) but shallow configuration used by parity tests and fixtures."""
    return EncoderDecoderConfig(
        encoder=CvtConfig(depth=tuple(depth), image_size=image_size),
        decoder=BertConfig(vocab_size=vocab_size, num_hidden_layers=decoder_layers, lora_r=lora_r),
    )
