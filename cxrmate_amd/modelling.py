"""Drop-in counterparts of the reference's encoder-to-decoder models, running on the MI355X engines.

    reference class (modules/transformers/...)                         here
    single_model/modelling_single.py   SingleCXREncoderDecoderModel  -> SingleCXREncoderDecoderModel
    multi_model/modelling_multi.py     MultiCXREncoderDecoderModel   -> MultiCXREncoderDecoderModel
    longitudinal_model/modelling_longitudinal.py
                     LongitudinalPromptMultiCXREncoderDecoderModel   -> LongitudinalPromptMultiCXREncoderDecoderModel

Same surface as the reference uses from its Lightning modules: `.encoder(pixel_values)`, `.decoder`,
`forward(pixel_values=|encoder_outputs=, decoder_input_ids=, decoder_attention_mask=, decoder_token_type_ids=,
[decoder_position_ids=]).logits` (differentiable), `generate(...)` / `generate.__wrapped__(...)`, the token helpers, HF
state-dict key names. There is no PyTorch fallback for the math: every op goes through libcxrmate_hip.so.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
from torch import nn

from . import ops, weights
from .config import EncoderDecoderConfig, from_hf_config
from .decoder import BertEngine
from .encoder import CvtEncoderEngine
from .generation import GenerationMixin
from .store import ParamStore, _Node
from .token_helpers import TokenHelpers


class ModelOutput(dict):
    """Minimal stand-in for transformers' ModelOutput: attribute, key and integer access; item assignment
    (the SCST caller does `sample['sequences'] = ...`, reference scst/gt_prompt.py:185-186)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)


# ---------------------------------------------------------------------------------------------------- autograd bridges
class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, px, *params):
        feats, saved = model._enc.forward(px, save=True)
        ctx.model, ctx.saved, ctx.nparams = model, saved, len(params)
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        model = ctx.model
        if not model.direct_grads:
            model.zero_grads_prefix("encoder.")
        model._enc.backward(ctx.saved, dfeats.contiguous())
        ctx.saved = None
        return (None, None) + model._collect_grads("encoder.", ctx.nparams)


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, enc, enc_mask, ids, attn_mask, tt, pos, seed, embeds, *params):
        logits, saved = model._dec.forward(ids, enc, enc_mask, attn_mask, tt, pos, save=True, seed=seed, inputs_embeds=embeds)
        ctx.model, ctx.saved, ctx.nparams = model, saved, len(params)
        ctx.need_denc = enc is not None and enc.requires_grad
        ctx.embeds_like = embeds if (embeds is not None and embeds.requires_grad) else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if not model.direct_grads:
            model.zero_grads_prefix("decoder.")
        B, T, V = dlogits.shape
        d16 = dlogits.reshape(B * T, V)
        d16 = ops.cast_to_bf16(d16.contiguous()) if d16.dtype == torch.float32 else d16.contiguous()
        denc = model._dec.backward(ctx.saved, dlogits=d16, need_denc=ctx.need_denc)
        d_emb = None
        if ctx.embeds_like is not None:
            d_emb = ctx.saved["d_embeds"].view(ctx.embeds_like.shape).to(ctx.embeds_like.dtype)
        ctx.saved = None
        return (None, denc, None, None, None, None, None, None, d_emb) + model._collect_grads("decoder.", ctx.nparams)


# ---------------------------------------------------------------------------------------------------- sub-modules
class _EncoderModule(_Node):
    """`encoder_decoder.encoder(...)` of the reference (CvtWithProjectionHead / MultiCvtWithProjectionHead)."""

    def forward(self, pixel_values=None, output_hidden_states=None, return_dict=None, output_attentions=None):
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        out = self.__dict__["_owner"]()._encode(pixel_values)
        if return_dict is False:
            return out.last_hidden_state
        return out


class _DecoderModule(_Node):
    def print_trainable_parameters(self):
        t = sum(p.numel() for p in self.parameters() if p.requires_grad)
        a = sum(p.numel() for p in self.parameters())
        print(f"trainable params: {t} || all params: {a}")


# ---------------------------------------------------------------------------------------------------- models
class _CrossEntropyFn(torch.autograd.Function):
    """`CrossEntropyLoss()(logits.reshape(-1, V), labels.reshape(-1))` of the reference's forward(labels=...) on the fused loss kernel
    (csrc/loss.hip: log-softmax + nll + d(logits) in one pass over the vocabulary row; ignore_index = -100, mean over the counted labels)."""

    @staticmethod
    def forward(ctx, logits, labels):
        V = logits.shape[-1]
        flat = logits.detach().reshape(-1, V)
        flat = flat if flat.stride(1) == 1 else flat.contiguous()
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        w = ops.ce_weights(lab, -100, mode=0)
        loss, _, dl = ops.softmax_ce(flat, lab, -100, w, need_grad=True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl.float() * g).view(ctx.shape), None


def _is_bookkeeping_key(k: str) -> bool:
    """State-dict entries that are not weights: older transformers checkpoints persist `...embeddings.position_ids` (an arange buffer) and
    `num_batches_tracked` counters are kept by this build in one place of its own."""
    return k.endswith(".position_ids") or k.endswith("token_type_ids_buffer")


class _CXREncoderDecoderBase(ParamStore, GenerationMixin, TokenHelpers):
    kind = "single"
    main_input_name = "pixel_values"

    def __init__(self, config=None, encoder=None, decoder=None, device="cuda", seed: Optional[int] = 0, perturb: float = 0.0):
        """config: this build's EncoderDecoderConfig, or the HF `VisionEncoderDecoderConfig` the reference's Lightning modules construct
        (modules/lightning_modules/single.py:205-216) -- duck-typed, no transformers import. `encoder` / `decoder` modules (the reference's
        second constructor form, modelling_single.py:88-95) contribute their `.config`; their WEIGHTS arrive through load_state_dict()."""
        if config is None and (encoder is None or decoder is None):
            raise ValueError("Either a configuration or an encoder and a decoder has to be provided.")
        if config is None:
            config = {"encoder": encoder.config, "decoder": decoder.config}
        config = from_hf_config(config)
        assert config.decoder.add_cross_attention, '"add_cross_attention" must be True for the given decoder'
        assert config.decoder.is_decoder, '"is_decoder" must be True for the given decoder'
        shapes = weights.encoder_decoder_param_shapes(config)
        aliases = weights.tied_aliases(config.decoder)
        enc_mod, dec_mod = _EncoderModule(), _DecoderModule()
        ParamStore.__init__(self, shapes, aliases, device, trainable=self._initially_trainable,
                            root_modules={"encoder": enc_mod, "decoder": dec_mod})
        enc_mod.__dict__["_owner"] = weakref.ref(self)
        self.config = config
        self._enc = CvtEncoderEngine(self, config.encoder)
        self._dec = BertEngine(self, config.decoder)
        self.direct_grads = False
        if seed is not None:
            self.load_state_dict(weights.init_encoder_decoder(config, seed=seed, perturb=perturb))
        # Second constructor form of the reference (modelling_single.py:88-95; lightning_modules/single.py:218-221 builds the model from
        # `CvtWithProjectionHead.from_pretrained(...)` + a fresh decoder): the passed modules' WEIGHTS are the model's weights.
        for prefix, mod in (("encoder.", encoder), ("decoder.", decoder)):
            self._adopt_module_weights(prefix, mod)

    def _adopt_module_weights(self, prefix, mod):
        """Copy `mod.state_dict()` in under `prefix` (strict: every parameter / buffer of that half must be there, nothing else may be, apart
        from the non-parameter bookkeeping keys HF checkpoints carry). Modules without tensors (a bare object holding `.config`) are ignored."""
        sd_fn = getattr(mod, "state_dict", None)
        if mod is None or sd_fn is None:
            return
        sd = {k: v for k, v in sd_fn().items() if torch.is_tensor(v) and not _is_bookkeeping_key(k)}
        if not sd:
            return
        own = {k for k in self.state_dict().keys() if k.startswith(prefix)}
        given = {prefix + k for k in sd}
        for alias, canon in weights.tied_aliases(self.config.decoder).items():      # a tied tensor may be stored once
            if alias in own and alias not in given and canon in given:
                sd[alias[len(prefix):]] = sd[canon[len(prefix):]]
                given.add(alias)
        missing, unexpected = sorted(own - given), sorted(given - own)
        if missing or unexpected:
            raise RuntimeError(f"{type(mod).__name__} passed as `{prefix[:-1]}=` does not match this model's {prefix[:-1]}: "
                               f"missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}, unexpected keys {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        self.load_state_dict({prefix + k: v for k, v in sd.items()}, strict=False)

    def _initially_trainable(self, key):
        return True

    # -------------------------------------------------------------------------------------- checkpoints (HF directory layout)
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, device="cuda", **kwargs):
        """The reference's `Model.from_pretrained(ckpt_dir)` for a LOCAL checkpoint directory in the HF layout: `config.json` (a
        VisionEncoderDecoderConfig: `encoder` / `decoder` sub-dicts) + `model.safetensors` or `pytorch_model.bin` with the HF state-dict key names.
        There is no Hub access from this process: a model id that is not a directory raises. Constructed in eval mode, like transformers."""
        import json
        import os
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (this build never contacts the Hugging Face Hub; download the checkpoint first)")
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        model = cls(config=cfg, device=device, seed=None, **kwargs)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        for alias, canon in weights.tied_aliases(model.config.decoder).items():       # transformers writes tied tensors once (LM projection = word embeddings)
            if alias not in sd and canon in sd:
                sd[alias] = sd[canon]
        sd = {k: v for k, v in sd.items() if not _is_bookkeeping_key(k)}              # e.g. `decoder.bert.embeddings.position_ids` of older pytorch_model.bin files
        model.load_state_dict(sd)
        model.eval()
        return model

    def save_pretrained(self, save_directory, safe_serialization: bool = True):
        """`config.json` + weights in the HF directory layout (what from_pretrained reads; the state-dict keys are the reference's)."""
        import dataclasses
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        # sub-configurations carry their `model_type` so that transformers' AutoConfig can read the directory too
        cfg = {"model_type": "vision-encoder-decoder", "encoder": dict(dataclasses.asdict(self.config.encoder), model_type="cvt"),
               "decoder": dict(dataclasses.asdict(self.config.decoder), model_type="bert")}
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    # -------------------------------------------------------------------------------------- gradient plumbing
    def enable_direct_grads(self):
        """p.grad are views of the flat gradient buffer (no autograd copies); pair with FusedAdamW."""
        self.direct_grads = True
        self.attach_grads()

    def zero_grads_prefix(self, prefix):
        """Zero the gradient range of the PARAMETERS under `prefix` (buffers are stored behind all parameters and have no gradient: including
        their offsets would stretch the span over every later parameter)."""
        self.ensure_grads()
        offs = [(self._offsets[k], self._numel(k)) for k in self._offsets if k.startswith(prefix) and self._offsets[k] < self._param_total]
        lo, hi = min(o for o, _ in offs), max(o + n for o, n in offs)
        self.gflat[lo:hi].zero_()

    def _grad_params(self, prefix):
        return [(k, p) for k, p in self._params.items() if k.startswith(prefix) and p.requires_grad]

    def _collect_grads(self, prefix, n):
        if self.direct_grads:
            return (None,) * n
        return tuple(self.grad(k).clone() for k, _ in self._grad_params(prefix))

    # -------------------------------------------------------------------------------------- encoder
    def _pixels(self, pixel_values):
        px = pixel_values
        if px.device != self.device:
            px = px.to(self.device)
        return px.float().contiguous()

    def enable_fp8_encoder(self, calibration_pixel_values, margin: float = 2.0):
        """Run the frozen encoder's linear layers as e4m3 (OCP) GEMMs with per-tensor scales from here on (gradient-free forwards only; BASELINE.json
        configs[4]). `calibration_pixel_values` [B,(N,)3,H,W]: images the static activation scales are taken from. `disable_fp8_encoder()` undoes it.
        See CvtEncoderEngine.enable_fp8."""
        px = self._pixels(calibration_pixel_values)
        self._enc.enable_fp8(px.view(-1, *px.shape[-3:]), margin=margin)
        return self

    def disable_fp8_encoder(self):
        self._enc.fp8 = None
        return self

    def _encode(self, pixel_values):
        px = self._pixels(pixel_values)
        multi = px.dim() == 5
        flat = px.view(-1, *px.shape[-3:]) if multi else px
        params = [p for _, p in self._grad_params("encoder.")]
        if torch.is_grad_enabled() and params:
            feats = _EncodeFn.apply(self, flat, *params)
        else:
            feats, _ = self._enc.forward(flat, save=False)
        tokens = self.config.encoder.tokens_per_image
        if multi:
            B, N = px.shape[:2]
            return ModelOutput(last_hidden_state=feats.view(B, N * tokens, -1), attention_mask=ops.image_mask(px, tokens).bool())
        return ModelOutput(last_hidden_state=feats.view(px.shape[0], tokens, -1))

    # -------------------------------------------------------------------------------------- forward
    def forward(self, pixel_values=None, decoder_input_ids=None, decoder_attention_mask=None, encoder_outputs=None, past_key_values=None,
                decoder_inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, **kwargs):
        """reference modelling_longitudinal.py:173-249 (and the single/multi copies)."""
        kwargs_decoder = {k[len("decoder_"):]: v for k, v in kwargs.items() if k.startswith("decoder_")}
        if encoder_outputs is None:
            if pixel_values is None:
                raise ValueError("You have to specify pixel_values")
            encoder_outputs = self._encode(pixel_values)
        elif isinstance(encoder_outputs, tuple):
            encoder_outputs = ModelOutput(last_hidden_state=encoder_outputs[0])
        enc = encoder_outputs[0]
        enc_mask = encoder_outputs.get("attention_mask") if isinstance(encoder_outputs, dict) else None
        if self.kind == "single":
            enc_mask = None                                                  # modelling_single.py:176
        if past_key_values is not None:
            raise NotImplementedError("external past_key_values are not part of the accelerated path (generate() owns the KV cache)")
        if (decoder_input_ids is None) == (decoder_inputs_embeds is None):
            raise ValueError("You have to specify exactly one of decoder_input_ids or decoder_inputs_embeds")
        logits = self._decode_tf(decoder_input_ids, enc, enc_mask, decoder_attention_mask, kwargs_decoder.get("token_type_ids"),
                                 kwargs_decoder.get("position_ids"), embeds=decoder_inputs_embeds)
        loss = None
        if labels is not None:
            loss = _CrossEntropyFn.apply(logits, labels.to(logits.device))    # CrossEntropyLoss(): mean over labels != -100 (reference :239-241)
        out = ModelOutput(loss=loss, logits=logits, past_key_values=None, encoder_last_hidden_state=enc)
        if return_dict is False:
            return out.to_tuple()
        return out

    @staticmethod
    def _u8(mask, device):
        if mask is None:
            return None
        return mask.to(device=device, dtype=torch.uint8).contiguous()

    @staticmethod
    def _i64(t, device):
        if t is None:
            return None
        return t.to(device=device, dtype=torch.int64).contiguous()

    def _decode_tf(self, ids, enc, enc_mask, attn_mask, tt, pos, seed=None, embeds=None):
        dev = self.device
        ids, tt, pos = self._i64(ids, dev), self._i64(tt, dev), self._i64(pos, dev)
        if embeds is not None:
            embeds = embeds.to(dev)
        attn_mask, enc_mask = self._u8(attn_mask, dev), self._u8(enc_mask, dev)
        enc = enc if enc.dtype == torch.bfloat16 else ops.cast_to_bf16(enc.float().contiguous())
        enc = enc.contiguous()
        params = [p for _, p in self._grad_params("decoder.")]
        if torch.is_grad_enabled() and (params or enc.requires_grad or (embeds is not None and embeds.requires_grad)):
            return _DecodeFn.apply(self, enc, enc_mask, ids, attn_mask, tt, pos, seed, embeds, *params)
        logits, _ = self._dec.forward(ids, enc, enc_mask, attn_mask, tt, pos, save=False, seed=seed, inputs_embeds=embeds)
        return logits


class SingleCXREncoderDecoderModel(_CXREncoderDecoderBase):
    kind = "single"


class MultiCXREncoderDecoderModel(_CXREncoderDecoderBase):
    kind = "multi"


class LongitudinalPromptMultiCXREncoderDecoderModel(_CXREncoderDecoderBase):
    """Frozen encoder, LoRA (r=8, alpha=32) on decoder self-attention query/key (modelling_longitudinal.py:158-171)."""
    kind = "longitudinal"

    def __init__(self, config=None, encoder=None, decoder=None, device="cuda", seed=0, perturb=0.0):
        if config is not None:
            config = from_hf_config(config)
            if not config.decoder.lora_r:
                config.decoder.lora_r = 8                         # peft LoraConfig(r=8, lora_alpha=32, lora_dropout=0.1) of modelling_longitudinal.py:163-170
        super().__init__(config, encoder, decoder, device=device, seed=seed, perturb=perturb)
        if not self.config.decoder.lora_r:
            raise ValueError("the longitudinal model wraps the decoder's self-attention query / key in LoRA: pass a configuration")
        self.decoder.print_trainable_parameters()

    def _initially_trainable(self, key):
        return "lora_" in key
