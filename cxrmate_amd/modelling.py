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

import os
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


# ---------------------------------------------------------------------------------------------------- what the callers do with logits / scores
class BoundaryTensor(torch.Tensor):
    """`.logits` of forward() and the per-step `scores` of generate.__wrapped__() as the reference's callers receive them: ordinary fp32 tensors
    in every respect, except that the three torch calls those callers make on them are recognised and served without the [B, V, T] detour:

      * TF step (reference modules/lightning_modules/single.py:467-469): `F.cross_entropy(logits.permute([0, 2, 1]), labels, ignore_index=pad)`
        -- torch runs its "spatial" softmax over the strided class dimension (39 + 26 ms forward + backward for [32, 30000, 256] on MI355X);
        here the permuted view remembers its [B, T, V] base and the loss runs on the fused loss kernel (csrc/loss.hip, one pass per vocabulary row).
      * SCST step (scst/gt_prompt.py:189,230-235): `torch.stack(sample['scores'], dim=-1)` then `log_softmax(logits, dim=1)` then `nll_loss`:
        the stack of ALL steps of one generate call, in order, is the permuted view of the tensor they were unbound from (no copy), and the
        log-softmax over dim 1 of such a view is a row softmax of the contiguous base.
    Anything else falls through to torch with plain tensors (results are never BoundaryTensors), so no other code path changes behaviour."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        handler = _BOUNDARY_HANDLERS.get(func)
        if handler is not None:
            out = handler(*args, **kwargs)
            if out is not NotImplemented:
                return out
        if func not in _METADATA_ONLY:
            if _is_inplace(func) and args:
                m = _meta(args[0])
                if m is not None and m.get("kind") == "pending_logp":
                    m["dirty"] = True                       # the edit lands in the cached log-softmax: nll_loss must read that, not the scores
            args, kwargs = _materialise_pending(args), _materialise_pending(kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _as_boundary(tensor, **meta):
    b = tensor.as_subclass(BoundaryTensor)
    b._cxr = meta
    return b


def _meta(t):
    return getattr(t, "_cxr", None) if isinstance(t, BoundaryTensor) else None


def _plain(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, BoundaryTensor) else t


def _h_permute(x, *dims, **kw):
    m = _meta(x)
    if "dims" in kw:
        dims = (kw["dims"],)
    d = tuple(dims[0]) if len(dims) == 1 and isinstance(dims[0], (list, tuple)) else tuple(dims)
    if m is None or m.get("kind") != "btv" or d not in ((0, 2, 1), (0, -1, 1), (0, -1, -2), (0, 2, -2)):
        return NotImplemented
    return _as_boundary(_plain(x).permute(0, 2, 1), kind="bvt", base=_plain(x))


def _h_transpose(x, d0, d1):
    m = _meta(x)
    if m is None or m.get("kind") != "btv" or {d0 % 3, d1 % 3} != {1, 2}:
        return NotImplemented
    return _as_boundary(_plain(x).permute(0, 2, 1), kind="bvt", base=_plain(x))


def _h_cross_entropy(input, target, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean", label_smoothing=0.0):
    m = _meta(input)
    if (m is None or m.get("kind") != "bvt" or weight is not None or size_average is not None or reduce is not None or reduction != "mean"
            or label_smoothing != 0.0 or not torch.is_tensor(target) or target.dtype != torch.int64 or tuple(target.shape) != tuple(m["base"].shape[:2])):
        return NotImplemented
    return _CrossEntropyFn.apply(m["base"], _plain(target).to(m["base"].device), int(ignore_index))


def _h_stack(tensors, dim=0, out=None):
    if out is not None or dim not in (-1, 2) or not tensors:
        return NotImplemented
    metas = [_meta(t) for t in tensors]
    m0 = metas[0]
    if m0 is None or m0.get("kind") != "step" or len(tensors) != m0["base"].shape[1]:
        return NotImplemented
    if any(m is None or m.get("kind") != "step" or m["base"] is not m0["base"] or m["t"] != i for i, m in enumerate(metas)):
        return NotImplemented
    return _as_boundary(m0["base"].permute(0, 2, 1), kind="bvt", base=m0["base"])


def _h_log_softmax(input, dim=None, _stacklevel=3, dtype=None):
    """log_softmax over dim 1 of the [B, V, T] stack: PENDING. The SCST caller feeds it straight into nll_loss(reduction='none')
    (scst/gt_prompt.py:230-235), which the fused loss kernel serves from the scores themselves (one pass per vocabulary row, no [B, V, T]
    log-probability tensor, no dense one-hot gradient); the tensor handed back carries the scores' shape / dtype / device, and ANY other use of it
    computes the log-softmax first (BoundaryTensor.__torch_function__ -> _materialise_pending), so it behaves as the real result everywhere."""
    m = _meta(input)
    if m is None or m.get("kind") != "bvt" or dim not in (1, -2) or dtype is not None or m["base"].dtype != torch.float32:
        return NotImplemented
    return _as_boundary(m["base"].permute(0, 2, 1), kind="pending_logp", base=m["base"])


def _h_nll_loss(input, target, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean"):
    m = _meta(input)
    if m is not None and (m.get("dirty") or (m.get("value") is not None and m["value"]._version != m.get("value_version"))):
        # edited in place since -- directly (dirty) or through a view of the materialised value (`lp[:, 3].zero_()` goes through getitem, which
        # hands out a plain view; the edit moves the value's version counter): torch's nll_loss on the (cached, edited) log-probabilities
        return NotImplemented
    if (m is None or m.get("kind") != "pending_logp" or weight is not None or size_average is not None or reduce is not None or reduction != "none"
            or not torch.is_tensor(target) or target.dtype != torch.int64 or tuple(target.shape) != tuple(m["base"].shape[:2])):
        return NotImplemented
    return _RowNllFn.apply(m["base"], _plain(target).to(m["base"].device), int(ignore_index))


def _is_inplace(func):
    name = getattr(func, "__name__", "")
    return (name.endswith("_") and not name.endswith("__")) or name in ("__setitem__", "__iadd__", "__isub__", "__imul__", "__itruediv__", "__idiv__")


def _materialise_pending(obj):
    """Replace every pending log-softmax in a (nested) argument structure by the computed one. Computed ONCE per pending tensor and kept in its
    meta record: later uses (and in-place edits, which must be visible to later uses) see the same tensor, as with torch's own result."""
    if isinstance(obj, BoundaryTensor):
        m = getattr(obj, "_cxr", None)
        if m is not None and m.get("kind") == "pending_logp":
            v = m.get("value")
            if v is None:
                v = m["value"] = torch.log_softmax(m["base"], dim=-1).permute(0, 2, 1)
                m["value_version"] = v._version
            return v
        return obj
    if isinstance(obj, (list, tuple)):
        return type(obj)(_materialise_pending(o) for o in obj)
    if isinstance(obj, dict):
        return {k: _materialise_pending(v) for k, v in obj.items()}
    return obj


class _RowNllFn(torch.autograd.Function):
    """nll_loss(log_softmax(scores, vocabulary), target, ignore_index, reduction='none') -> [B, T] on the fused loss kernel (csrc/loss.hip): forward
    = one pass per row (log-sum-exp minus the target's score; -inf entries of the top-k filtered scores carry zero probability as in torch);
    backward = the same pass weighted by the incoming per-row gradient, emitting d(scores) = g * (softmax - onehot) directly."""

    @staticmethod
    def forward(ctx, scores, target, ignore_index):
        B, T, V = scores.shape
        flat = scores.detach().reshape(B * T, V)
        lab = target.reshape(-1).contiguous()
        ones = torch.ones((B * T,), dtype=torch.float32, device=scores.device)
        _, rows, _ = ops.softmax_ce(flat, lab, ignore_index, ones, need_grad=False)
        ctx.save_for_backward(flat, lab)
        ctx.ignore_index, ctx.shape = ignore_index, (B, T, V)
        return rows.view(B, T)

    @staticmethod
    def backward(ctx, g):
        flat, lab = ctx.saved_tensors
        w = g.reshape(-1).float().contiguous()
        _, _, dl = ops.softmax_ce(flat, lab, ctx.ignore_index, w, need_grad=True)       # rows with an ignored label get a zero gradient row
        return dl.view(ctx.shape), None, None


_BOUNDARY_HANDLERS = {
    torch.Tensor.permute: _h_permute, torch.permute: _h_permute, torch.Tensor.transpose: _h_transpose, torch.transpose: _h_transpose,
    torch.nn.functional.cross_entropy: _h_cross_entropy, torch.stack: _h_stack,
    torch.nn.functional.log_softmax: _h_log_softmax, torch.log_softmax: _h_log_softmax, torch.Tensor.log_softmax: _h_log_softmax,
    torch.nn.functional.nll_loss: _h_nll_loss,
}
# attribute reads that are the same for a pending log-softmax and for its scores: answered without computing anything
_METADATA_ONLY = {getattr(torch.Tensor, a).__get__ for a in ("shape", "dtype", "device", "ndim", "is_cuda", "requires_grad", "layout")} | {
    torch.Tensor.size, torch.Tensor.dim, torch.Tensor.numel, torch.Tensor.stride, torch.Tensor.is_contiguous, torch.Tensor.is_floating_point}


# ---------------------------------------------------------------------------------------------------- autograd bridges
class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, px, *params):
        model._settle_owed_join()
        feats, saved = model._enc.forward(px, save=True)
        ctx.model, ctx.saved, ctx.nparams = model, saved, len(params)
        # the stage-1 weight gradient is contracted against an im2col matrix that the BACKWARD builds from the pixels (the forward is an implicit
        # GEMM): the caller's tensor is held by reference, so an in-place edit between forward and backward must be caught, as autograd's own
        # saved-tensor version check would (px is not a differentiable input: save_for_backward has nothing to check it against)
        ctx.px, ctx.px_version = px, px._version
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        from .training import wgrad_overlap
        model = ctx.model
        if ctx.px._version != ctx.px_version:
            raise RuntimeError("pixel_values was modified in place between the encoder forward and its backward (version "
                               f"{ctx.px_version} -> {ctx.px._version}): the patch-embedding weight gradient is computed from it in backward")
        bound = model._grads_bindable("encoder.")
        if not model.direct_grads and bound != "accumulate":
            model.zero_grads_prefix("encoder.")
        with wgrad_overlap():                                              # weight-gradient GEMMs beside the dX chain, as in the fused step
            model._enc.backward(ctx.saved, dfeats.contiguous())
            if not bound:
                ops.wgrad_join()                                           # ... complete before autograd reads them
            else:
                model._queue_backward_join()
        ctx.saved = ctx.px = None
        if bound:
            return (None, None) + model._bind_grads("encoder.", ctx.nparams)
        return (None, None) + model._collect_grads("encoder.", ctx.nparams)


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, enc, enc_mask, ids, attn_mask, tt, pos, seed, embeds, cross_kv, logit_from, *params):
        model._settle_owed_join()
        logits, saved = model._dec.forward(ids, enc, enc_mask, attn_mask, tt, pos, save=True, seed=seed, inputs_embeds=embeds, cross_kv=cross_kv,
                                           logit_from=logit_from)
        ctx.model, ctx.saved, ctx.nparams = model, saved, len(params)
        ctx.need_denc = enc is not None and enc.requires_grad
        ctx.embeds_like = embeds if (embeds is not None and embeds.requires_grad) else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        bound = model._grads_bindable("decoder.")
        if not model.direct_grads and bound != "accumulate":
            model.zero_grads_prefix("decoder.")
        B, T, V = dlogits.shape
        d16 = dlogits.reshape(B * T, V)
        d16 = ops.cast_to_bf16(d16.contiguous()) if d16.dtype == torch.float32 else d16.contiguous()
        from .training import wgrad_overlap
        with wgrad_overlap():
            denc = model._dec.backward(ctx.saved, dlogits=d16, need_denc=ctx.need_denc)
            if not bound:
                ops.wgrad_join()
            else:
                model._queue_backward_join()                               # the decoder's weight gradients keep running under the encoder backward
        d_emb = None
        if ctx.embeds_like is not None:
            d_emb = ctx.saved["d_embeds"].view(ctx.embeds_like.shape).to(ctx.embeds_like.dtype)
        ctx.saved = None
        if bound:
            return (None, denc, None, None, None, None, None, None, d_emb, None, None) + model._bind_grads("decoder.", ctx.nparams)
        return (None, denc, None, None, None, None, None, None, d_emb, None, None) + model._collect_grads("decoder.", ctx.nparams)


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
    def forward(ctx, logits, labels, ignore_index=-100):
        V = logits.shape[-1]
        flat = logits.detach().reshape(-1, V)
        flat = flat if flat.stride(1) == 1 else flat.contiguous()
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        w = ops.ce_weights(lab, ignore_index, mode=0)
        loss, _, dl = ops.softmax_ce(flat, lab, ignore_index, w, need_grad=True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl.float() * g).view(ctx.shape), None, None


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

    def _grads_bindable(self, prefix):
        """The autograd bridges may hand the gradients of the parameters under `prefix` over by BINDING `p.grad` to views of the flat gradient buffer
        (no clone of the range, no AccumulateGrad copy per parameter, and the weight-gradient stream is joined once, when the whole backward pass is
        over, instead of at the end of every bridge). Two states allow it: every trainable parameter's .grad is None -- what `optimizer.zero_grad()`
        (set_to_none=True: torch's and Lightning's default) leaves behind: the range is zeroed and filled -- or every .grad already IS such a view
        (zero_grad(set_to_none=False), or gradient accumulation over several backward calls): the engine adds into the buffer, which is what autograd
        would do. Returns 'fresh' / 'accumulate' / None. In any other state the gradients go through autograd as before, after .grad tensors that
        alias the flat buffer have been detached from it (the bridge is about to overwrite that buffer)."""
        if self.direct_grads or os.environ.get("CXR_BIND_GRADS", "1") == "0":
            return None
        ps = self._grad_params(prefix)
        if not ps:
            return None
        # Binding skips autograd's AccumulateGrad nodes, so it is only allowed where that cannot be observed: no tensor hook / post-accumulate hook
        # on any parameter, and no torch.distributed process group (DistributedDataParallel's reducer hangs its hooks on the AccumulateGrad nodes
        # themselves, where they cannot be seen from here: under a process group the gradients always take the autograd route; this build's own
        # data-parallel path -- training.tf_train_step / dp.GradReducer -- uses direct_grads and never reaches this point).
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return self._unbind_grads(ps)
        if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for _, p in ps):
            return self._unbind_grads(ps)
        if all(p.grad is None for _, p in ps):
            return "fresh"
        lo, hi = self.gflat.data_ptr(), self.gflat.data_ptr() + self.gflat.numel() * 4
        own = [p.grad is not None and lo <= p.grad.data_ptr() < hi for _, p in ps]
        if all(own):
            return "accumulate"
        return self._unbind_grads(ps)

    def _unbind_grads(self, ps):
        """The gradients of `ps` go through autograd: .grad tensors that alias the flat gradient buffer are detached from it first (the bridge is
        about to overwrite that buffer, and autograd would otherwise accumulate into the very memory it is handed). -> None"""
        if self.gflat is None:
            return None
        lo, hi = self.gflat.data_ptr(), self.gflat.data_ptr() + self.gflat.numel() * 4
        for _, p in ps:
            if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                p.grad = p.grad.clone()
        return None

    def _bind_grads(self, prefix, n):
        for k, p in self._grad_params(prefix):
            o = self._offsets[k]
            p.grad = self.gflat[o: o + self._numel(k)].view(p.shape)
        return (None,) * n

    def _queue_backward_join(self):
        """One join of the weight-gradient stream at the end of the running backward pass (autograd final callback)."""
        if self.__dict__.get("_join_queued"):
            return
        from .training import wgrad_overlap
        side, cur = wgrad_overlap._stream, torch.cuda.current_stream(self.device)
        self.__dict__["_join_queued"] = True

        def done():
            self.__dict__["_join_queued"] = False
            with torch.cuda.stream(cur):
                ops.wgrad_join(side)

        torch.autograd.Variable._execution_engine.queue_callback(done)

    def _settle_owed_join(self):
        """A backward pass that RAISED never runs its final callbacks (the autograd engine skips them on an exception), so the join queued by
        _queue_backward_join() is still owed and the flag would keep every later backward pass from queueing its own: a new forward pass joins the
        weight-gradient stream itself and clears the flag (nothing to do in the normal case)."""
        if self.__dict__.get("_join_queued"):
            from .training import wgrad_overlap
            self.__dict__["_join_queued"] = False
            ops.wgrad_join(wgrad_overlap._stream)

    def _collect_grads(self, prefix, n):
        """Gradients handed to autograd for the trainable parameters under `prefix`: views of ONE copy of the flat gradient range they span
        (a clone per parameter was ~400 copy launches per backward)."""
        if self.direct_grads:
            return (None,) * n
        keys = [k for k, _ in self._grad_params(prefix)]
        if not keys:
            return ()
        lo = min(self._offsets[k] for k in keys)
        hi = max(self._offsets[k] + self._numel(k) for k in keys)
        snap = self.gflat[lo:hi].clone()
        return tuple(snap[self._offsets[k] - lo: self._offsets[k] - lo + self._numel(k)].view(self._params[k].shape) for k in keys)

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
        self._spec_result = None                       # a new encoder pass: no greedy baseline decoded along an earlier batch can be asked for any more
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
        if (decoder_input_ids is None) == (decoder_inputs_embeds is None):
            raise ValueError("You have to specify exactly one of decoder_input_ids or decoder_inputs_embeds")
        if past_key_values is not None:
            return self._forward_cached(decoder_input_ids, enc, enc_mask, decoder_attention_mask, kwargs_decoder, past_key_values, labels,
                                        decoder_inputs_embeds, return_dict)
        logits = self._decode_tf(decoder_input_ids, enc, enc_mask, decoder_attention_mask, kwargs_decoder.get("token_type_ids"),
                                 kwargs_decoder.get("position_ids"), embeds=decoder_inputs_embeds)
        loss = None
        if labels is not None:
            loss = _CrossEntropyFn.apply(logits, labels.to(logits.device), -100)    # CrossEntropyLoss(): mean over labels != -100 (reference :239-241)
        if logits.dim() == 3:
            logits = _as_boundary(logits, kind="btv")                       # see BoundaryTensor: the callers' permute + F.cross_entropy stays on the fused loss kernel
        past = None
        if use_cache and decoder_input_ids is not None and not torch.is_grad_enabled():
            # use_cache=True WITHOUT a cache, the library's contract (TF5 models/bert/modeling_bert.py:851-905): logits for EVERY fed position (the
            # teacher-forced pass above) plus a cache filled with those positions, which the next call hands back with its new tokens only. The
            # cache is built by the cached-step kernels' prefill over the same inputs (its last-position logits are not needed). Under autograd no
            # cache is returned (the cached kernels have no backward): the call is then an ordinary teacher-forced pass, as in training.
            past = self._forward_cached(decoder_input_ids, enc, enc_mask, decoder_attention_mask, kwargs_decoder, None, None, None, True).past_key_values
        out = ModelOutput(loss=loss, logits=logits, past_key_values=past, encoder_last_hidden_state=enc)
        if return_dict is False:
            return out.to_tuple()
        return out

    def _forward_cached(self, ids, enc, enc_mask, attn_mask, kwargs_decoder, past, labels, embeds, return_dict):
        """forward(..., past_key_values=cache) for callers that run their own decoding loop (transformers' generate does exactly this through
        prepare_inputs_for_generation, modelling_longitudinal.py:251-295): `past_key_values` is the engine's own KV cache object (opaque: hand back what the
        previous call returned), decoder_input_ids holds the NEW token of every row, decoder_attention_mask the full mask [B, past + new]. Gradient-free
        (the cached kernels have no backward). Logits come back for the fed position, [B, 1, V], as from the library; feeding several new tokens on top
        of a non-empty cache (chunked prefill) is not something the reference's callers do and is refused rather than answered with one position."""
        if labels is not None or embeds is not None:
            raise NotImplementedError("past_key_values: decoder_input_ids only, no labels (the cached path is the decoding path)")
        if torch.is_grad_enabled() and any(p.requires_grad for _, p in self._grad_params("decoder.")):
            raise RuntimeError("past_key_values needs torch.no_grad(): the cached decode kernels have no backward")
        if past is not None and past.len > 0 and ids.shape[1] != 1:
            raise NotImplementedError("past_key_values with more than one new token per row: only the last position's logits would come back")
        dev = self.device
        ids = self._i64(ids, dev)
        B, Tn = ids.shape
        enc16 = (enc if enc.dtype == torch.bfloat16 else ops.cast_to_bf16(enc.float().contiguous())).contiguous()
        cache = past
        if cache is None:
            cache = self._dec.new_cache(B, int(self.config.decoder.max_position_embeddings), dev)
        if cache.len + Tn > cache.Tmax:
            raise ValueError(f"cache holds {cache.len} positions, {Tn} more exceed max_position_embeddings = {cache.Tmax}")
        seed = self.next_dropout_seed() if self.training else None
        logits = self._dec.decode(cache, ids, enc16, self._u8(enc_mask, dev), self._u8(attn_mask, dev), self._i64(kwargs_decoder.get("token_type_ids"), dev),
                                  self._i64(kwargs_decoder.get("position_ids"), dev), train=self.training, seed=seed)
        out = ModelOutput(loss=None, logits=logits.view(B, 1, -1), past_key_values=cache, encoder_last_hidden_state=enc)
        return out.to_tuple() if return_dict is False else out

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

    def _decode_tf(self, ids, enc, enc_mask, attn_mask, tt, pos, seed=None, embeds=None, cross_kv=None, logit_from=0):
        dev = self.device
        ids, tt, pos = self._i64(ids, dev), self._i64(tt, dev), self._i64(pos, dev)
        if embeds is not None:
            embeds = embeds.to(dev)
        attn_mask, enc_mask = self._u8(attn_mask, dev), self._u8(enc_mask, dev)
        enc = enc if enc.dtype == torch.bfloat16 else ops.cast_to_bf16(enc.float().contiguous())
        enc = enc.contiguous()
        params = [p for _, p in self._grad_params("decoder.")]
        if torch.is_grad_enabled() and (params or enc.requires_grad or (embeds is not None and embeds.requires_grad)):
            return _DecodeFn.apply(self, enc, enc_mask, ids, attn_mask, tt, pos, seed, embeds, cross_kv, int(logit_from), *params)
        logits, _ = self._dec.forward(ids, enc, enc_mask, attn_mask, tt, pos, save=False, seed=seed, inputs_embeds=embeds, cross_kv=cross_kv,
                                      logit_from=logit_from)
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
