"""Flat parameter storage for the MI355X engines.

All floating-point parameters/buffers of a model live in ONE fp32 master buffer (288 GB of HBM: no reason to scatter 700
small allocations), with a bf16 shadow of identical layout that the MFMA kernels read, and an fp32 gradient buffer of the
same layout that the backward kernels accumulate into. The nn.Parameters a caller sees (HF key names, reference
`*_to_hub.ipynb` cell 6) are views into the master buffer, so `load_state_dict`, `state_dict`, `parameters()` and any
torch optimiser keep working, while the fused AdamW kernel (ops.adamw_step) updates master + shadow in one pass.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Dict, Tuple

import torch
from torch import nn

from . import ops
from .weights import is_buffer

ALIGN = 64   # elements; keeps every view 128/256-byte aligned


class _Node(nn.Module):
    """Anonymous container used to reproduce HF's dotted module paths."""


class ParamStore(nn.Module):
    def __init__(self, shapes: "OrderedDict[str, Tuple[int, ...]]", aliases: Dict[str, str], device, trainable=lambda k: True,
                 root_modules: Dict[str, nn.Module] | None = None):
        super().__init__()
        for name, mod in (root_modules or {}).items():      # callable nodes (`.encoder(...)`, `.decoder`) the HF keys hang under
            self.add_module(name, mod)
        self._shapes = OrderedDict((k, tuple(v)) for k, v in shapes.items() if k not in aliases)
        self._aliases = dict(aliases)
        self._trainable = trainable
        self._offsets: Dict[str, int] = {}
        off = 0
        # parameters first (trainable ranges stay contiguous for the fused optimiser / gradient all-reduce), buffers after them
        for want_buffer in (False, True):
            for k, shp in self._shapes.items():
                if k.endswith("num_batches_tracked") or is_buffer(k) != want_buffer:
                    continue
                n = 1
                for d in shp:
                    n *= d
                self._offsets[k] = off
                off += ((n + ALIGN - 1) // ALIGN) * ALIGN
            if not want_buffer:
                self._param_total = off
        self._total = off
        self._views32: Dict[str, torch.Tensor] = {}
        self._views16: Dict[str, torch.Tensor] = {}
        self._viewsg: Dict[str, torch.Tensor] = {}
        self._params: Dict[str, torch.Tensor] = {}
        self.flat32 = self.flat16 = self.gflat = None
        self.shadow_dirty = True
        self.shadow_version = 0
        self._master_version = -1
        self._build(torch.device(device))
        self.static_dropout_seed = None          # set (device int32 [1]) by graph-captured training steps: advanced on the device per replay
        self.train(False)                        # like from_pretrained(): eval until the trainer calls .train()

    def next_dropout_seed(self):
        """Device int32 [1] seed of one train-mode pass. Drawn from torch's CPU generator (reproducible under torch.manual_seed); when a
        static seed is installed (hipGraph capture) it is advanced in place on the device instead, so every replay sees fresh masks."""
        if self.static_dropout_seed is not None:
            from . import ops
            ops.increment_(self.static_dropout_seed)
            return self.static_dropout_seed
        val = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return torch.full((1,), val, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------------------------------ construction
    def _numel(self, k):
        n = 1
        for d in self._shapes[k]:
            n *= d
        return n

    def _build(self, device, source: Dict[str, torch.Tensor] | None = None):
        flat32 = torch.zeros(self._total, dtype=torch.float32, device=device)
        self.flat32 = flat32
        self.flat16 = torch.zeros(self._total, dtype=torch.bfloat16, device=device)
        self.gflat = None
        self._views32.clear(); self._views16.clear(); self._viewsg.clear()
        nbt_keys = [k for k in self._shapes if k.endswith("num_batches_tracked")]
        self.num_batches_tracked = torch.zeros(len(nbt_keys), dtype=torch.int64, device=device)     # all BatchNorm counters: one add per step
        for i, k in enumerate(nbt_keys):
            if source is not None:
                self.num_batches_tracked[i] = source[k].to(device)
            self._register(k, self.num_batches_tracked[i], buffer=True)
        for k, shp in self._shapes.items():
            if k.endswith("num_batches_tracked"):
                continue
            o, n = self._offsets[k], self._numel(k)
            v = flat32[o:o + n].view(shp)
            if source is not None:
                v.copy_(source[k])
            self._views32[k] = v
            self._views16[k] = self.flat16[o:o + n].view(shp)
            if is_buffer(k):
                self._register(k, v, buffer=True)
            else:
                p = self._params.get(k)
                if p is None:
                    p = nn.Parameter(v, requires_grad=bool(self._trainable(k)))
                    self._params[k] = p
                    p._cxr_store, p._cxr_key = weakref.ref(self), k       # lets optim.AdamW find the flat buffers behind a bare parameter list
                else:
                    p.data = v
                self._register(k, p, buffer=False)
        for alias, canon in self._aliases.items():
            self._register(alias, self._params[canon], buffer=False)
        self.shadow_dirty = True

    def _register(self, key, tensor, buffer):
        mod = self
        parts = key.split(".")
        for name in parts[:-1]:
            if name not in mod._modules:
                mod.add_module(name, _Node())
            mod = mod._modules[name]
        leaf = parts[-1]
        if buffer:
            if leaf in mod._buffers:
                mod._buffers[leaf] = tensor
            else:
                mod.register_buffer(leaf, tensor)
        else:
            if leaf in mod._parameters:
                mod._parameters[leaf] = tensor
            else:
                mod.register_parameter(leaf, tensor)

    def _apply(self, fn, recurse=True):
        """`.to()/.cuda()/.float()`: move, then re-pack into one flat buffer on the new device."""
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self.flat32.device))
        if probe.device == self.flat32.device and probe.dtype == torch.float32:
            return self
        src = {k: v.detach().clone() for k, v in self._views32.items()}
        for k in self._shapes:
            if k.endswith("num_batches_tracked"):
                src[k] = self.get_buffer(k)
        self._build(probe.device, src)
        return self

    # ------------------------------------------------------------------------------------------ access
    @property
    def device(self):
        return self.flat32.device

    def keys(self):
        return self._views32.keys()

    def has(self, k):
        return k in self._views32

    def f32(self, k) -> torch.Tensor:
        return self._views32[k]

    def w16(self, k) -> torch.Tensor:
        return self._views16[k]

    def grad(self, k) -> torch.Tensor:
        if self.gflat is None:
            self.ensure_grads()
        return self._viewsg[k]

    def span(self, keys, kind="w16"):
        """One flat view over several parameters that lie back to back in storage order (kind: 'f32' master, 'w16' bf16 shadow, 'grad'), or
        None when they do not."""
        off = self._offsets[keys[0]]
        end = off
        for k in keys:
            if self._offsets.get(k) != end:
                return None
            end += self._numel(k)
        if kind == "grad" and self.gflat is None:
            self.ensure_grads()
        flat = {"f32": self.flat32, "w16": self.flat16, "grad": self.gflat}[kind]
        return flat[off:end]

    def param(self, k) -> nn.Parameter:
        return self._params[self._aliases.get(k, k)]

    def ensure_grads(self):
        if self.gflat is None:
            self.gflat = torch.zeros(self._total, dtype=torch.float32, device=self.device)
            for k, shp in self._shapes.items():
                if k in self._offsets:
                    o, n = self._offsets[k], self._numel(k)
                    self._viewsg[k] = self.gflat[o:o + n].view(shp)

    def attach_grads(self):
        """Direct mode: `p.grad` IS the flat gradient view (no autograd copies)."""
        self.ensure_grads()
        for k, p in self._params.items():
            if p.requires_grad:
                p.grad = self._viewsg[k]

    def trainable_ranges(self):
        """Merged [lo, hi) element ranges of the flat buffers covered by parameters with requires_grad (alignment gaps included)."""
        spans = sorted((self._offsets[k], self._offsets[k] + ((self._numel(k) + ALIGN - 1) // ALIGN) * ALIGN)
                       for k, p in self._params.items() if p.requires_grad)
        out = []
        for lo, hi in spans:
            if out and lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return [(lo, hi) for lo, hi in out]

    def zero_grads(self):
        if self.gflat is not None:
            self.gflat.zero_()

    def refresh_shadow(self, force=False):
        """fp32 master -> bf16 shadow (one pass over the flat buffer).

        Besides the explicit flag, an in-place torch edit of ANY parameter marks the shadow stale: the nn.Parameters are views of the master
        buffer and share its autograd version counter, so `torch.optim.AdamW.step()` (the reference's optimiser, single.py:426-431),
        `p.data.copy_()`, `p.add_()` ... all move `flat32._version`. The fused AdamW kernel writes master and shadow together through raw
        pointers and does not move it."""
        ver = self.flat32._version
        if ver != self._master_version:
            self._master_version = ver
            self.shadow_dirty = True
        if self.shadow_dirty or force:
            ops.cast_to_bf16(self.flat32, self.flat16)
            self.shadow_dirty = False
            self.shadow_version += 1

    def mark_dirty(self):
        self.shadow_dirty = True

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self.shadow_dirty = True
        enc = getattr(self, "_enc", None)
        if enc is not None and getattr(enc, "fp8", None) is not None:
            enc.fp8 = None              # e4m3 weight snapshot + activation scales belong to the old weights: back to bf16 until enable_fp8_encoder() runs again
        return out
