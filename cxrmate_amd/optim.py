"""`torch.optim.AdamW` for the drop-in model classes, on the fused kernel.

The reference's Lightning modules build their optimiser in `configure_optimizers` as `torch.optim.AdamW(self.parameters(), lr=self.lr)`
(modules/lightning_modules/single.py:426-431; the SCST stage over the decoder's parameters only, longitudinal/scst/gt_prompt.py:34-40).
That works unchanged on this build's models -- the nn.Parameters are views of one flat fp32 buffer -- but torch then updates ~700 small tensors
with its foreach kernels (~3.5 ms per step for 112 M parameters on MI355X) and the bf16 shadow the MFMA kernels read has to be re-cast afterwards.

    from cxrmate_amd.optim import AdamW
    optimiser = {'optimizer': AdamW(self.parameters(), lr=self.lr)}

is the same optimiser (constructor, param groups, `zero_grad`, `step(closure)`, `state_dict` / `load_state_dict`, lr schedulers writing
`group['lr']`) whose `step()` is ONE kernel launch per contiguous run of parameters: fp32 master weights, both moments and the bf16 shadow in a
single pass (csrc/misc.hip adamw_kernel), reading the gradients where the autograd bridges left them (views of the flat gradient buffer).
Parameters that do not belong to a cxrmate_amd model are updated by a plain per-tensor implementation of the same formula.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import ops
from .store import ALIGN


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False, maximize: bool = False, foreach=None, capturable: bool = False, differentiable: bool = False, fused=None):
        # torch.optim.AdamW's remaining keywords: `foreach` / `fused` only choose among torch's own implementations (any value is accepted and
        # ignored -- this IS a fused implementation); the ones that change semantics are refused, as is anything unknown (TypeError, as in torch)
        if amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("amsgrad / maximize / capturable / differentiable are not used by the reference (torch.optim.AdamW defaults)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._moments: Dict[int, Tuple[object, torch.Tensor, torch.Tensor]] = {}       # id(store) -> (store, m flat, v flat)
        self._plans: Dict[int, dict] = {}                                               # param-group index -> the launches of the previous step (fast path)

    # ------------------------------------------------------------------------------------------ state
    @staticmethod
    def _owner(p):
        ref = getattr(p, "_cxr_store", None)
        store = ref() if ref is not None else None
        if store is None or not p.is_cuda:
            return None, None
        key = p._cxr_key
        if store._params.get(key) is not p or p.data_ptr() != store.flat32.data_ptr() + 4 * store._offsets[key]:
            return None, None                               # the store was re-packed under the parameter (.to()): plain path until re-created
        return store, key

    def _flat_moments(self, store):
        hit = self._moments.get(id(store))
        if hit is None or hit[1].device != store.flat32.device or hit[1].numel() != store.flat32.numel():
            hit = (store, torch.zeros_like(store.flat32), torch.zeros_like(store.flat32))
            self._moments[id(store)] = hit
        return hit[1], hit[2]

    def _state_of(self, p, store=None, key=None):
        """Per-parameter state in torch.optim.AdamW's layout (step, exp_avg, exp_avg_sq). For a parameter of a cxrmate_amd model the two moments
        are VIEWS of flat buffers laid out like the parameter store (state_dict() sees the live values); moments that arrived through
        load_state_dict(), or that belong to a store that was re-packed since (.to()), are copied into the flat buffers and re-bound here."""
        st = self.state[p]
        if not st:
            st["step"] = 0
            if store is None:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
        if store is not None:
            m, v = self._flat_moments(store)
            o, n = store._offsets[key], p.numel()
            mv, vv = m[o:o + n].view(p.shape), v[o:o + n].view(p.shape)
            if "exp_avg" in st and st["exp_avg"].data_ptr() != mv.data_ptr():
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = mv, vv
        return st

    def load_state_dict(self, state_dict):
        self._plans.clear()
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())

    # ------------------------------------------------------------------------------------------ step
    def _flush_plan(self, gi):
        """Write a group's shared step count back into the per-parameter states. While the fast path below is active the per-parameter
        `state[p]['step']` is NOT advanced every step (one counter per group, `plan['t']`): it is brought up to date here -- whenever the fast path is
        left -- and by state_dict(); read step counts through state_dict(), as checkpointing code does."""
        plan = self._plans.pop(gi, None)
        if plan is not None:
            for p in plan["params"]:
                self.state[p]["step"] = plan["t"]

    def state_dict(self):
        for gi in list(self._plans):
            plan = self._plans[gi]
            for p in plan["params"]:
                self.state[p]["step"] = plan["t"]
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        touched = {}
        for gi, group in enumerate(self.param_groups):
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            plan = self._plans.get(gi)
            if plan is not None:
                # steady state: the same parameters, every gradient where the autograd bridges bind it (a view of the flat gradient buffer), the same
                # flat buffers -> the launches of the previous step again, nothing per parameter but the pointer check
                gp = group["params"]
                ok = (len(gp) == len(plan["params"]) and all(a is b for a, b in zip(gp, plan["params"]))      # the SAME parameters, by identity
                      and all(st_.flat32.data_ptr() == ptr for st_, ptr in plan["stores"]))
                if ok:
                    for p, ptr in plan["checks"]:
                        g = p.grad
                        if g is None or g.data_ptr() != ptr:
                            ok = False
                            break
                if ok:
                    plan["t"] += 1
                    for store, lo, hi, m, v in plan["runs"]:
                        ops.adamw_step(store.flat32[lo:hi], store.gflat[lo:hi], m[lo:hi], v[lo:hi], store.flat16[lo:hi], lr, b1, b2, eps, wd, plan["t"], 1.0)
                        touched[id(store)] = store
                    continue
                self._flush_plan(gi)
            runs: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}          # (store id, step count) -> [(lo, hi)] of parameters with gradients
            checks, plain = [], False
            for p in group["params"]:
                if p.grad is None:
                    plain = True                                             # (a parameter without a gradient this step: no fast path next time)
                    continue
                store, key = self._owner(p)
                st = self._state_of(p, store, key)
                st["step"] += 1
                if store is None:
                    self._plain_update(p, st, lr, b1, b2, eps, wd)
                    plain = True
                    continue
                store.ensure_grads()
                o, n = store._offsets[key], p.numel()
                gview = store.gflat[o:o + n]
                if p.grad.data_ptr() != gview.data_ptr():                    # gradient accumulated by autograd (hooks / DDP / foreign .grad): one copy in
                    gview.view(p.shape).copy_(p.grad)
                    plain = True
                checks.append((p, gview.data_ptr()))
                runs.setdefault((id(store), st["step"]), []).append((o, o + ((n + ALIGN - 1) // ALIGN) * ALIGN))
                touched[id(store)] = store
            launches = []
            for (sid, t), spans in runs.items():
                store = touched[sid]
                m, v = self._flat_moments(store)
                for lo, hi in _merge(spans):
                    # alignment gaps between parameters hold zeros in every buffer and stay zero under the update
                    ops.adamw_step(store.flat32[lo:hi], store.gflat[lo:hi], m[lo:hi], v[lo:hi], store.flat16[lo:hi], lr, b1, b2, eps, wd, t, 1.0)
                    launches.append((store, lo, hi, m, v))
            steps = {t for _, t in runs}
            if not plain and len(steps) == 1:
                self._plans[gi] = dict(params=[p for p, _ in checks], checks=checks, runs=launches, t=steps.pop(),
                                       stores=[(touched[sid], touched[sid].flat32.data_ptr()) for sid in {sid for sid, _ in runs}])
        for store in touched.values():
            store.shadow_version += 1        # master and bf16 shadow were written together: the engines re-derive their per-version weight re-layouts
        return loss

    @staticmethod
    def _plain_update(p, st, lr, b1, b2, eps, wd):
        g = p.grad
        p.mul_(1.0 - lr * wd)
        st["exp_avg"].mul_(b1).add_(g, alpha=1.0 - b1)
        st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        t = st["step"]
        denom = (st["exp_avg_sq"] / (1.0 - b2 ** t)).sqrt_().add_(eps)
        p.addcdiv_(st["exp_avg"], denom, value=-lr / (1.0 - b1 ** t))


def _merge(spans):
    out: List[List[int]] = []
    for lo, hi in sorted(spans):
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return [(lo, hi) for lo, hi in out]
