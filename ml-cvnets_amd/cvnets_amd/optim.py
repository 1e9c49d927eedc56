"""Fused optimizer / EMA step (SURVEY §8f "next" row 1): ``AdamW`` with torch.optim.AdamW's interface (param_groups, state_dict-free
stepping, per-group ``lr`` / ``weight_decay`` that a scheduler may rewrite every iteration — optim/adamw.py:16-46,
optim/scheduler/*) whose ``step()`` is ONE kernel launch over every parameter tensor (cvh_adamw_multi), optionally folding
``EMA.update_parameters`` (cvnets/misc/averaging_utils.py:43-55) into the same pass."""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import _lib, ops
from .ops import _p, _stream


class AdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False, ema: Optional[tuple] = None, ema_momentum: float = 0.0005):
        """ema = (model, ema_model): ema_model (a deepcopy of model, averaging_utils.py:33) is updated in the same pass"""
        if amsgrad:
            raise NotImplementedError("amsgrad is not on the HIP hot path")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        b = {tuple(g["betas"]) for g in self.param_groups} | {(g["eps"],) for g in self.param_groups}
        if len(b) != 2:
            raise NotImplementedError("per-group betas / eps are not supported (the reference uses one setting, optim/adamw.py:30-40)")
        self._plan = None
        self._loaded_state = {}
        self._rebuilds = 0
        self._ema = ema
        self.ema_momentum = float(ema_momentum)
        self._flat = None  # (flat fp32 buffer, [(parameter, view)]) when this optimizer owns the gradient storage (`flat_grads`)

    # ---- engine-driven use (cvnets_amd/launch.py): the reference's Trainer owns the loop, this object the gradient storage -------------
    @classmethod
    def from_torch(cls, optimizer: torch.optim.Optimizer, flat_grads: bool = True) -> "AdamW":
        """A fused AdamW with the parameter groups (parameters, lr, weight_decay, betas, eps — every other key rides along untouched, so a
        scheduler that rewrites ``param_groups[i]["lr"]`` keeps working) of a torch.optim.AdamW built by the reference's
        ``build_optimizer`` (optim/adamw.py:16-46).  State is not carried over: call it before the first step (or load a checkpoint after)."""
        if any(g.get("amsgrad", False) for g in optimizer.param_groups):
            raise NotImplementedError("amsgrad is not on the HIP hot path")
        groups = [{k: v for k, v in g.items()} for g in optimizer.param_groups]
        g0 = groups[0]
        new = cls(groups, lr=g0["lr"], betas=tuple(g0["betas"]), eps=g0["eps"], weight_decay=g0["weight_decay"])
        if flat_grads:
            new.adopt_flat_grads()
        return new

    def adopt_flat_grads(self) -> None:
        """Point every parameter's ``.grad`` into ONE flat fp32 buffer that this optimizer owns (parameters whose gradient already is a view
        of somebody's flat storage — cvnets_amd.ddp's buckets — are left alone).  ``zero_grad`` then is one memset whatever ``set_to_none``
        says, the gradient addresses never move (the one-launch plan stays valid), and with ``ops.set_inplace_param_grads(True)`` the
        backward kernels add straight into the buffer: no per-parameter AccumulateGrad work."""
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad and p.grad is None and p.is_cuda and p.dtype == torch.float32]
        if not ps:
            return
        total = sum((p.numel() + 3) // 4 * 4 for p in ps)
        flat = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            v = flat[off: off + p.numel()].view_as(p)
            p.grad = v
            views.append((p, v))
            off += (p.numel() + 3) // 4 * 4
        self._flat = (flat, views)

    def zero_grad(self, set_to_none: bool = True) -> None:
        """torch's contract, except that gradients this optimizer owns are zeroed IN PLACE (engine/training_engine.py:224,307 passes
        set_to_none=True; dropping them would move every gradient address each iteration)."""
        if self._flat is None:
            return super().zero_grad(set_to_none=set_to_none)
        flat, views = self._flat
        owned = set()
        for p, v in views:
            owned.add(id(p))
            g = p.grad
            if g is None or g.data_ptr() != v.data_ptr():  # somebody replaced it (autograd allocates a fresh tensor for a None gradient)
                p.grad = v
        flat.zero_()
        for g in self.param_groups:
            for p in g["params"]:
                if id(p) not in owned and p.grad is not None:
                    if p.grad.grad_fn is not None:
                        p.grad.detach_()
                    p.grad.zero_()  # (e.g. views of cvnets_amd.ddp's buckets: stay in place as well)

    # -------------------------------------------------------------------------------------------------
    def _build(self):
        entries, skipped = [], []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if p.grad is None:  # torch.optim.AdamW skips parameters without a gradient (no decay of weights or moments): so do we
                    skipped.append(p)
                else:
                    entries.append((p, gi))
        if not entries:
            raise RuntimeError("no parameter has a gradient")
        dev = entries[0][0].device
        if dev.type != "cuda":
            raise RuntimeError("cvnets_amd.optim.AdamW has no CPU path")
        ema_by_id = {}
        if self._ema is not None:  # pair parameters by registration order (EMA model = deepcopy of the model)
            model, ema_model = self._ema
            for p, e in zip(model.parameters(), ema_model.parameters()):
                if p.shape != e.shape:
                    raise RuntimeError("EMA model does not mirror the model")
                ema_by_id[id(p)] = e
        rows, offsets, off = [], [], 0
        for p, gi in entries:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous float32")
            e = ema_by_id.get(id(p))
            rows.append([p.data_ptr(), p.grad.data_ptr(), e.data_ptr() if e is not None else 0, off, p.numel(), gi, off, 0])
            offsets.append(off)
            off += (p.numel() + 3) // 4 * 4  # every tensor starts on a 4-element boundary: the kernel moves float4
        rows.append([0, 0, 0, 0, 0, 0, off, 0])
        plan = {
            "entries": entries, "skipped": skipped, "offsets": offsets, "total": off, "n": len(entries), "device": dev,
            "table": torch.tensor(rows, dtype=torch.int64, device=dev),
            "m": torch.zeros(off, dtype=torch.float32, device=dev), "v": torch.zeros(off, dtype=torch.float32, device=dev),
            "step": torch.zeros(1, dtype=torch.float32, device=dev),
            "hp": torch.zeros(len(self.param_groups), 2, dtype=torch.float32, device=dev),
            "hp_host": None, "grad_ptrs": [p.grad.data_ptr() for p, _ in entries],
        }
        old = self._plan
        if old is not None:  # gradients were re-allocated / the set of live parameters changed: carry the moments over per parameter
            prev = {id(p): o for (p, _), o in zip(old["entries"], old["offsets"])}
            for (p, _), o in zip(entries, offsets):
                po = prev.get(id(p))
                if po is not None:
                    plan["m"][o: o + p.numel()].copy_(old["m"][po: po + p.numel()])
                    plan["v"][o: o + p.numel()].copy_(old["v"][po: po + p.numel()])
            plan["step"] = old["step"]
            self._rebuilds += 1
            if self._rebuilds == 3:
                import warnings
                warnings.warn("cvnets_amd.optim.AdamW re-plans every step because the gradient tensors keep moving (zero_grad(set_to_none=True)?): "
                              "keep gradients in place (set_to_none=False or cvnets_amd.ddp's flat buckets) for the one-launch fast path")
        self._plan = plan
        self._apply_loaded_state()
        return plan

    def _valid(self, plan) -> bool:
        return all(p.grad is not None and p.grad.data_ptr() == gp for (p, _), gp in zip(plan["entries"], plan["grad_ptrs"])) \
            and all(p.grad is None for p in plan["skipped"])

    # ---- checkpointing: torch.optim.AdamW's state layout (step / exp_avg / exp_avg_sq per parameter index) -------------------------
    def _param_index(self):
        idx, i = {}, 0
        for g in self.param_groups:
            for p in g["params"]:
                idx[id(p)] = i
                i += 1
        return idx

    def state_dict(self):
        sd = super().state_dict()
        plan = self._plan
        if plan is not None:
            idx = self._param_index()
            step = plan["step"].detach().clone().reshape(())
            sd["state"] = {idx[id(p)]: {"step": step.clone(), "exp_avg": plan["m"][o: o + p.numel()].view_as(p).clone(),
                                        "exp_avg_sq": plan["v"][o: o + p.numel()].view_as(p).clone()}
                           for (p, _), o in zip(plan["entries"], plan["offsets"])}
        elif self._loaded_state:
            sd["state"] = self._loaded_state
        return sd

    def load_state_dict(self, state_dict):
        loaded = dict(state_dict.get("state", {}))
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        self._loaded_state = {int(k): v for k, v in loaded.items()}
        if self._plan is not None:
            self._apply_loaded_state()
            self._plan["hp_host"] = None  # the groups' lr / weight_decay may have changed

    def _apply_loaded_state(self):
        if not self._loaded_state or self._plan is None:
            return
        plan, idx = self._plan, self._param_index()
        steps = []
        for (p, _), o in zip(plan["entries"], plan["offsets"]):
            st = self._loaded_state.get(idx[id(p)])
            if st is None:
                continue
            plan["m"][o: o + p.numel()].copy_(st["exp_avg"].reshape(-1).to(plan["m"]))
            plan["v"][o: o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1).to(plan["v"]))
            steps.append(float(st["step"]))
        if steps:
            plan["step"].fill_(max(steps))
        self._loaded_state = {}

    def sync_hyperparameters(self) -> None:
        """copy the per-group (lr, weight_decay) to the device table; call after a scheduler changed them (outside graph replay)."""
        plan = self._plan or self._build()
        host = [[float(g["lr"]), float(g["weight_decay"])] for g in self.param_groups]
        if host != plan["hp_host"]:
            plan["hp"].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=True)  # plumbing: 2 floats per group
            plan["hp_host"] = host

    @torch.no_grad()
    def step(self, closure=None, inv_grad_scale: Optional[torch.Tensor] = None, sync_hyperparameters: bool = True):
        loss = closure() if closure is not None else None
        ops.finish_backward()  # side-stream dW work / deferred reductions of a backward whose end-of-backward callback was lost
        import sys
        _ddp = sys.modules.get(__package__ + ".ddp")
        if _ddp is not None:
            _ddp.exchange_pending()  # a hooked forward whose backward never reached the gradient exchange: run it before the update
        if not any(p.grad is not None for g in self.param_groups for p in g["params"]):
            return loss  # torch.optim.AdamW.step() with no gradients anywhere is a no-op
        plan = self._plan
        if plan is None or not self._valid(plan):
            plan = self._build()  # carries the moments of the previous plan over, parameter by parameter
        if sync_hyperparameters:
            self.sync_hyperparameters()
        g0 = self.param_groups[0]
        _lib.call("cvh_adamw_multi", _p(plan["table"]), plan["n"], plan["total"], _p(plan["m"]), _p(plan["v"]), _p(plan["hp"]),
                  float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), _p(plan["step"]), _p(inv_grad_scale),
                  self.ema_momentum if self._ema is not None else 0.0, _stream())
        ops.invalidate_packed()  # parameters (and EMA parameters) changed through raw pointers: cached packed weights are stale
        return loss


class EMABuffers:
    """EMA of the model's floating-point BUFFERS (BatchNorm running statistics), the part of EMA.update_parameters that is not a
    parameter (averaging_utils.py:47-55 iterates the whole state_dict); integer buffers (num_batches_tracked) are copied."""

    def __init__(self, model: torch.nn.Module, ema_model: torch.nn.Module, momentum: float = 0.0005):
        rows, off, self.ints = [], 0, []
        for (k, s), (k2, d) in zip(model.named_buffers(), ema_model.named_buffers()):
            if k != k2 or s.shape != d.shape:
                raise RuntimeError("EMA model does not mirror the model's buffers")
            if s.dtype == torch.float32:
                rows.append([d.data_ptr(), s.data_ptr(), s.numel(), off])
                off += s.numel()
            else:
                self.ints.append((d, s))
        rows.append([0, 0, 0, off])
        self.n, self.total, self.momentum = len(rows) - 1, off, float(momentum)
        self.table = torch.tensor(rows, dtype=torch.int64, device=next(model.buffers()).device) if self.n else None
        self._keep = (model, ema_model)

    @torch.no_grad()
    def update(self):
        if self.n:
            _lib.call("cvh_lerp_multi", _p(self.table), self.n, self.total, self.momentum, _stream())
            ops.invalidate_packed()
        for d, s in self.ints:  # averaging an integer counter is a copy after the cast back (ema_v.copy_(...) on an int64 tensor)
            d.copy_((d * (1.0 - self.momentum) + self.momentum * s).to(d.dtype))
