"""TEST INFRASTRUCTURE (never imported by the product path): the fp32 oracle of oracle/mobilevit_oracle.py evaluated with bf16 ROUNDING
POINTS — every value the HIP path keeps as a bf16 tensor in HBM, or hands to the matrix pipe as a bf16 operand, is rounded to bf16
(round-to-nearest-even, `x.bfloat16().float()`) at the same place; everything between those points stays fp32, exactly as inside the
kernels.

Why: bf16 HIP results differ from the fp32 reference by 2e-2 (logits) / 7e-2 (gradients) on MobileViT-S — as much as the reference's own
bf16-autocast run does — so comparing against fp32 cannot tell storage noise from an implementation error of that size.  Against THIS
evaluation the storage noise cancels (same rounding points, only fp32 summation orders differ), and what remains measures the
implementation: tests/test_bf16_parity_gpu.py asserts the HIP bf16 step is several times closer to it than to the fp32 reference.

How: a TorchFunctionMode intercepts the torch calls the oracle makes (it shares no code with the product and is not modified):
  conv2d / linear      activation input rounded (a tensor read from HBM is bf16), weight rounded (the packed bf16 copy the GEMMs read; the
                       weight GRADIENT stays fp32), output rounded (conv / linear outputs are bf16 tensors: DESIGN §3)
  silu / gelu          output rounded (BatchNorm + activation are fused: the normalised value is never rounded on its own, the activated one
                       is a bf16 operand of the next kernel)
  layer_norm, softmax  input and output rounded (LN output is a bf16 tensor; the probabilities enter the P.V MFMA as bf16)
  tensor + tensor      output rounded (residual sums are stored)
  matmul               the P.V product's output is rounded (attention output O), the scores Q.K^T stay fp32 (they never leave registers)
  mean                 output rounded (global pool output)
The rounding function also rounds the GRADIENT flowing back through the same point (the gradient of a bf16 tensor is a bf16 tensor).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.overrides import TorchFunctionMode

from . import mobilevit_oracle as orc


class _Round(torch.autograd.Function):
    """value and incoming gradient rounded to bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundValue(torch.autograd.Function):
    """value rounded to bf16, gradient untouched (weights: the gradient of the fp32 master copy stays fp32)"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def q(x: Tensor) -> Tensor:
    return _Round.apply(x) if (isinstance(x, Tensor) and x.is_floating_point()) else x


def qw(w: Tensor) -> Tensor:
    return _RoundValue.apply(w) if (isinstance(w, Tensor) and w.is_floating_point()) else w


class Bf16Points(TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self.n_matmul = 0
        self.counts: Dict[str, int] = {}

    def _hit(self, name):
        self.counts[name] = self.counts.get(name, 0) + 1

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (F.conv2d, F.linear):
            self._hit(func.__name__)
            a = list(args)
            a[0] = q(a[0])
            a[1] = qw(a[1])
            return q(func(*a, **kwargs))
        if func in (F.silu, F.gelu):
            self._hit(func.__name__)
            return q(func(*args, **kwargs))
        if func is F.layer_norm:
            self._hit("layer_norm")
            a = list(args)
            a[0] = q(a[0])
            return q(func(*a, **kwargs))
        if func is torch.softmax or func is F.softmax or func is Tensor.softmax:
            self._hit("softmax")
            return q(func(*args, **kwargs))
        if func is torch.matmul or func is Tensor.matmul:
            self.n_matmul += 1
            out = func(*args, **kwargs)
            if self.n_matmul % 2 == 0:  # multi_head_attention: scores first (fp32), then P.V (bf16 tensor O)
                self._hit("matmul_pv")
                return q(out)
            return out
        if func in (Tensor.add, Tensor.__add__, torch.add):
            out = func(*args, **kwargs)
            if len(args) == 2 and isinstance(args[0], Tensor) and isinstance(args[1], Tensor) and args[0].shape == args[1].shape \
                    and args[0].is_floating_point() and args[0].dim() >= 3:
                self._hit("residual_add")
                return q(out)
            return out
        if func is torch.mean or func is Tensor.mean:
            self._hit("mean")
            return q(func(*args, **kwargs))
        return func(*args, **kwargs)


def train_step(sd: Dict[str, Tensor], x: Tensor, y: Tensor, mode: str = "small", label_smoothing: float = 0.1):
    """oracle.mobilevit_oracle.train_step (fwd + loss + bwd of MobileViT) with bf16 rounding points; same return values"""
    m = Bf16Points()
    with m:
        out = orc.train_step(sd, q(x), y, mode=mode, label_smoothing=label_smoothing)
    assert m.counts.get("conv2d", 0) > 0 and m.counts.get("residual_add", 0) > 0 and m.counts.get("matmul_pv", 0) > 0, m.counts
    return out
