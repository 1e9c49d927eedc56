"""Device-side input stage: RandomMixup / RandomCutmix and ``apply_mixing_transforms`` of
data/transforms/image_torch.py (called at engine/training_engine.py:238 on the batch already moved to the device), as ONE HIP pass
that can also deliver the NHWC compute-dtype tensor the models consume (mixing + channels_last conversion + fp32 -> bf16 cast fused:
``cvh_mix_batch``).

The host-side random draws are the reference's, call for call (``torch.rand(1)`` gate, ``torch._sample_dirichlet`` for lambda,
``torch.randint`` for the box centre, ``random.choice`` between the two transforms), so a seeded run picks the same lambda / box as the
reference; the per-pixel work and the one-hot target mixing run on the GPU.
"""
from __future__ import annotations

import math
import random
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .layers import opt


def _soft_targets(target: Tensor, num_classes: int, lam: float, dtype=torch.float32) -> Tensor:
    """lam * onehot(y) + (1 - lam) * onehot(roll(y, 1))  — plumbing: [B, num_classes] scatter on the device"""
    B = target.shape[0]
    out = torch.zeros(B, num_classes, dtype=dtype, device=target.device)
    out.scatter_add_(1, target.view(B, 1), torch.full((B, 1), float(lam), dtype=dtype, device=target.device))
    out.scatter_add_(1, target.roll(1, 0).view(B, 1), torch.full((B, 1), 1.0 - float(lam), dtype=dtype, device=target.device))
    return out


class _MixBase:
    name = ""

    def __init__(self, opts, num_classes: int, *args, **kwargs) -> None:
        self.num_classes = num_classes
        self.alpha = opt(opts, f"image_augmentation.{self.name}.alpha", 1.0)
        self.p = opt(opts, f"image_augmentation.{self.name}.p", 1.0)
        self.inplace = opt(opts, f"image_augmentation.{self.name}.inplace", False)
        self.sample_key = opt(opts, f"image_augmentation.{self.name}.sample_key", None)
        self.target_key = opt(opts, f"image_augmentation.{self.name}.target_key", None)
        if not (num_classes > 0 and self.alpha > 0.0 and 0.0 < self.p <= 1.0):
            raise ValueError(f"{self.__class__.__name__}: need num_classes > 0, alpha > 0 and 0 < p <= 1")
        # extension over the reference: hand the model its NHWC compute-dtype tensor directly (None: NCHW float32 like the reference)
        self.to_nhwc_dtype: Optional[torch.dtype] = None

    def _draw(self, W: int, H: int) -> Tuple[float, Optional[Tuple[int, int, int, int]]]:
        raise NotImplementedError

    def apply(self, image_tensor: Tensor, target_tensor: Tensor) -> Tuple[Tensor, Tensor]:
        if image_tensor.ndim != 4 or target_tensor.ndim != 1:
            raise ValueError("Batch ndim should be 4 and target ndim 1")
        if not image_tensor.is_floating_point() or target_tensor.dtype != torch.int64:
            raise ValueError("Batch must be floating point and targets int64")
        lam, box = self._draw(image_tensor.shape[3], image_tensor.shape[2])
        mixed = ops.mix_batch(image_tensor, lam, box, to_nhwc_dtype=self.to_nhwc_dtype)
        return mixed, _soft_targets(target_tensor, self.num_classes, lam, dtype=image_tensor.dtype)

    def __call__(self, data: Dict) -> Dict:
        if torch.rand(1).item() >= self.p:
            return data
        samples, targets = data.pop("samples"), data.pop("targets")
        s = samples[self.sample_key] if self.sample_key is not None else samples
        t = targets[self.target_key] if self.target_key is not None else targets
        s, t = self.apply(s, t)
        if self.sample_key is not None:
            samples[self.sample_key] = s
        else:
            samples = s
        if self.target_key is not None:
            targets[self.target_key] = t
        else:
            targets = t
        data.update({"samples": samples, "targets": targets})
        return data

    def __repr__(self) -> str:
        return "{}(num_classes={}, p={}, alpha={}, inplace={})".format(self.__class__.__name__, self.num_classes, self.p, self.alpha, self.inplace)


class RandomMixup(_MixBase):
    """data/transforms/image_torch.py:21-160"""
    name = "mixup"

    def _draw(self, W, H):
        lam = float(torch._sample_dirichlet(torch.tensor([self.alpha, self.alpha]))[0])
        return lam, None


class RandomCutmix(_MixBase):
    """data/transforms/image_torch.py:211-336"""
    name = "cutmix"

    def _draw(self, W, H):
        lam = float(torch._sample_dirichlet(torch.tensor([self.alpha, self.alpha]))[0])
        r_x = torch.randint(W, (1,))
        r_y = torch.randint(H, (1,))
        r = 0.5 * math.sqrt(1.0 - lam)
        r_w_half, r_h_half = int(r * W), int(r * H)
        x1 = int(torch.clamp(r_x - r_w_half, min=0))
        y1 = int(torch.clamp(r_y - r_h_half, min=0))
        x2 = int(torch.clamp(r_x + r_w_half, max=W))
        y2 = int(torch.clamp(r_y + r_h_half, max=H))
        lam = float(1.0 - (x2 - x1) * (y2 - y1) / (W * H))  # the target weight follows the box that was actually pasted
        return lam, (x1, y1, x2, y2)


def apply_mixing_transforms(opts, data: Dict, to_nhwc_dtype: Optional[torch.dtype] = None) -> Dict:
    """data/transforms/image_torch.py:416-463: if both transforms are enabled one is chosen at random and applied (with its own p)."""
    transforms = []
    n_classes = opt(opts, "model.classification.n_classes", None)
    for cls, key in ((RandomMixup, "mixup"), (RandomCutmix, "cutmix")):
        if opt(opts, f"image_augmentation.{key}.enable", False):
            if n_classes is None:
                raise ValueError("Please specify number of classes. Got None.")
            t = cls(opts=opts, num_classes=n_classes)
            t.to_nhwc_dtype = to_nhwc_dtype
            transforms.append(t)
    if transforms:
        data = random.choice(transforms)(data)
    return data
