"""Host-side mirrors of the two loss functions that sit on the hot path's critical section between forward end and backward start
(SURVEY §8 row a14 and §8f "next" row 2): ``ContrastiveLossClip`` (loss_fn/multi_modal_img_text/contrastive_loss_clip.py:20-172) and
``CrossEntropy`` (loss_fn/classification/cross_entropy.py:16-100).  Same constructor / forward contracts as the reference classes;
the arithmetic runs in HIP kernels (cvh_scaled_ce_*, cvh_ce_*, the MFMA logits GEMMs) and the RCCL all-gather of ddp.py."""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor, nn

from . import ops
from .layers import opt


class ContrastiveLossClip(nn.Module):
    """loss_fn/multi_modal_img_text/contrastive_loss_clip.py:20-142.  The two [N, N*W] logit GEMMs run on the MFMA linear kernel,
    the scaled cross-entropies in cvh_scaled_ce_*; cross-rank features come from an autograd-aware RCCL all-gather (ddp.py)."""

    def __init__(self, opts, *args, **kwargs) -> None:
        super().__init__()
        self.rank = opt(opts, "ddp.rank", 0)
        self.use_distributed = opt(opts, "ddp.use_distributed", False)

    def _forward_clip(self, prediction: Dict[str, Tensor], *args, **kwargs) -> Dict[str, Tensor]:
        from .ddp import gather_all_features

        if not {"image", "text"}.issubset(prediction.keys()):
            raise KeyError(f"image and text are mandatory keys for {self.__class__.__name__}.")
        image_features, text_features = prediction.pop("image"), prediction.pop("text")
        logit_scale = prediction.pop("logit_scale", 1.0)
        if image_features is None or text_features is None:
            raise ValueError(f"Image / text features can't be None in {self.__class__.__name__}")
        if not isinstance(logit_scale, Tensor):
            logit_scale = torch.tensor(float(logit_scale), device=image_features.device)
        g_img, g_txt = image_features, text_features
        if self.use_distributed:
            g_img, g_txt = gather_all_features(image_features), gather_all_features(text_features)
        logits_per_image = ops.linear(image_features.contiguous(), g_txt)  # image @ gathered_text^T  (scale applied inside the CE kernel)
        logits_per_text = ops.linear(text_features.contiguous(), g_img)
        offset = image_features.shape[0] * self.rank
        text_loss = ops.scaled_cross_entropy(logits_per_text, logit_scale, offset) * 0.5
        image_loss = ops.scaled_cross_entropy(logits_per_image, logit_scale, offset) * 0.5
        return {"total_loss": image_loss + text_loss, "image_loss": image_loss, "text_loss": text_loss, "logit_scale": logit_scale}

    def forward(self, input_sample, prediction: Dict[str, Tensor], *args, **kwargs) -> Dict:
        if not self.training:
            return {"total_loss": torch.tensor(0.0, device=prediction["logit_scale"].device)}
        return self._forward_clip(prediction=prediction)


class CrossEntropy(nn.Module):
    """loss_fn/classification/cross_entropy.py:16-100 (+ base_classification_criteria.py forward): label-smoothed cross-entropy over the
    classifier logits, forward and backward in one HIP kernel each (cvh_ce_fwd / cvh_ce_bwd)."""

    def __init__(self, opts, *args, **kwargs) -> None:
        super().__init__()
        self.ignore_idx = opt(opts, "loss.classification.cross_entropy.ignore_index", -1)
        self.use_class_wts = opt(opts, "loss.classification.cross_entropy.class_weights", False)
        self.label_smoothing = opt(opts, "loss.classification.cross_entropy.label_smoothing", 0.0)
        if self.use_class_wts:
            raise NotImplementedError("class-weighted cross-entropy is not on the HIP hot path")

    def _compute_loss(self, prediction: Tensor, target: Tensor, *args, **kwargs) -> Tensor:
        return ops.cross_entropy(prediction, target, self.label_smoothing if self.training else 0.0, self.ignore_idx)

    def forward(self, input_sample, prediction, target: Tensor, *args, **kwargs) -> Tensor:
        if isinstance(prediction, Tensor):
            return self._compute_loss(prediction, target)
        if isinstance(prediction, dict):
            if prediction.get("logits") is None:
                raise KeyError(f"logits is a required key in {self.__class__.__name__} when prediction is a dictionary")
            return self._compute_loss(prediction["logits"], target)
        raise TypeError(f"Prediction should be either a Tensor or Dictionary[str, Tensor]. Got: {type(prediction)}")

    def extra_repr(self) -> str:
        return f"\n\t ignore_idx={self.ignore_idx}\n\t class_weighting={self.use_class_wts}\n\t label_smoothing={self.label_smoothing}"
