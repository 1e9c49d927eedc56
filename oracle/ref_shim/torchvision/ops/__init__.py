import torch
from torch import nn


class StochasticDepth(nn.Module):
    """Same semantics as torchvision.ops.StochasticDepth (row/batch mode Bernoulli(1-p)/(1-p))."""

    def __init__(self, p: float, mode: str) -> None:
        super().__init__()
        self.p = p
        self.mode = mode

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        size = [x.shape[0]] + [1] * (x.ndim - 1) if self.mode == "row" else [1] * x.ndim
        noise = torch.empty(size, dtype=x.dtype, device=x.device).bernoulli_(keep)
        if keep > 0.0:
            noise.div_(keep)
        return x * noise


class _Placeholder:
    def __init__(self, *a, **k):
        raise NotImplementedError("torchvision shim: detection ops are not available")


class MultiScaleRoIAlign(_Placeholder):
    pass


def batched_nms(*a, **k):
    raise NotImplementedError("torchvision shim")


def __getattr__(name):  # PEP 562: anything else the reference's data / detection stack names at import time is an inert placeholder
    if name.startswith("__"):
        raise AttributeError(name)
    from autostub import _make
    return _make(name)
