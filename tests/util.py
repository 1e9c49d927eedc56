import torch

TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().float()
    b = b.detach().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def l2_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double()
    b = b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, got, ref, dtype, scale=1.0):
    e = rel_err(got, ref)
    tol = TOL[dtype] * scale
    assert e == e and e < tol, f"{name}: rel err {e:.3e} >= {tol:.1e} (dtype {dtype})"
    return e


def nhwc(x: torch.Tensor, dtype) -> torch.Tensor:
    """NCHW tensor -> logical-NCHW tensor with NHWC memory, in `dtype` (test-side helper, ATen)."""
    return x.to(dtype).contiguous(memory_format=torch.channels_last)
