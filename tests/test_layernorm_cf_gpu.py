"""LayerNorm's channel-first branch on genuine [B, C, H, W] feature maps (cvnets/layers/normalization/layer_norm.py:51-66:
(x - mean_c) / (std_c + eps) * weight[c] + bias[c] per pixel) against the REFERENCE's own outputs and gradients
(tests/golden/layernorm_channel_first.npz, written by oracle/make_layer_fixtures.py from the reference class run on CPU in fp32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "layernorm_channel_first.npz")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dtype,tol_y,tol_g", [(torch.float32, 1e-5, 1e-4), (torch.bfloat16, 1e-2, 2e-2)])
@pytest.mark.parametrize("case", ["c64", "c144", "c8"])
def test_channel_first_layernorm_matches_the_reference(case, dtype, tol_y, tol_g):
    import cvnets_amd
    from cvnets_amd.layers import LayerNorm
    from oracle.make_layer_fixtures import LN_CF_CASES, ln_cf_tensors

    gold = np.load(GOLD)
    name, B, C, H, W = next(c for c in LN_CF_CASES if c[0] == case)
    x, w, b, g = ln_cf_tensors(name, B, C, H, W)
    ln = LayerNorm(C, eps=1e-5).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(w)
        ln.bias.copy_(b)
    cvnets_amd.set_compute_dtype(dtype)
    try:
        xg = x.to(DEV).requires_grad_(True)
        y = ln(xg)
        assert tuple(y.shape) == (B, C, H, W)
        (y.float() * g.to(DEV)).sum().backward()
        assert _rel(y.float(), torch.from_numpy(gold[case + "_y"])) < tol_y
        assert _rel(xg.grad, torch.from_numpy(gold[case + "_dx"])) < tol_g
        assert _rel(ln.weight.grad, torch.from_numpy(gold[case + "_dw"])) < tol_g
        assert _rel(ln.bias.grad, torch.from_numpy(gold[case + "_db"])) < tol_g
    finally:
        cvnets_amd.set_compute_dtype(None)


def test_token_tensor_still_takes_the_channel_last_branch():
    """[B, S, C] with S != C is the documented channel-last LayerNorm (layer_norm.py:67-68), as before."""
    from cvnets_amd.layers import LayerNorm

    torch.manual_seed(0)
    ln = LayerNorm(64).to(DEV)
    x = torch.randn(3, 10, 64, device=DEV)
    ref = torch.nn.functional.layer_norm(x, (64,), ln.weight, ln.bias, 1e-5)
    assert _rel(ln(x).float(), ref.detach().cpu()) < 1e-5
