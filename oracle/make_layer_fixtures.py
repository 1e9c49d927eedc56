"""ORACLE tooling — test infrastructure, NOT product code.

Generates tests/golden/layernorm_channel_first.npz by RUNNING THE REFERENCE's LayerNorm (cvnets/layers/normalization/layer_norm.py:51-66) on
genuine [B, C, H, W] feature maps on CPU in fp32: the channel-first branch, (x - mean_c) / (std_c + eps) * weight[c] + bias[c] per pixel.
Runs only in the authoring container (needs /root/reference).  Inputs / weights are the seeded values of oracle/weights.py, so the GPU
test regenerates them without the reference; the fixture keeps the outputs and the gradients of the input, weight and bias under
loss = <y, g>.

    python oracle/make_layer_fixtures.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle.make_golden import REF  # noqa: E402,F401  (puts the shim + /root/reference on sys.path)
from oracle.weights import seeded_input  # noqa: E402

# name, B, C, H, W
LN_CF_CASES = [("c64", 2, 64, 6, 5), ("c144", 3, 144, 4, 4), ("c8", 2, 8, 3, 7)]


def ln_cf_tensors(name, B, C, H, W):
    """seeded input, weight, bias, upstream gradient of one case (shared with tests/test_layernorm_cf_gpu.py)"""
    k = sum(ord(ch) for ch in name)
    x = seeded_input((B, C, H, W), seed=300 + k) * 1.5 + 0.25
    w = 1.0 + 0.2 * seeded_input((C,), seed=301 + k)
    b = 0.1 * seeded_input((C,), seed=302 + k)
    g = seeded_input((B, C, H, W), seed=303 + k)
    return x, w, b, g


def main():
    cwd = os.getcwd()
    os.chdir(REF)
    from cvnets.layers.normalization.layer_norm import LayerNorm
    os.chdir(cwd)
    out = {}
    for name, B, C, H, W in LN_CF_CASES:
        x, w, b, g = ln_cf_tensors(name, B, C, H, W)
        ln = LayerNorm(C, eps=1e-5)
        with torch.no_grad():
            ln.weight.copy_(w)
            ln.bias.copy_(b)
        x = x.clone().requires_grad_(True)
        y = ln(x)
        assert y.shape == x.shape
        # the branch really is the channel-first one: compare with the written-out formula
        s, u = torch.std_mean(x.detach(), dim=1, keepdim=True, unbiased=False)
        assert float((y.detach() - ((x.detach() - u) / (s + 1e-5) * w.view(1, C, 1, 1) + b.view(1, C, 1, 1))).abs().max()) < 1e-5
        (y * g).sum().backward()
        out[name + "_y"] = y.detach().numpy()
        out[name + "_dx"] = x.grad.numpy()
        out[name + "_dw"] = ln.weight.grad.numpy()
        out[name + "_db"] = ln.bias.grad.numpy()
        print(name, "y", float(y.abs().max()), "dx", float(x.grad.abs().max()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "layernorm_channel_first.npz"), **out)


if __name__ == "__main__":
    main()
