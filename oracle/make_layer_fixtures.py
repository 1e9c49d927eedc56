"""ORACLE tooling — test infrastructure, NOT product code.

Layer-level fixtures written by RUNNING THE REFERENCE's own classes on CPU in fp32 (authoring container only: needs /root/reference).

(1) tests/golden/mobilevitv2_block_temporal.npz: MobileViTBlockv2.forward((x, x_prev)) -> forward_temporal (cvnets/modules/mobilevit_block.py:
628-655; LinearAttnFFN with x_prev, cvnets/modules/transformer.py:246-264; LinearSelfAttention._forward_cross_attn, cvnets/layers/
linear_attention.py:163-207), two chained frames in train mode: fm1, p1 = block((x1, None)); fm2, p2 = block((x2, p1)); loss = <fm2, g> +
<p2, gp>; outputs and the gradients of every parameter and of both inputs (p1 is not detached).

(2) tests/golden/layernorm_channel_first.npz by RUNNING THE REFERENCE's LayerNorm (cvnets/layers/normalization/layer_norm.py:51-66) on
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
from oracle.weights import seeded_input, seeded_state_dict  # noqa: E402

# name, batch, in_channels, attn_unit_dim, ffn_multiplier, blocks, patch, H, W
V2T_CASES = [("p2", 2, 32, 64, 2.0, 2, 2, 12, 16), ("p4", 2, 48, 96, 2.0, 1, 4, 16, 16)]


def reference_v2_block(cin, d, ffn_mult, blocks, patch):
    import argparse
    cwd = os.getcwd()
    os.chdir(REF)
    import cvnets
    from cvnets.modules import MobileViTBlockv2

    opts = cvnets.modeling_arguments(argparse.ArgumentParser()).parse_args([])
    setattr(opts, "model.normalization.name", "batch_norm")
    setattr(opts, "model.normalization.momentum", 0.1)
    setattr(opts, "model.activation.name", "swish")
    setattr(opts, "model.layer.conv_init", "kaiming_normal")
    setattr(opts, "model.layer.linear_init", "trunc_normal")
    block = MobileViTBlockv2(opts, in_channels=cin, attn_unit_dim=d, ffn_multiplier=ffn_mult, n_attn_blocks=blocks, patch_h=patch, patch_w=patch,
                             attn_dropout=0.0, dropout=0.0, ffn_dropout=0.0)
    os.chdir(cwd)
    return block


def v2_temporal():
    out = {}
    for name, b, cin, d, fm_, blocks, patch, H, W in V2T_CASES:
        block = reference_v2_block(cin, d, fm_, blocks, patch).train()
        shapes = {k: tuple(v.shape) for k, v in block.state_dict().items()}
        block.load_state_dict(seeded_state_dict(shapes, seed=23), strict=True)
        x1 = seeded_input((b, cin, H, W), seed=51).requires_grad_(True)
        x2 = seeded_input((b, cin, H, W), seed=52).requires_grad_(True)
        fm1, p1 = block((x1, None))
        fm2, p2 = block((x2, p1))
        g = seeded_input(tuple(fm2.shape), seed=53)
        gp = seeded_input(tuple(p2.shape), seed=54)
        loss = (fm2 * g).sum() + (p2 * gp).sum()
        params = dict(block.named_parameters())
        grads = torch.autograd.grad(loss, [x1, x2] + list(params.values()))
        out[f"{name}::cfg"] = np.array([b, cin, d, int(fm_), blocks, patch, H, W])
        for k, v in (("fm1", fm1), ("p1", p1), ("fm2", fm2), ("p2", p2), ("grad_x1", grads[0]), ("grad_x2", grads[1])):
            out[f"{name}::{k}"] = v.detach().numpy()
        for k, gr in zip(params.keys(), grads[2:]):
            out[f"{name}::grad::{k}"] = gr.numpy()
        out[f"{name}::keys"] = np.array(list(shapes.keys()))
        out[f"{name}::shapes"] = np.array([",".join(str(i) for i in s) for s in shapes.values()])
        print(f"v2 temporal {name}: fm2 {tuple(fm2.shape)} patches {tuple(p2.shape)} loss {float(loss):.5f} |grad_x1| {float(grads[0].norm()):.4f} "
              f"({len(params)} parameter gradients)")
    path = os.path.join(REPO, "tests", "golden", "mobilevitv2_block_temporal.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")

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
    v2_temporal()
