"""ORACLE tooling — test infrastructure, NOT product code.

Generates tests/golden/mobilevit_block_temporal.npz by RUNNING THE REFERENCE's MobileViTBlock (cvnets/modules/mobilevit_block.py:289-326,
`forward((x, x_prev))` -> forward_temporal) on CPU in fp32.  Runs only in the authoring container (needs /root/reference).

Two frames per case, the way the spatio-temporal model chains the block:

    fm1, p1 = block((x1, None))        # first frame: plain self-attention, returns its patches [B*P, N, d]
    fm2, p2 = block((x2, p1))          # second frame: every TransformerEncoder takes keys / values from p1

in train mode (batch statistics in the BatchNorms), dropout off, with loss = <fm2, g> + <p2, gp>; the fixture keeps fm1, p1, fm2, p2 and the
gradients of every parameter and of both inputs (they flow through BOTH frames: p1 is not detached).  Weights / inputs are the seeded
values of oracle/weights.py, so the GPU test regenerates them without the reference.

    python oracle/make_temporal_fixture.py
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle.make_golden import REF  # noqa: E402,F401  (puts the shim + /root/reference on sys.path)
from oracle.weights import seeded_input, seeded_state_dict  # noqa: E402

# name, batch, in_channels, transformer_dim, ffn_dim, blocks, head_dim, patch, H, W
CASES = [
    ("even", 2, 32, 64, 128, 2, 16, 2, 12, 12),     # 36 patches of 4 pixels
    ("resized", 2, 32, 48, 96, 2, 16, 2, 13, 11),   # not a multiple of the patch: bilinear resize to 14 x 12 and back (mobilevit_block.py:191-200)
]


def reference_block(cin, d, ffn, blocks, head_dim, patch):
    cwd = os.getcwd()
    os.chdir(REF)
    import cvnets
    from cvnets.modules import MobileViTBlock

    opts = cvnets.modeling_arguments(argparse.ArgumentParser()).parse_args([])
    setattr(opts, "model.normalization.name", "batch_norm")
    setattr(opts, "model.normalization.momentum", 0.1)
    setattr(opts, "model.activation.name", "swish")
    setattr(opts, "model.layer.conv_init", "kaiming_normal")
    setattr(opts, "model.layer.linear_init", "trunc_normal")
    block = MobileViTBlock(opts, in_channels=cin, transformer_dim=d, ffn_dim=ffn, n_transformer_blocks=blocks, head_dim=head_dim,
                           patch_h=patch, patch_w=patch, attn_dropout=0.0, dropout=0.0, ffn_dropout=0.0)
    os.chdir(cwd)
    return block


def oracle_errors(sd, heads, blocks, patch, inputs, ref_out, ref_grads, names):
    """(worst output, worst gradient) relative L2 distance of the oracle's two chained frames from the reference's"""
    from oracle import mobilevit_oracle as orc

    x1, x2, g, gp = (t.detach().clone() for t in inputs)
    x1.requires_grad_(True), x2.requires_grad_(True)
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    fm1, p1 = orc.mobilevit_block_temporal(leaf, "", x1, None, blocks, heads, True, None, patch, patch)
    fm2, p2 = orc.mobilevit_block_temporal(leaf, "", x2, p1, blocks, heads, True, None, patch, patch)
    loss = (fm2 * g).sum() + (p2 * gp).sum()
    grads = torch.autograd.grad(loss, [x1, x2] + [leaf[k] for k in names])
    rel = lambda a, b: float((a.detach() - b.detach()).norm() / b.detach().norm())  # noqa: E731
    return (max(rel(a, b) for a, b in zip((fm1, p1, fm2, p2), ref_out)), max(rel(a, b) for a, b in zip(grads, ref_grads)))


def main():
    out = {}
    for name, b, cin, d, ffn, blocks, hd, patch, H, W in CASES:
        block = reference_block(cin, d, ffn, blocks, hd, patch).train()
        shapes = {k: tuple(v.shape) for k, v in block.state_dict().items()}
        sd = seeded_state_dict(shapes, seed=21)
        block.load_state_dict(sd, strict=True)
        x1 = seeded_input((b, cin, H, W), seed=31).requires_grad_(True)
        x2 = seeded_input((b, cin, H, W), seed=32).requires_grad_(True)
        fm1, p1 = block((x1, None))
        fm2, p2 = block((x2, p1))
        g = seeded_input(tuple(fm2.shape), seed=33)
        gp = seeded_input(tuple(p2.shape), seed=34)
        loss = (fm2 * g).sum() + (p2 * gp).sum()
        params = dict(block.named_parameters())
        grads = torch.autograd.grad(loss, [x1, x2] + list(params.values()))
        # pin the oracle restatement (oracle/mobilevit_oracle.py: mobilevit_block_temporal) on what the reference just computed
        errs = oracle_errors(sd, d // hd, blocks, patch, (x1, x2, g, gp), (fm1, p1, fm2, p2), grads, list(params.keys()))
        print(f"{name}: oracle vs reference — outputs {errs[0]:.2e}, gradients {errs[1]:.2e}")
        assert errs[0] < 1e-5 and errs[1] < 1e-4, errs
        out[f"{name}::cfg"] = np.array([b, cin, d, ffn, blocks, hd, patch, H, W])
        for k, v in (("fm1", fm1), ("p1", p1), ("fm2", fm2), ("p2", p2), ("grad_x1", grads[0]), ("grad_x2", grads[1])):
            out[f"{name}::{k}"] = v.detach().numpy()
        for k, gr in zip(params.keys(), grads[2:]):
            out[f"{name}::grad::{k}"] = gr.numpy()
        out[f"{name}::keys"] = np.array(list(shapes.keys()))
        out[f"{name}::shapes"] = np.array([",".join(str(i) for i in s) for s in shapes.values()])
        print(f"{name}: fm2 {tuple(fm2.shape)} patches {tuple(p2.shape)} loss {float(loss):.5f} |grad_x1| {float(grads[0].norm()):.4f} "
              f"({len(params)} parameter gradients)")
    path = os.path.join(REPO, "tests", "golden", "mobilevit_block_temporal.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
