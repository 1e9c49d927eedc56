"""MobileViTBlockv2.forward((x, x_prev)) — the temporal path of the MobileViTv2 block (cvnets/modules/mobilevit_block.py:628-655, LinearAttnFFN with
x_prev: cvnets/modules/transformer.py:246-264, LinearSelfAttention._forward_cross_attn: cvnets/layers/linear_attention.py:163-207) — against the
REFERENCE's own outputs (tests/golden/mobilevitv2_block_temporal.npz, written by oracle/make_layer_fixtures.py from the reference run on CPU in
fp32): two chained frames in train mode, the second one taking query / key from the first one's patches; both feature maps, both patch tensors
(the reference's [B, C, P, N] layout) and the gradient of every parameter and of both inputs (which flow through both frames)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mobilevitv2_block_temporal.npz")

OUT_TOL = 2e-4   # rel-L2, fp32 compute (summation order only)
GRAD_TOL = 2e-3  # rel-L2 per tensor, as for the model-level fp32 parity tests


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("case", ["p2", "p4"])
def test_v2_temporal_block_matches_the_reference(case):
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlockv2
    from oracle.weights import seeded_input, seeded_state_dict

    gold = np.load(GOLD)
    b, cin, d, ffn_mult, blocks, patch, H, W = (int(v) for v in gold[f"{case}::cfg"])
    block = MobileViTBlockv2(default_opts(), in_channels=cin, attn_unit_dim=d, ffn_multiplier=float(ffn_mult), n_attn_blocks=blocks, patch_h=patch,
                             patch_w=patch)
    shapes = {k: tuple(v.shape) for k, v in block.state_dict().items()}
    ref_shapes = {str(k): tuple(int(i) for i in str(s).split(",") if i) for k, s in zip(gold[f"{case}::keys"], gold[f"{case}::shapes"])}
    assert shapes == ref_shapes  # same state-dict keys and shapes as the reference block
    block.load_state_dict(seeded_state_dict(shapes, seed=23))
    block = block.to(DEV).train()
    x1 = seeded_input((b, cin, H, W), seed=51).to(DEV).requires_grad_(True)
    x2 = seeded_input((b, cin, H, W), seed=52).to(DEV).requires_grad_(True)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        fm1, p1 = block((x1, None))
        fm2, p2 = block((x2, p1))
        g = seeded_input(tuple(fm2.shape), seed=53).to(DEV)
        gp = seeded_input(tuple(p2.shape), seed=54).to(DEV)
        loss = (fm2.float() * g).sum() + (p2.float() * gp).sum()
        params = dict(block.named_parameters())
        grads = torch.autograd.grad(loss, [x1, x2] + list(params.values()))
        torch.cuda.synchronize()
    finally:
        cvnets_amd.set_compute_dtype(None)
    for name, got in (("fm1", fm1), ("p1", p1), ("fm2", fm2), ("p2", p2)):
        want = torch.from_numpy(gold[f"{case}::{name}"])
        assert tuple(got.shape) == tuple(want.shape), name
        assert _rel(got.detach().float().cpu(), want) < OUT_TOL, (name, _rel(got.detach().float().cpu(), want))
    worst = ("", 0.0)
    for name, got in zip(["grad_x1", "grad_x2"] + ["grad::" + k for k in params], grads):
        want = torch.from_numpy(gold[f"{case}::{name}"])
        e = _rel(got.float().cpu(), want)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] < GRAD_TOL, worst


def test_v2_temporal_first_frame_is_the_spatial_block():
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlockv2

    torch.manual_seed(5)
    block = MobileViTBlockv2(default_opts(), in_channels=32, attn_unit_dim=64, n_attn_blocks=2, patch_h=2, patch_w=2).to(DEV).eval()
    x = torch.randn(3, 32, 16, 20, device=DEV)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        with torch.no_grad():
            a = block(x).float()
            fm, patches = block((x, None))
    finally:
        cvnets_amd.set_compute_dtype(None)
    assert tuple(patches.shape) == (3, 64, 4, 8 * 10)
    assert _rel(fm.float(), a) < 1e-5
