"""MobileViTBlock.forward((x, x_prev)) — the block of the reference's spatio-temporal MobileViT (cvnets/modules/mobilevit_block.py:289-326) —
against the REFERENCE's own outputs (tests/golden/mobilevit_block_temporal.npz, written by oracle/make_temporal_fixture.py from the reference
run on CPU in fp32): two chained frames, the second one cross-attending to the first one's patches, train mode; both feature maps, both
patch tensors (in the reference's [B*P, N, d] order), and the gradient of every parameter and of both inputs (which flow through both frames).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mobilevit_block_temporal.npz")

OUT_TOL = 2e-4   # rel-L2, fp32 compute (summation order only)
GRAD_TOL = 2e-3  # rel-L2 per tensor, as for the model-level fp32 parity tests


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("case", ["even", "resized"])
def test_temporal_block_matches_the_reference(case):
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlock
    from oracle.weights import seeded_input, seeded_state_dict

    gold = np.load(GOLD)
    b, cin, d, ffn, blocks, hd, patch, H, W = (int(v) for v in gold[f"{case}::cfg"])
    block = MobileViTBlock(default_opts(), in_channels=cin, transformer_dim=d, ffn_dim=ffn, n_transformer_blocks=blocks, head_dim=hd,
                           patch_h=patch, patch_w=patch)
    shapes = {k: tuple(v.shape) for k, v in block.state_dict().items()}
    ref_shapes = {str(k): tuple(int(i) for i in str(s).split(",") if i) for k, s in zip(gold[f"{case}::keys"], gold[f"{case}::shapes"])}
    assert shapes == ref_shapes  # same state-dict keys and shapes as the reference block
    block.load_state_dict(seeded_state_dict(shapes, seed=21))
    block = block.to(DEV).train()
    x1 = seeded_input((b, cin, H, W), seed=31).to(DEV).requires_grad_(True)
    x2 = seeded_input((b, cin, H, W), seed=32).to(DEV).requires_grad_(True)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        fm1, p1 = block((x1, None))
        fm2, p2 = block((x2, p1))
        g = seeded_input(tuple(fm2.shape), seed=33).to(DEV)
        gp = seeded_input(tuple(p2.shape), seed=34).to(DEV)
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


def test_temporal_first_frame_is_the_spatial_block():
    """with x_prev = None the temporal path is the spatial block plus the returned patches: same feature map as forward(x) (which runs the
    sequences in place through the strided gather of the attention kernels instead of materialising [B*P, N, d])"""
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlock

    torch.manual_seed(5)
    block = MobileViTBlock(default_opts(), in_channels=32, transformer_dim=64, ffn_dim=128, n_transformer_blocks=2, head_dim=16,
                           patch_h=2, patch_w=2).to(DEV).eval()
    x = torch.randn(3, 32, 16, 20, device=DEV)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        with torch.no_grad():
            a = block(x).float()
            fm, patches = block((x, None))
    finally:
        cvnets_amd.set_compute_dtype(None)
    assert tuple(patches.shape) == (3 * 4, 8 * 10, 64)
    assert _rel(fm.float(), a) < 1e-5


@pytest.mark.parametrize("T,dtype", [(20, torch.float32), (50, torch.float32), (36, torch.bfloat16)])
def test_temporal_block_against_the_oracle_with_a_foreign_x_prev(T, dtype):
    """x_prev need not have the current frame's patch count (another resolution of the previous frame): T != N goes through the padded
    cross-attention; checked against the CPU oracle (pinned on the reference by oracle/make_temporal_fixture.py), outputs and the gradients
    of x, x_prev and every parameter.  bf16: rounding only (bounds = what fp32-accumulating bf16 storage gives on a block this size)."""
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlock
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_state_dict

    b, cin, d, ffn, blocks, hd, patch, H, W = 2, 32, 64, 128, 2, 16, 2, 12, 12
    block = MobileViTBlock(default_opts(), in_channels=cin, transformer_dim=d, ffn_dim=ffn, n_transformer_blocks=blocks, head_dim=hd,
                           patch_h=patch, patch_w=patch)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in block.state_dict().items()}, seed=22)
    block.load_state_dict(sd)
    block = block.to(DEV).train()
    names = [k for k, _ in block.named_parameters()]
    x = seeded_input((b, cin, H, W), seed=41)
    xp = seeded_input((b * patch * patch, T, d), seed=42)
    # oracle (CPU, fp32)
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    xo, xpo = x.clone().requires_grad_(True), xp.clone().requires_grad_(True)
    fm_o, p_o = orc.mobilevit_block_temporal(leaf, "", xo, xpo, blocks, d // hd, True, None, patch, patch)
    g, gp = seeded_input(tuple(fm_o.shape), seed=43), seeded_input(tuple(p_o.shape), seed=44)
    want = torch.autograd.grad((fm_o * g).sum() + (p_o * gp).sum(), [xo, xpo] + [leaf[k] for k in names])
    # HIP path
    xg, xpg = x.to(DEV).requires_grad_(True), xp.to(DEV).requires_grad_(True)
    cvnets_amd.set_compute_dtype(dtype)
    try:
        fm, p = block((xg, xpg))
        loss = (fm.float() * g.to(DEV)).sum() + (p.float() * gp.to(DEV)).sum()
        got = torch.autograd.grad(loss, [xg, xpg] + [dict(block.named_parameters())[k] for k in names])
        torch.cuda.synchronize()
    finally:
        cvnets_amd.set_compute_dtype(None)
    out_tol, grad_tol = (OUT_TOL, GRAD_TOL) if dtype == torch.float32 else (2e-2, 5e-2)
    assert tuple(p.shape) == tuple(p_o.shape) == (b * patch * patch, (H // patch) * (W // patch), d)
    assert _rel(fm.detach().float().cpu(), fm_o.detach()) < out_tol and _rel(p.detach().float().cpu(), p_o.detach()) < out_tol
    num = sum(float((a.float().cpu().double() - w.double()).pow(2).sum()) for a, w in zip(got, want))
    den = sum(float(w.double().pow(2).sum()) for w in want)
    assert (num / den) ** 0.5 < grad_tol, (num / den) ** 0.5
    for nm, a, w in zip(["x", "x_prev"], got[:2], want[:2]):
        assert _rel(a.float().cpu(), w) < (grad_tol if dtype == torch.float32 else 1e-1), nm
