"""The statistics of a block's incoming gradient handed BACKWARD by the kernel that produces it (cvnets_amd/ops.py: offer_grad_stats /
take_grad_stats; csrc/ir_bwd.hip: cvh_ir_exp_bwd_s): for a chain InvertedResidual(no residual) -> InvertedResidual
(cvnets/modules/mobilenetv2.py:231-235 twice) the BatchNorm backward of the first block's projection takes (sum dout, sum dout * xhat)
from the second block's expansion backward instead of a pass over (dout, y3).  Checked against the same chain with the hand-over off
and against the fp32 torch evaluation; a tensor modified between the blocks must NOT take the handed statistics."""
import pytest
import torch

from test_fused_ir_gpu import _build, _torch_reference
from util import l2_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run_chain(blocks, x, dout, dtype, modify=False):
    import cvnets_amd

    cvnets_amd.set_compute_dtype(dtype)
    try:
        for b in blocks:
            b.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        h = blocks[0](xg)
        if modify:
            h = cvnets_amd.ops.add(h, torch.zeros_like(h))  # another tensor between the blocks: nothing may be handed across it
        y = blocks[1](h)
        y.backward(dout.to(y.dtype))
        torch.cuda.synchronize()
        grads = {f"{i}.{n}": p.grad.detach().float().clone() for i, b in enumerate(blocks) for n, p in b.named_parameters()}
        return y.detach().float(), xg.grad.detach().float(), grads
    finally:
        cvnets_amd.set_compute_dtype(None)


# (cin0, cout0, stride0, cin1, cout1, stride1, H, B): MobileViT-S layer_1 -> layer_2, layer_2 block 1 -> block 2; B so that the second block's
# input has >= 65536 rows (the fused expansion backward cvh_ir_exp_bwd runs from there)
@pytest.mark.parametrize("cfg", [(16, 32, 1, 32, 64, 2, 64, 36), (32, 64, 2, 64, 64, 1, 64, 72)])
def test_handed_statistics_match_the_separate_pass(cfg, monkeypatch):
    from cvnets_amd import ops

    cin0, cout0, s0, cin1, cout1, s1, H, B = cfg
    blocks = [_build(cin0, cout0, s0, 3).to(DEV).train(), _build(cin1, cout1, s1, 4).to(DEV).train()]
    assert not blocks[0].use_res_connect
    torch.manual_seed(0)
    x = torch.randn(B, cin0, H, H, device=DEV)
    Ho = H // s0 // s1
    dout = torch.randn(B, cout1, Ho, Ho, device=DEV)
    monkeypatch.setattr(ops, "_BN_HANDOVER", False)
    _run_chain(blocks, x, dout, torch.bfloat16)  # (the forward-direction hand-over of the Gram matrix is learned during the first call)
    y0, dx0, g0 = _run_chain(blocks, x, dout, torch.bfloat16)
    monkeypatch.setattr(ops, "_BN_HANDOVER", True)
    taken = []
    orig = ops.take_grad_stats
    monkeypatch.setattr(ops, "take_grad_stats", lambda *a: (taken.append(orig(*a)), taken[-1])[1])
    y1, dx1, g1 = _run_chain(blocks, x, dout, torch.bfloat16)
    assert any(t is not None for t in taken), "the hand-over did not happen on this shape"
    assert torch.equal(y0, y1)
    # same quantities from two different reductions: bf16 storage noise only
    assert l2_err(dx1, dx0) < 2e-2
    for k in g0:
        assert l2_err(g1[k], g0[k]) < 3e-2, k
    # and both against fp32 torch on the first block's BatchNorm parameters (the ones the handed sums feed directly)
    h_ref, _, _, _ = _torch_reference(blocks[0], x, torch.zeros(B, cout0, H // s0, H // s0, device=DEV))
    _, dh_ref, gr1, _ = _torch_reference(blocks[1], h_ref, dout)
    _, _, gr0, _ = _torch_reference(blocks[0], x, dh_ref)
    n = "block.red_1x1.block.norm.weight"
    assert l2_err(g1["0." + n], gr0[n]) < 4e-2 and l2_err(g0["0." + n], gr0[n]) < 4e-2
    # (the bias gradient of this BatchNorm is the column sum of a gradient that a train-mode BatchNorm has just centred: zero in exact
    # arithmetic, the sum of the bf16 rounding errors of `rows` stored values in both evaluations — bounded as such)
    nb = "block.red_1x1.block.norm.bias"
    if blocks[1].use_res_connect:  # the second block's residual branch passes its own incoming gradient through: an ordinary sum
        assert l2_err(g1["0." + nb], gr0[nb]) < 4e-2 and l2_err(g0["0." + nb], gr0[nb]) < 4e-2
    else:
        rows = B * (H // s0) ** 2
        bound = 6.0 * (rows ** 0.5) * 2.0 ** -9 * float(dh_ref.abs().max())
        assert float(g1["0." + nb].abs().max()) < bound and float(gr0[nb].abs().max()) < bound
    # a tensor in between: the second block's input is not the first block's output -> no hand-over, same results as with it off
    taken.clear()
    y2, dx2, g2 = _run_chain(blocks, x, dout, torch.bfloat16, modify=True)
    assert all(t is None for t in taken)
    assert l2_err(dx2, dx0) < 2e-2
