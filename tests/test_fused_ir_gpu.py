"""GPU parity tests of the BatchNorm-link kernels (csrc/bnlink.hpp, gemm_fx.hip, dwfused.hip) and of the block-level
InvertedResidual function built on them (cvnets_amd/fused.py), against
  (a) a plain PyTorch fp32 evaluation of cvnets/modules/mobilenetv2.py:231-235 (conv -> BatchNorm2d(train) -> SiLU ...) with autograd,
  (b) the per-layer HIP path of ops.py (which is itself pinned to the reference fixtures).
Forward output, input gradient, every parameter gradient and the BatchNorm running statistics are compared.
Tolerances: fp32 2e-4 of the reference's max magnitude (gradients 1e-3), bf16 3e-2 (train-mode BatchNorm amplifies bf16 rounding)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from util import l2_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cin, cout, stride, seed):
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import InvertedResidual

    torch.manual_seed(seed)
    m = InvertedResidual(default_opts(), cin, cout, stride=stride, expand_ratio=4)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / fan ** 0.5))
            elif n.endswith("weight"):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for n, b in m.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
    return m


def _torch_reference(m, x, dout, training=True):
    """fp32 PyTorch evaluation of the same block from its state_dict (ATen ops + autograd, on the GPU for speed)."""
    sd = {k: v.detach().clone().float().to(DEV) for k, v in m.state_dict().items()}
    ps = {k: v.requires_grad_(True) for k, v in sd.items() if "running" not in k and "num_batches" not in k}
    x = x.detach().clone().float().requires_grad_(True)
    stride = m.stride

    def bn(t, pre):
        return F.batch_norm(t, sd[pre + ".running_mean"], sd[pre + ".running_var"], ps[pre + ".weight"], ps[pre + ".bias"], training, 0.1, 1e-5)

    y = F.silu(bn(F.conv2d(x, ps["block.exp_1x1.block.conv.weight"]), "block.exp_1x1.block.norm"))
    y = F.silu(bn(F.conv2d(y, ps["block.conv_3x3.block.conv.weight"], stride=stride, padding=1, groups=y.shape[1]), "block.conv_3x3.block.norm"))
    y = bn(F.conv2d(y, ps["block.red_1x1.block.conv.weight"]), "block.red_1x1.block.norm")
    if m.use_res_connect:
        y = y + x
    y.backward(dout.float())
    grads = {k: v.grad for k, v in ps.items()}
    return y.detach(), x.grad, grads, sd


def _run_hip(m, x, dout, dtype, fused, training=True):
    import cvnets_amd
    from cvnets_amd import modules, ops

    cvnets_amd.set_compute_dtype(dtype)
    modules.set_fused_inverted_residual(fused)
    try:
        m = m.to(DEV)
        m.train(training)
        m.zero_grad(set_to_none=True)
        xin = ops.to_nhwc(x.to(DEV), dtype).detach().requires_grad_(True)
        y = m(xin)
        y.backward(dout.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        bufs = {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k}
        return y.detach().float(), xin.grad.detach().float(), grads, bufs
    finally:
        cvnets_amd.set_compute_dtype(None)
        modules.set_fused_inverted_residual(True)


CASES = [
    # cin, cout, stride, B, H, W
    (16, 32, 1, 2, 32, 32),     # layer_1 shape family (no residual)
    (64, 64, 1, 3, 24, 40),     # residual, rectangular, W not a multiple of the 16-wide tile
    (32, 64, 2, 2, 32, 32),     # stride 2
    (24, 24, 1, 2, 9, 21),      # hidden = 96: partial 64-channel chunk; ragged tiles
    (64, 96, 2, 2, 17, 23),     # stride 2 on odd sizes
    (128, 160, 2, 4, 16, 16),   # hidden 512 (8 chunks), N = 160 (5-fragment GEMM)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_fused_inverted_residual_matches_torch_and_unfused(case, dtype):
    import copy

    cin, cout, stride, B, H, W = case
    m0 = _build(cin, cout, stride, seed=cin + stride)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dout = torch.randn(B, cout, Ho, Wo, generator=g).to(DEV) * 0.5
    ref_y, ref_dx, ref_g, ref_sd = _torch_reference(copy.deepcopy(m0), x, dout)
    out = {}
    for fused in (True, False):
        out[fused] = _run_hip(copy.deepcopy(m0), x, dout, dtype, fused)
    tol = {torch.float32: (2e-4, 1e-3), torch.bfloat16: (3e-2, 6e-2)}[dtype]
    for fused in (True, False):
        y, dx, grads, bufs = out[fused]
        tag = f"{'fused' if fused else 'unfused'} {dtype} {case}"
        assert rel_err(y, ref_y) < tol[0], (tag, "y", rel_err(y, ref_y))
        assert rel_err(dx[:, :cin], ref_dx) < tol[1], (tag, "dx", rel_err(dx[:, :cin], ref_dx))
        for k, gr in grads.items():
            e = l2_err(gr, ref_g[k])
            assert e < tol[1] * 2, (tag, k, e)
        for k, b in bufs.items():
            assert rel_err(b, ref_sd[k]) < tol[0], (tag, k, rel_err(b, ref_sd[k]))
    if dtype == torch.float32:  # the two HIP paths agree much more tightly than either does with ATen
        assert rel_err(out[True][0], out[False][0]) < 2e-5
        assert rel_err(out[True][1], out[False][1]) < 2e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_inverted_residual_eval_mode(dtype):
    """eval-mode BatchNorm (running statistics): forward only statistics-free links, backward with frozen coefficients."""
    import copy

    m0 = _build(32, 32, 1, seed=3)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 32, 20, 20, generator=g).to(DEV)
    dout = torch.randn(2, 32, 20, 20, generator=g).to(DEV)
    ref_y, ref_dx, ref_g, _ = _torch_reference(copy.deepcopy(m0), x, dout, training=False)
    y, dx, grads, bufs = _run_hip(copy.deepcopy(m0), x, dout, dtype, True, training=False)
    tol = {torch.float32: (2e-4, 1e-3), torch.bfloat16: (3e-2, 6e-2)}[dtype]
    assert rel_err(y, ref_y) < tol[0]
    assert rel_err(dx, ref_dx) < tol[1]
    for k, gr in grads.items():
        assert l2_err(gr, ref_g[k]) < tol[1] * 2, k
    for k, b in bufs.items():  # untouched
        assert torch.equal(b.cpu(), m0.state_dict()[k].cpu()), k


def test_fused_inverted_residual_inplace_grads_and_repeat():
    """in-place parameter-gradient accumulation (bench path) and back-to-back steps: the link accumulators must come back clean."""
    import copy

    import cvnets_amd
    from cvnets_amd import ops

    m = _build(32, 32, 1, seed=5).to(DEV).train()
    ref = copy.deepcopy(m)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(4, 32, 16, 16, generator=g).to(DEV)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        xin = ops.to_nhwc(x, torch.float32)
        grads = []
        for mod, inplace in ((m, True), (ref, False)):
            ops.set_inplace_param_grads(inplace)
            for p in mod.parameters():
                p.grad = torch.zeros_like(p) if inplace else None
            for _ in range(3):  # three accumulated steps
                mod(xin).float().square().mean().backward()
            torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in mod.named_parameters()})
        for k in grads[0]:
            assert l2_err(grads[0][k], grads[1][k]) < 1e-5, k
        for k, v in m.state_dict().items():
            if "running" in k:
                assert rel_err(v, ref.state_dict()[k]) < 1e-6, k
    finally:
        ops.set_inplace_param_grads(False)
        cvnets_amd.set_compute_dtype(None)
