"""ir_exp_bwd_kernel (csrc/ir_bwd.hip): the input gradient and the raw weight-gradient product of an InvertedResidual expansion conv
(cvnets/modules/mobilenetv2.py:180-193; nn.Conv2d 1x1 backward behind a BatchNorm, cvnets/layers/conv_layer.py:254-255) from one pass
over the wide gradient, against fp32 matmuls on the same bf16 operands (dX: one bf16 rounding, 8e-3 of the magnitude; P: fp32
accumulation of bf16 products, 2e-3), and through InvertedResidualFn against the two-kernel path (CVH_IR_EXP_FUSED off).
Row counts are not multiples of the 64-row tile; with and without the residual gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("hid,Cin", [(64, 16), (128, 32), (256, 64)])
@pytest.mark.parametrize("M,res", [(65536 + 37, True), (70001, False), (262144, True)])
def test_ir_exp_bwd_matches_matmul(hid, Cin, M, res):
    from cvnets_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(hid + M)
    gt = torch.randn(M, hid, device=DEV, generator=g).bfloat16()
    x = torch.randn(M, Cin, device=DEV, generator=g).bfloat16()
    wcat = (torch.randn(Cin, hid + Cin, device=DEV, generator=g) * (hid + Cin) ** -0.5).bfloat16()
    bias = torch.randn(Cin, device=DEV, generator=g)
    r = torch.randn(M, Cin, device=DEV, generator=g).bfloat16() if res else None
    R = _lib.query("cvh_ir_exp_bwd_rows", M, hid, Cin)
    assert R > 0
    dx = torch.full((M, Cin), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((R, hid, Cin), float("nan"), device=DEV)
    _lib.call("cvh_ir_exp_bwd", 1, gt.data_ptr(), x.data_ptr(), wcat.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, dx.data_ptr(),
              part.data_ptr(), M, hid, Cin, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.cat([gt.float(), x.float()], 1) @ wcat.float().t() + bias
    if res:
        ref = ref + r.float()
    assert not torch.isnan(dx.float()).any()
    assert float((dx.float() - ref).abs().max() / ref.abs().max()) < 8e-3
    P = part.sum(0)
    Pref = (gt.double().t() @ x.double()).float()
    assert float((P - Pref).abs().max() / Pref.abs().max()) < 2e-3


def test_inverted_residual_step_with_and_without_the_fused_expansion_backward():
    from cvnets_amd import fused, layers, ops
    from cvnets_amd.modules import InvertedResidual
    opts = layers.default_opts()
    x = (torch.randn(20, 64, 64, 64, device=DEV)).bfloat16().contiguous(memory_format=torch.channels_last)
    go = None
    res = {}
    for on in (True, False):
        torch.manual_seed(5)
        m = InvertedResidual(opts, 64, 64, stride=1, expand_ratio=4).to(DEV).train()
        xin = x.clone().requires_grad_(True)
        fused._IR_EXP_FUSED = fused._IR_RED_FUSED = on
        ops.set_compute_dtype(torch.bfloat16)
        try:
            out = m(xin)
            if go is None:
                go = torch.randn_like(out.float()).to(out.dtype)
            out.backward(go)
            ops.finish_backward()
        finally:
            fused._IR_EXP_FUSED = fused._IR_RED_FUSED = True
            ops.set_compute_dtype(None)
        torch.cuda.synchronize()
        res[on] = [out.detach().float(), xin.grad.float()] + [p_.grad.float().clone() for p_ in m.parameters()]
    for a, b in zip(res[True], res[False]):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) / scale < 2e-2, (a.shape, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("hid,N", [(64, 32), (128, 64), (256, 64), (256, 96)])
@pytest.mark.parametrize("M", [65536 + 37, 140003])
def test_ir_red_fwd_matches_matmul(hid, N, M):
    """ir_red_fwd_kernel (csrc/ir_fwd.hip): projection GEMM with BatchNorm + SiLU applied on load, against the same formula in fp32 on the
    bf16-rounded activated operand (one bf16 rounding of the result: 8e-3 of the magnitude); statistics = sums over the stored values."""
    from cvnets_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(hid + N + M)
    y2 = torch.randn(M, hid, device=DEV, generator=g).bfloat16()
    scale = torch.rand(hid, device=DEV, generator=g) + 0.5
    shift = torch.randn(hid, device=DEV, generator=g) * 0.3
    w = (torch.randn(N, hid, device=DEV, generator=g) * hid ** -0.5).bfloat16()
    R = _lib.query("cvh_ir_red_fwd_rows", M, hid, N)
    assert R > 0
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((R, 2, N), float("nan"), device=DEV)
    _lib.call("cvh_ir_red_fwd", 1, y2.data_ptr(), scale.data_ptr(), shift.data_ptr(), ops.ACT_SILU, w.data_ptr(), out.data_ptr(), part.data_ptr(),
              M, hid, N, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a = torch.nn.functional.silu(y2.float() * scale + shift).bfloat16().float()
    ref = a @ w.float().t()
    assert not torch.isnan(out.float()).any()
    assert float((out.float() - ref).abs().max() / ref.abs().max()) < 8e-3
    of = out.float().double()
    st = part.sum(0).double()
    assert torch.allclose(st[0], of.sum(0), rtol=1e-4, atol=1e-3 * float(of.abs().sum(0).max()))
    assert torch.allclose(st[1], (of * of).sum(0), rtol=1e-4)
