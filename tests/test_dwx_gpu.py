"""dwx kernels (csrc/dwx.hip): expansion 1x1 conv + BatchNorm + SiLU + depthwise 3x3 conv of an InvertedResidual block
(cvnets/modules/mobilenetv2.py:180-207,231-235) with the 4x-wide expansion output recomputed from the narrow block input instead of stored —
forward (y2 and its statistics), the Gram-matrix BatchNorm statistics of the expansion, and backward (g1, depthwise dW, statistics) —
against the same formulas in fp32 torch ops on the same bf16-rounded operands and rounding points (the activated tensor and dy are
bf16 operands in the reference too; y1 stays fp32 on both sides).  Output tolerances: one bf16 rounding of the result (8e-3 of the magnitude); sums: 2e-3.
Ragged image sizes (tiles overhang the image), stride 1 and 2, every input width the kernels cover; then the whole fused
InvertedResidual against the y1-storing path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # (B, H, W, Cin, hid, stride)
    (2, 16, 32, 16, 64, 1),
    (1, 21, 27, 64, 256, 1),
    (3, 8, 16, 32, 128, 1),
    (2, 32, 32, 32, 128, 2),
    (1, 19, 23, 64, 256, 2),
    (2, 16, 16, 96, 384, 2),
    (2, 16, 16, 128, 512, 2),
    (1, 10, 12, 16, 72, 2),   # hid not a multiple of the 64-channel chunk
    # full strips (H % 8 == 0, W % 16 == 0, hid % 64 == 0, Cin <= 64): the strip-streaming forward kernel (csrc/dwxs.hip)
    (2, 24, 48, 16, 64, 1),    # three chunks per strip, left / interior / right strips
    (1, 64, 32, 64, 256, 1),   # eight chunks, four channel chunks
    (3, 8, 16, 64, 128, 1),    # single chunk: first and last rows in the same chunk
    (2, 40, 32, 32, 128, 2),   # stride 2, five chunks
    (1, 64, 64, 64, 256, 2),
    (1000, 16, 16, 16, 64, 2), # more units than workgroups: several strips per workgroup
    (300, 16, 32, 64, 64, 1),
    (1, 128, 16, 32, 64, 1),   # one strip: the row segments are shortened to spread the workgroups
]


def _inputs(B, H, W, Cin, hid, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=g).bfloat16()
    w1 = (torch.randn(hid, Cin, device=DEV, generator=g) * Cin ** -0.5).bfloat16()
    wd = (torch.randn(9, hid, device=DEV, generator=g) * 0.4).bfloat16()
    scale = torch.rand(hid, device=DEV, generator=g) + 0.5
    shift = torch.randn(hid, device=DEV, generator=g) * 0.3
    return g, x, w1, wd, scale, shift


def _ref_forward(x, w1, wd, scale, shift, stride):
    """y1 (fp32: it is never a tensor in these kernels, nothing rounds it) [B,H,W,hid], a1 = silu(bn(y1)) rounded, y2 fp32 [B,Ho,Wo,hid]"""
    y1 = x.float() @ w1.float().t()
    a1 = F.silu(y1 * scale + shift).bfloat16().float()
    hid = w1.shape[0]
    wt = wd.float().t().reshape(hid, 1, 3, 3)  # wd[kh*3+kw][c] -> [c][1][kh][kw]
    y2 = F.conv2d(a1.permute(0, 3, 1, 2), wt, stride=stride, padding=1, groups=hid).permute(0, 2, 3, 1)
    return y1, a1, y2.contiguous()


@pytest.mark.parametrize("B,H,W,Cin,hid,stride", CASES)
def test_dwx_fwd_matches_torch(B, H, W, Cin, hid, stride):
    from cvnets_amd import _lib
    _, x, w1, wd, scale, shift = _inputs(B, H, W, Cin, hid, 11 + H + hid)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    _, _, ref = _ref_forward(x, w1, wd, scale, shift, stride)
    R = _lib.query("cvh_dwx_fwd_rows", B, H, W, Cin, hid, stride)
    y2 = torch.full((B, Ho, Wo, hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((R, 2, hid), float("nan"), device=DEV)
    _lib.call("cvh_dwx_fwd", 1, x.data_ptr(), w1.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, wd.data_ptr(), y2.data_ptr(), part.data_ptr(),
              B, H, W, Ho, Wo, Cin, hid, stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not torch.isnan(y2.float()).any()
    err = float((y2.float() - ref).abs().max() / ref.abs().max())
    assert err < 8e-3, err
    st = part.sum(0)
    yf = y2.float().reshape(-1, hid)
    assert float((st[0] - yf.sum(0)).abs().max() / (yf.abs().sum(0).max() + 1e-6)) < 2e-3
    assert float((st[1] - (yf * yf).sum(0)).abs().max() / (yf * yf).sum(0).max()) < 2e-3


@pytest.mark.parametrize("M,K,hid", [(4099, 16, 64), (3000, 64, 256), (2048, 96, 384), (1500, 128, 512)])
def test_gram_bn_stats(M, K, hid):
    """(sum y1, sum y1^2) from G = x^T x and s = 1^T x equal the sums over the explicit product"""
    from cvnets_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(M + K)
    x = (torch.randn(M, K, device=DEV, generator=g) + 0.3).bfloat16()
    w1 = (torch.randn(hid, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    G = (x.double().t() @ x.double()).float().contiguous()
    s = x.double().sum(0).float()
    part = torch.full((2, hid), float("nan"), device=DEV)
    _lib.call("cvh_gram_bn_stats", G.data_ptr(), s.data_ptr(), w1.data_ptr(), part.data_ptr(), hid, K, K, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = x.double() @ w1.double().t()
    assert float((part[0].double() - y.sum(0)).abs().max() / y.abs().sum(0).max()) < 1e-5
    assert float((part[1].double() - (y * y).sum(0)).abs().max() / (y * y).sum(0).max()) < 1e-5


@pytest.mark.parametrize("two_src", [True, False])
@pytest.mark.parametrize("B,H,W,Cin,hid,stride", CASES)
def test_dwx_bwd_matches_autograd(B, H, W, Cin, hid, stride, two_src):
    from cvnets_amd import _lib
    g, x, w1, wd, scale, shift = _inputs(B, H, W, Cin, hid, 23 + W + hid)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    mean = torch.randn(hid, device=DEV, generator=g) * 0.2
    invstd = torch.rand(hid, device=DEV, generator=g) + 0.7
    in_stats = torch.stack([mean, invstd, scale, shift]).contiguous()
    g2 = torch.randn(B, Ho, Wo, hid, device=DEV, generator=g).bfloat16()
    y2 = torch.randn(B, Ho, Wo, hid, device=DEV, generator=g).bfloat16()
    ca = torch.rand(hid, device=DEV, generator=g) + 0.5
    cb = torch.randn(hid, device=DEV, generator=g) * 0.2
    cc = torch.randn(hid, device=DEV, generator=g) * 0.1
    dy = (ca * g2.float() + cb * y2.float() + cc).bfloat16().float() if two_src else g2.float()
    # reference: autograd through the depthwise conv and the activation on the rounded tensors
    y1 = (x.float() @ w1.float().t()).requires_grad_(True)
    a1 = F.silu(y1 * scale + shift)
    a1r = (a1.detach().bfloat16().float() - a1.detach() + a1)  # value rounded to bf16, gradient of the unrounded expression
    wt = wd.float().t().reshape(hid, 1, 3, 3).clone().requires_grad_(True)
    out = F.conv2d(a1r.permute(0, 3, 1, 2), wt, stride=stride, padding=1, groups=hid).permute(0, 2, 3, 1)
    out.backward(dy)
    g1_ref = y1.grad / scale  # d/dy1 = scale * act'(yh) * dz  ->  the kernel's g1 = dz * act'(yh)
    dw_ref = wt.grad.reshape(hid, 9)  # [c][kh*3+kw]
    R = _lib.query("cvh_dwx_rows", B, Ho, Wo, hid, stride)
    g1 = torch.full((B, H, W, hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((R, 2, hid), float("nan"), device=DEV)
    dwp = torch.full((R, hid * 9), float("nan"), device=DEV)
    _lib.call("cvh_dwx_bwd", 1, x.data_ptr(), w1.data_ptr(), in_stats.data_ptr(), 1, g2.data_ptr(), y2.data_ptr() if two_src else None,
              ca.data_ptr() if two_src else None, cb.data_ptr() if two_src else None, cc.data_ptr() if two_src else None, wd.data_ptr(),
              g1.data_ptr(), part.data_ptr(), dwp.data_ptr(), B, H, W, Ho, Wo, Cin, hid, stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not torch.isnan(g1.float()).any()
    err = float((g1.float() - g1_ref).abs().max() / g1_ref.abs().max())
    assert err < 8e-3, err
    dw = dwp.sum(0).reshape(hid, 9)
    err_w = float((dw - dw_ref).abs().max() / dw_ref.abs().max())
    assert err_w < 1e-2, err_w  # z enters the product as a bf16 operand (the reference multiplies the rounded a1 too)
    st = part.sum(0)
    gf = g1.float().reshape(-1, hid)
    xh = ((y1.detach() - mean) * invstd).reshape(-1, hid)
    assert float((st[0] - gf.sum(0)).abs().max() / (gf.abs().sum(0).max() + 1e-6)) < 2e-3
    assert float((st[1] - (gf * xh).sum(0)).abs().max() / ((gf * xh).abs().sum(0).max() + 1e-6)) < 2e-3


@pytest.mark.parametrize("M,hid,Cout,two_src", [(4096 + 77, 64, 32, False), (9000, 256, 64, True), (5000, 256, 96, False), (4200, 384, 128, True),
                                                 (4100, 512, 160, True), (20000, 128, 64, False), (70001, 256, 64, True)])
def test_ir_pb_projection_backward(M, hid, Cout, two_src):
    """cvh_ir_pb (csrc/ir_pb.hip): g2 = (dy3 W3) * act'(bn2(y2)), its BatchNorm-backward statistics and dW3 = dy3^T act(bn2(y2)) from one pass
    over y2, dy3 optionally formed on load from (dout, y3, coefficients) — against fp32 matmuls on the same bf16-rounded operands."""
    from cvnets_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(5 + hid + Cout)
    y2 = torch.randn(M, hid, device=DEV, generator=g).bfloat16()
    dout = (torch.randn(M, Cout, device=DEV, generator=g) * 0.5).bfloat16()
    y3 = torch.randn(M, Cout, device=DEV, generator=g).bfloat16()
    c3 = torch.stack([torch.rand(Cout, device=DEV, generator=g) + 0.5, torch.randn(Cout, device=DEV, generator=g) * 0.2,
                      torch.randn(Cout, device=DEV, generator=g) * 0.1]).contiguous()
    dy3 = (c3[0] * dout.float() + c3[1] * y3.float() + c3[2]).bfloat16().float() if two_src else dout.float()
    mean2 = torch.randn(hid, device=DEV, generator=g) * 0.2
    invstd2 = torch.rand(hid, device=DEV, generator=g) + 0.7
    sc2 = torch.rand(hid, device=DEV, generator=g) + 0.5
    sh2 = torch.randn(hid, device=DEV, generator=g) * 0.3
    st2 = torch.stack([mean2, invstd2, sc2, sh2]).contiguous()
    w3 = (torch.randn(Cout, hid, device=DEV, generator=g) * hid ** -0.5).bfloat16()
    yh = y2.float() * sc2 + sh2
    sg = torch.sigmoid(yh)
    z2, f1, xh = yh * sg, sg * (1 + yh * (1 - sg)), (y2.float() - mean2) * invstd2
    g2_ref = (dy3 @ w3.float()) * f1
    dw_ref = dy3.t() @ z2.bfloat16().float()  # z2 enters the product as a bf16 operand
    R = _lib.query("cvh_ir_pb_rows", M, hid, Cout)
    assert R > 0
    g2 = torch.full((M, hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((R, 2, hid), float("nan"), device=DEV)
    dwp = torch.full((R, Cout, hid), float("nan"), device=DEV)
    w3t = w3.t().contiguous()
    _lib.call("cvh_ir_pb", 1, dout.data_ptr(), y3.data_ptr() if two_src else None, c3.data_ptr() if two_src else None, y2.data_ptr(), st2.data_ptr(), 1,
              w3t.data_ptr(), g2.data_ptr(), part.data_ptr(), dwp.data_ptr(), M, hid, Cout, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not torch.isnan(g2.float()).any() and not torch.isnan(part).any() and not torch.isnan(dwp).any()
    assert float((g2.float() - g2_ref).abs().max() / g2_ref.abs().max()) < 8e-3
    dw = dwp.sum(0)
    assert float((dw - dw_ref).abs().max() / dw_ref.abs().max()) < 3e-3
    st = part.sum(0)
    assert float((st[0] - g2_ref.sum(0)).abs().max() / g2_ref.abs().sum(0).max()) < 2e-3
    assert float((st[1] - (g2_ref * xh).sum(0)).abs().max() / (g2_ref * xh).abs().sum(0).max()) < 2e-3


@pytest.mark.parametrize("Cin,Cout,stride,hw", [(64, 64, 1, 32), (32, 64, 2, 48), (16, 32, 1, 40), (96, 128, 2, 16)])
def test_inverted_residual_with_and_without_recomputed_expansion(Cin, Cout, stride, hw):
    """the whole fused block, forward + backward, with y1 recomputed (csrc/dwx.hip; "1": one-pass projection backward csrc/ir_pb.hip as
    well, "gemm": the projection backward as a dW GEMM + dX GEMM pair) against the y1-storing kernels"""
    from cvnets_amd import fused, layers, ops
    from cvnets_amd.modules import InvertedResidual
    opts = layers.default_opts()
    x = torch.randn(6, Cin, hw, hw + 8, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    go = None
    res = {}
    saved, saved_pb = fused._IR_X, fused._IR_PB
    for mode in ("1", "gemm", "fwd", "0"):
        torch.manual_seed(5)
        m = InvertedResidual(opts, Cin, Cout, stride=stride, expand_ratio=4).to(DEV).train()
        xin = x.clone().requires_grad_(True)
        fused._IR_X = "1" if mode == "gemm" else mode
        fused._IR_PB = mode == "1"
        ops.set_compute_dtype(torch.bfloat16)
        try:
            out = m(xin)
            if go is None:
                go = torch.randn_like(out.float()).to(out.dtype)
            out.backward(go)
            ops.finish_backward()
        finally:
            fused._IR_X, fused._IR_PB = saved, saved_pb
            ops.set_compute_dtype(None)
        torch.cuda.synchronize()
        res[mode] = [out.detach().float(), xin.grad.float()] + [p_.grad.float().clone() for p_ in m.parameters()] + \
                    [b_.float().clone() for b_ in m.buffers() if b_.dtype.is_floating_point]
    # the y1-storing path rounds y1 to bf16, the recomputing one keeps it in fp32: both are bf16-level results, and the per-channel BatchNorm
    # gradients of the expansion (sums over all pixels with heavy cancellation: the tensors with the largest bf16 error in the whole model,
    # tests/test_bf16_parity_gpu.py) move by a few per cent between them -> 1-D tensors get the looser bound
    for mode in ("1", "gemm", "fwd"):
        for a, b in zip(res[mode], res["0"]):
            scale = float(b.abs().max()) + 1e-6
            tol = 1e-1 if a.dim() == 1 else 2e-2
            assert float((a - b).abs().max()) / scale < tol, (mode, a.shape, float((a - b).abs().max()), scale)
