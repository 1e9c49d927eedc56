"""conv3x3_kernel (csrc/conv3x3.hip): the LDS-resident-halo 3x3 convolution of the MobileViT blocks (local_rep.conv_3x3 and the fusion conv over
cat(res, fm): cvnets/modules/mobilevit_block.py:102-148,269-288; nn.Conv2d via cvnets/layers/conv_layer.py:254-255) against
  (a) torch.nn.functional.conv2d in fp32 on the same bf16-rounded operands (one bf16 rounding of the result: 8e-3 of the magnitude),
  (b) the im2col conv_gemm_kernel on the same call (CVH_TUNE key 14 switches the new kernel off): outputs within one bf16 ulp, the
      BatchNorm column statistics (sum, sumsq of the stored values) equal.
Shapes: layer_3 / layer_4 geometry, two-source (virtual concat) inputs, ragged maps (H, W not multiples of the 8 x 16 tile), the dX
weight pack with a row offset (second source of the fusion conv)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEY = 14

CASES = [
    # B, H, W, C1, C2, N
    (16, 32, 32, 96, 0, 96), (16, 32, 32, 96, 96, 96), (64, 16, 16, 128, 0, 128), (64, 16, 16, 128, 128, 128),
    (40, 20, 24, 96, 0, 96), (36, 24, 40, 64, 32, 96), (20, 33, 31, 128, 96, 128), (16, 32, 32, 40, 0, 96),
    (16, 32, 32, 192, 0, 96), (64, 16, 16, 256, 0, 128), (64, 16, 16, 160, 160, 128), (16, 32, 32, 72, 200, 96),
]


def _run(ops, x, x2, w, N, stats):
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    wp = ops.pack_weight(w, torch.bfloat16, 0)
    y = torch.empty(B, H, W, N, device=DEV, dtype=torch.bfloat16)
    part = R = None
    if stats:
        from cvnets_amd import _lib
        R = _lib.query("cvh_conv_gemm_grid_rows", B * H * W, N)
        part = torch.full((R, 2, N), float("nan"), device=DEV)
    ops._conv_gemm(x, x2, C1, C2, wp, y, B, H, W, H, W, 3, 3, 1, 1, 1, N, stats_part=part)
    torch.cuda.synchronize()
    return y, (part.sum(0) if stats else None)


@pytest.mark.parametrize("B,H,W,C1,C2,N", CASES)
def test_conv3x3_matches_torch_and_im2col_kernel(B, H, W, C1, C2, N):
    from cvnets_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(B + H + C1)
    x = torch.randn(B, H, W, C1, device=DEV, generator=g).bfloat16()
    x2 = torch.randn(B, H, W, C2, device=DEV, generator=g).bfloat16() if C2 else None
    w = torch.randn(N, C1 + C2, 3, 3, device=DEV, generator=g) * (9 * (C1 + C2)) ** -0.5
    xin = x if x2 is None else torch.cat([x, x2], dim=3)
    ref = F.conv2d(xin.float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=1).permute(0, 2, 3, 1)
    for stats in (False, True):
        y, st = _run(ops, x, x2, w, N, stats)
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        assert err < 8e-3, (stats, err)
        _lib.call("cvh_set_tuning", KEY, 1)
        try:
            y_old, st_old = _run(ops, x, x2, w, N, stats)
        finally:
            _lib.call("cvh_set_tuning", KEY, 0)
        assert float((y.float() - y_old.float()).abs().max()) <= 2 ** -7 * float(ref.abs().max())
        if stats:
            yf = y.float().double().reshape(-1, N)
            assert torch.allclose(st[0].double(), yf.sum(0), rtol=1e-4, atol=1e-3 * float(yf.abs().sum(0).max()))
            assert torch.allclose(st[1].double(), (yf * yf).sum(0), rtol=1e-4)
            assert torch.allclose(st, st_old, rtol=2e-3, atol=2e-3 * float(st_old.abs().max()))


def test_conv3x3_input_gradient_of_two_source_conv():
    """dX of the fusion conv: two launches on the transposed / flipped pack, the second with a row offset (ops.ConvBNAct.backward)"""
    from cvnets_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    B, H, W, C1, C2, N = 16, 32, 32, 96, 96, 96
    dy = torch.randn(B, H, W, N, device=DEV, generator=g).bfloat16()
    w = torch.randn(N, C1 + C2, 3, 3, device=DEV, generator=g) * (9 * N) ** -0.5
    wpt = ops.pack_weight(w, torch.bfloat16, 1)   # [Cin][9 flipped][Cout]
    dx1 = torch.empty(B, H, W, C1, device=DEV, dtype=torch.bfloat16)
    dx2 = torch.empty(B, H, W, C2, device=DEV, dtype=torch.bfloat16)
    ops._conv_gemm(dy, None, N, 0, wpt, dx1, B, H, W, H, W, 3, 3, 1, 1, 1, C1)
    ops._conv_gemm(dy, None, N, 0, wpt, dx2, B, H, W, H, W, 3, 3, 1, 1, 1, C2, wp_offset=C1 * 9 * N)
    ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=1).permute(0, 2, 3, 1)
    got = torch.cat([dx1, dx2], dim=3).float()
    assert float((got - ref).abs().max() / ref.abs().max()) < 8e-3


DW_CASES = [
    # B, H, W, C1, C2, N
    (16, 32, 32, 96, 0, 96), (16, 32, 32, 96, 96, 96), (64, 16, 16, 128, 0, 128), (64, 16, 16, 128, 128, 128), (256, 8, 8, 160, 160, 160),
    (40, 20, 24, 96, 0, 96), (20, 33, 31, 128, 96, 128), (16, 32, 32, 48, 0, 96), (16, 32, 32, 64, 208, 96), (300, 8, 8, 160, 0, 160),
]


@pytest.mark.parametrize("B,H,W,C1,C2,N", DW_CASES)
def test_conv3x3_weight_gradient_matches_torch_and_im2col_kernel(B, H, W, C1, C2, N):
    """conv3x3_dw_kernel (csrc/conv3x3_dw.hip) through cvh_gemm_dw: against torch's conv weight gradient in fp32 on the same bf16 operands
    (fp32 accumulation of bf16 products: 2e-3 of the magnitude) and against the im2col dW kernel (CVH_TUNE key 15 switches the new one off)"""
    from cvnets_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(B + H + C1 + N)
    x = torch.randn(B, H, W, C1, device=DEV, generator=g).bfloat16()
    x2 = torch.randn(B, H, W, C2, device=DEV, generator=g).bfloat16() if C2 else None
    dy = torch.randn(B, H, W, N, device=DEV, generator=g).bfloat16()
    Cin = C1 + C2
    s = torch.cuda.current_stream().cuda_stream

    def run():
        n_scr = _lib.query("cvh_gemm_dw_scratch_elems_conv", 1, B, H, W, H, W, C1, C2, 3, 3, 1, 1, 1, N, 0)
        scr = torch.full((max(n_scr, 1),), float("nan"), device=DEV)
        dw = torch.full((N, Cin, 3, 3), float("nan"), device=DEV)
        _lib.call("cvh_gemm_dw", 1, dy.data_ptr(), x.data_ptr(), x2.data_ptr() if C2 else None, C1, C2, dw.data_ptr(), B, H, W, H, W, 3, 3, 1, 1, 1,
                  N, Cin, scr.data_ptr(), n_scr, 0, s)
        torch.cuda.synchronize()
        return dw, n_scr

    dw, n_new = run()
    _lib.call("cvh_set_tuning", 15, 1)
    try:
        dw_old, n_old = run()
    finally:
        _lib.call("cvh_set_tuning", 15, 0)
    assert n_new % (N * 9 * Cin) == 0 and n_old % (N * 9 * Cin) == 0
    xin = x if x2 is None else torch.cat([x, x2], dim=3)
    w = torch.zeros(N, Cin, 3, 3, device=DEV, requires_grad=True)
    F.conv2d(xin.float().permute(0, 3, 1, 2), w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    scale = float(w.grad.abs().max())
    assert not torch.isnan(dw).any()
    assert float((dw - w.grad).abs().max()) / scale < 2e-3
    assert float((dw - dw_old).abs().max()) / scale < 2e-3
