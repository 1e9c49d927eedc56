"""stem kernels (csrc/stem.hip): the 3x3 stride-2 pad-1 convolution of the raw NCHW image batch (MobileViT conv_1,
cvnets/models/classification/mobilevit.py:62-72; ConvLayer2d.forward, cvnets/layers/conv_layer.py:254-255), forward and weight gradient,
against
  (a) torch.nn.functional.conv2d / its weight gradient in fp32 on the same bf16-rounded operands (forward: one bf16 rounding of the
      result; dW: fp32 accumulation of bf16 products, 2e-3 of the magnitude),
  (b) the generic path of the same layer (NCHW -> NHWC(8) repack + implicit GEMM; CVH_STEM_KERNEL off): module output, BatchNorm
      running statistics and every gradient of one training step.
Shapes: the bench geometry (256 x 256), ragged maps (H, W not multiples of the 8 x 64 output tile, odd H), 16 and 32 output channels, fp32 and
bf16 images."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # B, H, W, Cout, image dtype
    (4, 256, 256, 16, torch.float32), (3, 64, 64, 16, torch.float32), (2, 30, 44, 16, torch.float32), (5, 33, 20, 16, torch.bfloat16),
    (2, 130, 260, 32, torch.float32), (3, 18, 132, 32, torch.bfloat16), (1, 2, 4, 16, torch.float32), (2, 224, 224, 32, torch.float32),
]


@pytest.mark.parametrize("B,H,W,Cout,xdt", CASES)
def test_stem_forward_and_dw_match_torch(B, H, W, Cout, xdt):
    from cvnets_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(B + H + Cout)
    x = torch.randn(B, 3, H, W, device=DEV, generator=g).to(xdt)
    w = torch.randn(Cout, 3, 3, 3, device=DEV, generator=g) * 27 ** -0.5
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    R = _lib.query("cvh_stem_rows", B, H, W, Cout)
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((R, 2, Cout), float("nan"), device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("cvh_stem_conv_fwd", ops._dt(x), x.data_ptr(), ops._dt(w), w.data_ptr(), y.data_ptr(), part.data_ptr(), B, H, W, Cout, s)
    xr, wr = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(xr, wr, stride=2, padding=1).permute(0, 2, 3, 1)
    assert ref.shape == y.shape
    assert not torch.isnan(y.float()).any()
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 8e-3
    yf = y.float().double().reshape(-1, Cout)
    st = part.sum(0).double()
    assert torch.allclose(st[0], yf.sum(0), rtol=1e-4, atol=1e-3 * float(yf.abs().sum(0).max()))
    assert torch.allclose(st[1], (yf * yf).sum(0), rtol=1e-4)

    dy = torch.randn(B, Ho, Wo, Cout, device=DEV, generator=g).bfloat16()
    dpart = torch.full((R, Cout, 9, 8), float("nan"), device=DEV)
    _lib.call("cvh_stem_conv_dw", ops._dt(x), x.data_ptr(), dy.data_ptr(), dpart.data_ptr(), B, H, W, Cout, s)
    got = dpart.sum(0)                                           # [Cout][tap][8]
    assert float(got[:, :, 3:].abs().max()) == 0.0              # the padded channels stay zero
    got = got[:, :, :3].permute(0, 2, 1).reshape(Cout, 3, 3, 3)  # torch layout [Cout][c][kh][kw]
    wref = wr.clone().requires_grad_(True)
    F.conv2d(xr, wref, stride=2, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert float((got - wref.grad).abs().max() / wref.grad.abs().max()) < 2e-3


@pytest.mark.parametrize("Cout,H,W", [(16, 64, 64), (32, 40, 72)])
def test_stem_layer_matches_generic_path_through_a_training_step(Cout, H, W):
    """ConvLayer2d(3 -> Cout, 3x3, s2, BN, SiLU) fed the raw NCHW batch: the stem kernels against the repack + implicit-GEMM path."""
    from cvnets_amd import layers, ops
    opts = layers.default_opts()
    x = torch.randn(6, 3, H, W, device=DEV)
    go = None
    res = {}
    for stem in (True, False):
        torch.manual_seed(3)
        m = layers.ConvLayer2d(opts, 3, Cout, 3, stride=2, use_norm=True, use_act=True).to(DEV).train()
        for p_ in m.parameters():
            p_.grad = None
        ops._STEM_KERNEL = stem
        ops.set_compute_dtype(torch.bfloat16)
        try:
            out = m(x)
            if go is None:
                go = torch.randn_like(out.float()).to(out.dtype)
            out.backward(go)
        finally:
            ops._STEM_KERNEL = True
            ops.set_compute_dtype(None)
        torch.cuda.synchronize()
        ops.finish_backward()
        torch.cuda.synchronize()
        res[stem] = (out.float(), m.block.conv.weight.grad.clone(), m.block.norm.weight.grad.clone(), m.block.norm.bias.grad.clone(),
                     m.block.norm.running_mean.clone(), m.block.norm.running_var.clone())
    names = ["out", "dW", "dgamma", "dbeta", "running_mean", "running_var"]
    for n, a, b in zip(names, res[True], res[False]):
        scale = float(b.abs().max()) + 1e-6
        tol = 2e-2 if n in ("out",) else 3e-2
        assert float((a - b).abs().max()) / scale < tol, (n, float((a - b).abs().max()), scale)
