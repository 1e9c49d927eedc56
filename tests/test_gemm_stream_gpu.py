"""gemm_stream_kernel (csrc/gemm_stream.hip): the weights-resident, barrier-free GEMM of the MobileViT token linears / 1x1 convolutions
(LinearLayer.forward, cvnets/layers/linear_layer.py:74-91; ConvLayer2d 1x1, cvnets/layers/conv_layer.py:254-255) against
  (a) plain PyTorch fp32 on the same bf16-rounded operands, every epilogue option (bias, activation + saved pre-activation, activation
      gradient, residual, dropout) — tolerance: one bf16 rounding of the result (8e-3 of the magnitude);
  (b) conv_gemm_kernel / gemm_nt128_kernel on the same call (CVH_TUNE key 13 switches the new kernel off): identical dropout masks,
      outputs within one bf16 ulp.
Shapes: every (K, N) of the MobileViT-S / XS / XXS blocks, ragged M (not a multiple of the 16-row tile), K not a multiple of 32."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEY_NO_STREAM = 13


def _call(ops, x, w, N, **kw):
    M, K = x.shape
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops._conv_gemm(x, None, K, 0, ops.pack_weight(w.view(N, K, 1, 1), torch.bfloat16, 0), y, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, **kw)
    return y


SHAPES = [
    # M, K, N
    (40000, 144, 432), (40000, 144, 144), (40000, 144, 288), (40000, 96, 144), (40000, 144, 96),
    (33001, 192, 576), (33001, 192, 192), (33001, 192, 384), (33001, 128, 192), (33001, 192, 128),
    (32768, 240, 720), (32768, 240, 240), (32777, 160, 240), (32800, 80, 96), (32800, 72, 64), (32800, 256, 160),
    (40000, 288, 144), (33001, 320, 96), (33001, 200, 128), (32768, 480, 240),
]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_stream_gemm_plain_and_bias(M, K, N):
    from cvnets_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    b = torch.randn(N, device=DEV, generator=g)
    ref = x.float() @ w.bfloat16().float().t()
    for bias in (None, b):
        y = _call(ops, x, w, N, bias=bias)
        r = ref + (bias if bias is not None else 0.0)
        err = float((y.float() - r).abs().max() / r.abs().max())
        assert err < 8e-3, (M, K, N, bias is not None, err)
        _lib.call("cvh_set_tuning", KEY_NO_STREAM, 1)
        try:
            y_old = _call(ops, x, w, N, bias=bias)
        finally:
            _lib.call("cvh_set_tuning", KEY_NO_STREAM, 0)
        d = (y.float() - y_old.float()).abs()
        assert float(d.max()) <= 2 ** -7 * float(r.abs().max()), (M, K, N, float(d.max()))   # fp32 summation order only: within one bf16 ulp
        assert float((d > 0).float().mean()) < 0.05


@pytest.mark.parametrize("M,K,N", [(40000, 144, 288), (33001, 192, 384), (32768, 96, 144), (32768, 128, 64)])
def test_stream_gemm_fused_epilogues(M, K, N):
    import torch.nn.functional as F
    from cvnets_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    b = torch.randn(N, device=DEV, generator=g) * 0.3
    res = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    aux = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    pre_ref = x.float() @ w.bfloat16().float().t() + b
    # activation with the saved pre-activation (fc1 of the FFN: swish; ViT: gelu) + residual
    for act, fn in ((1, F.silu), (2, F.gelu), (3, F.relu)):
        pre = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        y = _call(ops, x, w, N, bias=b, act=act, save_pre=pre, residual=res)
        assert float((pre.float() - pre_ref).abs().max() / pre_ref.abs().max()) < 8e-3
        ref = fn(pre.float()) + res.float()        # the kernel activates the ROUNDED pre-activation it stores
        assert float((y.float() - ref).abs().max() / ref.abs().max()) < 8e-3, act
    # activation gradient (dX of fc2 flowing into fc1's activation): out = (x W^T) * act'(aux)
    for act in (1, 2, 3):
        y = _call(ops, x, w, N, actgrad_aux=aux, actgrad_act=act)
        a = aux.float().requires_grad_(True)
        fa = {1: F.silu, 2: F.gelu, 3: F.relu}[act](a)
        (ga,) = torch.autograd.grad(fa.sum(), a)
        ref = (x.float() @ w.bfloat16().float().t()) * ga
        assert float((y.float() - ref).abs().max() / ref.abs().max()) < 1e-2, act
    # dropout + residual (out-proj / fc2 epilogue): the mask is the one of the standalone kernel and of the other GEMM kernels
    seed = torch.tensor([0x1234_5678_9abc_def1], dtype=torch.int64, device=DEV)
    y = _call(ops, x, w, N, bias=b, drop_p=0.1, seed=seed, stream_id=5, residual=res)
    plain = _call(ops, x, w, N, bias=b)
    dropped = torch.empty_like(plain)
    _lib.call("cvh_dropout", 1, plain.data_ptr(), dropped.data_ptr(), plain.numel(), 0.1, seed.data_ptr(), 5, torch.cuda.current_stream().cuda_stream)
    ref = dropped.float() + res.float()
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 1.2e-2   # the standalone path rounds the GEMM result once more before the mask
    keep = float((dropped != 0).float().mean())
    assert abs(keep - 0.9) < 2e-3, keep
    _lib.call("cvh_set_tuning", KEY_NO_STREAM, 1)
    try:
        y_old = _call(ops, x, w, N, bias=b, drop_p=0.1, seed=seed, stream_id=5, residual=res)
    finally:
        _lib.call("cvh_set_tuning", KEY_NO_STREAM, 0)
    assert float(((y == res) != (y_old == res)).float().mean()) < 1e-4   # same elements dropped (dropped => out == residual)
