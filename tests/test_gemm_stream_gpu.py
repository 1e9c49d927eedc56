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


@pytest.mark.parametrize("M,K,N", [(65536, 16, 64), (40000, 32, 128), (50000, 64, 256), (33000, 96, 384), (32768, 128, 512)])
def test_stream_gemm_statistics_epilogue(M, K, N):
    """cvh_pw_gemm_bn, plain operand + column statistics of the stored values (forward link of a BatchNorm behind the conv)"""
    from cvnets_amd import _lib, ops
    from cvnets_amd.fused import _pw_gemm
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    wp = ops.pack_weight(w.view(N, K, 1, 1), torch.bfloat16, 0)
    outs = []
    for off in (0, 1):
        _lib.call("cvh_set_tuning", KEY_NO_STREAM, off)
        try:
            y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            part, R = _pw_gemm(x, None, K, wp, y, M, N, want_stats=True)
            torch.cuda.synchronize()
            outs.append((y, part.view(R, 2, N).sum(0)))
        finally:
            _lib.call("cvh_set_tuning", KEY_NO_STREAM, 0)
    (y, st), (y_old, st_old) = outs
    ref = x.float() @ w.bfloat16().float().t()
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 8e-3
    yf = y.float().double()
    assert torch.allclose(st[0].double(), yf.sum(0), rtol=1e-4, atol=1e-3 * float(yf.abs().sum(0).max()))   # statistics OF THE STORED values
    assert torch.allclose(st[1].double(), (yf * yf).sum(0), rtol=1e-4)
    assert torch.allclose(st, st_old, rtol=2e-3, atol=2e-3 * float(st_old.abs().max()))


@pytest.mark.parametrize("M,K,N", [(65536, 32, 64), (40000, 64, 128), (50000, 64, 256)])
def test_stream_gemm_bn_backward_epilogue(M, K, N):
    """projection dX of the fused InvertedResidual: g = (dy W) * act'(scale * y + shift), statistics (sum g, sum g * xhat)"""
    from cvnets_amd import _lib, ops
    from cvnets_amd.fused import _pw_gemm
    g_ = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(M, K, device=DEV, generator=g_).bfloat16()
    w = torch.randn(N, K, device=DEV, generator=g_) * K ** -0.5
    wp = ops.pack_weight(w.view(N, K, 1, 1), torch.bfloat16, 0)
    yraw = (torch.randn(M, N, device=DEV, generator=g_) * 1.5 + 0.3).bfloat16()
    mean, var = yraw.float().mean(0), yraw.float().var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    gamma, beta = torch.rand(N, device=DEV, generator=g_) + 0.5, torch.randn(N, device=DEV, generator=g_) * 0.2
    stats = torch.stack([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
    for act in (1, 0):
        outs = []
        for off in (0, 1):
            _lib.call("cvh_set_tuning", KEY_NO_STREAM, off)
            try:
                out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
                part, R = _pw_gemm(dy, None, K, wp, out, M, N, e_mode=1, e_aux=yraw, e_stats=stats, e_act=act, want_stats=True)
                torch.cuda.synchronize()
                outs.append((out, part.view(R, 2, N).sum(0)))
            finally:
                _lib.call("cvh_set_tuning", KEY_NO_STREAM, 0)
        (o, st), (o_old, st_old) = outs
        z = (yraw.float() * stats[2] + stats[3]).requires_grad_(True)
        fz = torch.nn.functional.silu(z) if act == 1 else z
        (gz,) = torch.autograd.grad(fz.sum(), z)
        ref = (dy.float() @ w.bfloat16().float().t()) * gz
        assert float((o.float() - ref).abs().max() / ref.abs().max()) < 1e-2, act
        of = o.float().double()
        xhat = ((yraw.float() - mean) * invstd).double()
        assert torch.allclose(st[0].double(), of.sum(0), rtol=1e-3, atol=1e-3 * float(of.abs().sum(0).max()))
        assert torch.allclose(st[1].double(), (of * xhat).sum(0), rtol=1e-3, atol=1e-3 * float((of * xhat).abs().sum(0).max()))
        assert float((o.float() - o_old.float()).abs().max()) <= 2 ** -6 * float(ref.abs().max())
        assert torch.allclose(st, st_old, rtol=5e-3, atol=5e-3 * float(st_old.abs().max()))
