"""Weight-gradient GEMM dW = dY^T X through the C ABI (cvh_gemm_dw_bias) on the shapes of the early 1x1 convolutions — millions of rows
under one small output tile (gemm_tn_skinny_kernel) — and the bias gradient that rides along in the same kernel, against a float64
torch reference.  Replaces the weight / bias halves of Conv2d(1x1).backward and Linear.backward (cvnets/layers/conv_layer.py:254,
cvnets/layers/linear_layer.py:69-75 via autograd)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dtype, M, N, K, C2=0, bias=True, accumulate=False):
    from cvnets_amd import _lib

    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    dy = torch.randn(M, N, generator=g).to(dev).to(dtype)
    x = torch.randn(M, K, generator=g).to(dev).to(dtype)
    x2 = torch.randn(M, C2, generator=g).to(dev).to(dtype) if C2 else None
    Ktot = K + C2
    n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, Ktot)
    rows = n_scr // (N * Ktot)
    scr = torch.full((max(n_scr, 1),), float("nan"), device=dev)
    dw = torch.full((N, Ktot), 0.5 if accumulate else float("nan"), device=dev)
    bpart = torch.full((rows, N), float("nan"), device=dev) if bias else None
    dt = 1 if dtype == torch.bfloat16 else 0
    if bias and not _lib.query("cvh_gemm_dw_folds_bias", dt, M, N, Ktot):
        bias, bpart = False, None  # shapes of the direct-to-LDS kernel: the operands never pass through registers, the caller uses cvh_colsum
    _lib.call("cvh_gemm_dw_bias", dt, dy.data_ptr(), x.data_ptr(), None if x2 is None else x2.data_ptr(), K, C2, dw.data_ptr(),
              None if bpart is None else bpart.data_ptr(), M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, Ktot, scr.data_ptr(), n_scr, 1 if accumulate else 0,
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xx = x if x2 is None else torch.cat([x, x2], 1)
    ref = dy.double().t() @ xx.double() + (0.5 if accumulate else 0.0)
    scale = float(ref.abs().max())
    assert torch.isfinite(dw).all()
    assert float((dw.double() - ref).abs().max()) < 2e-5 * scale + 1e-3 * (M ** 0.5) * 1e-3, (M, N, K)
    if bias:
        db = bpart.double().sum(0)
        rb = dy.double().sum(0)
        assert torch.isfinite(bpart).all()
        assert float((db - rb).abs().max()) < 1e-5 * float(rb.abs().max()) + 1e-3
    return rows


@pytest.mark.parametrize("N,K", [(64, 16), (128, 32), (16, 16), (32, 32), (64, 64), (96, 16), (32, 64), (24, 8), (128, 8), (72, 24)])
@pytest.mark.parametrize("M", [32768, 100003, 262144 + 96])
def test_skinny_pointwise_dw_and_bias(N, K, M):
    rows = _run(torch.bfloat16, M, N, K)
    assert rows % 4 == 0  # four partial rows (one per wave) per workgroup


@pytest.mark.parametrize("N,K", [(64, 16), (64, 64)])
def test_skinny_without_bias_and_accumulate(N, K):
    _run(torch.bfloat16, 65536 + 40, N, K, bias=False, accumulate=True)


@pytest.mark.parametrize("dtype,C2", [(torch.float32, 0), (torch.bfloat16, 16)])
def test_skinny_shapes_on_the_general_kernel(dtype, C2):
    """fp32 and two-source launches of a skinny SHAPE run gemm_tn_kernel with the same partial-row count"""
    _run(dtype, 40000, 64, 16, C2=C2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4096, 144, 96), (20000, 288, 144), (1000, 1000, 640), (300, 8, 8), (70000, 256, 64)])
def test_general_kernel_bias_fold(dtype, M, N, K):
    _run(dtype, M, N, K)


def test_linear_backward_bias_rides_along():
    """ops.linear backward inside autograd with in-place parameter gradients: bias.grad comes from the dW kernel's partial rows"""
    import cvnets_amd
    from cvnets_amd import ops

    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.randn(50000, 64, device=dev).bfloat16().requires_grad_(True)
    w = torch.nn.Parameter(torch.randn(96, 64, device=dev) * 0.1)
    b = torch.nn.Parameter(torch.randn(96, device=dev) * 0.1)
    go = torch.randn(50000, 96, device=dev).bfloat16()
    w.grad = torch.zeros_like(w)
    b.grad = torch.zeros_like(b)
    ops.set_inplace_param_grads(True)
    try:
        y = ops.linear(x, w, b)
        y.backward(go)
        torch.cuda.synchronize()
    finally:
        ops.set_inplace_param_grads(False)
    rb = go.double().sum(0)
    rw = go.double().t() @ x.detach().double()
    assert float((b.grad.double() - rb).abs().max()) < 1e-4 * float(rb.abs().max()) + 1e-2
    assert float((w.grad.double() - rw).abs().max()) < 1e-4 * float(rw.abs().max()) + 1e-2
