"""gemm_tn_rows_kernel (csrc/gemm_rows.hip): the weight gradient dW = dY^T X (and the bias gradient 1^T dY) of LinearLayer / 1x1 Conv2d backward
(cvnets/layers/linear_layer.py:74-91) for the MobileViT-sized token linears under >= 128 k rows, through the C ABI (cvh_gemm_dw_bias), against
float64 products of the same bf16 operands.  Shapes: every (N, K) of the MobileViT-S / XS / XXS transformer linears (one part, N parts, K parts),
row counts that are not multiples of the 32-row stage or of the split, with and without the bias sums; fp32 accumulation over 4 k rows per
split, double across splits: 2e-5 of the largest entry."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,N,K,bias,parts", [
    (262144, 144, 144, True, 0), (262144 + 37, 288, 144, True, 0), (300001, 144, 288, False, 0), (524288, 96, 96, True, 0),
    (1048576, 240, 240, True, 0), (1048576, 192, 192, True, 0),
    # column parts (N x K beyond one workgroup's accumulators): measured slower than the tiled kernel, off by default — CVH_TUNE key 23 = 1
    (400000, 432, 144, True, 1), (262144 + 5, 480, 240, True, 1), (524288, 240, 480, True, 1), (400000, 720, 240, False, 1),
])
def test_rows_dw_matches_float64(M, N, K, bias, parts):
    from cvnets_amd import _lib
    _lib.call("cvh_set_tuning", 23, parts)
    try:
        g = torch.Generator(device=DEV).manual_seed(M % 1000 + N + K)
        dy = (torch.randn(M, N, device=DEV, generator=g) * 0.5).bfloat16()
        x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, K)
        rows = n_scr // (N * K)
        # the whole-row plan: one partial row per workgroup along M (256 / parts of them), and the kernel sums the columns of dY itself
        assert rows in (256, 255, 128, 127, 85, 84) and _lib.query("cvh_gemm_dw_folds_bias", 1, M, N, K) == 1
        scr = torch.full((n_scr,), float("nan"), device=DEV)
        bpart = torch.full((rows, N), float("nan"), device=DEV) if bias else None
        dw = torch.full((N, K), float("nan"), device=DEV)
        _lib.call("cvh_gemm_dw_bias", 1, dy.data_ptr(), x.data_ptr(), None, K, 0, dw.data_ptr(), bpart.data_ptr() if bias else None, M, 1, 1, 1, 1, 1, 1, 1,
                  0, 1, N, K, scr.data_ptr(), n_scr, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        _lib.call("cvh_set_tuning", 23, 0)
    ref = torch.zeros(N, K, device=DEV, dtype=torch.float64)
    for a in range(0, M, 65536):  # chunks: the float64 copies of a million rows do not have to exist at once
        ref += dy[a:a + 65536].double().t() @ x[a:a + 65536].double()
    assert not torch.isnan(dw).any()
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    if bias:
        bref = dy.double().sum(0)
        assert float((bpart.double().sum(0) - bref).abs().max() / (dy.double().abs().sum(0).max())) < 2e-5


def test_plan_leaves_small_problems_to_the_tiled_kernels():
    from cvnets_amd import _lib
    # 131 k rows: below the gate; 262 k rows x (384, 192): the partial tiles would be a quarter of the operand stream; (432, 144): column parts
    for M, N, K in [(131072, 144, 144), (262144, 384, 192), (1048576, 432, 144)]:
        assert _lib.query("cvh_gemm_dw_scratch_elems", M, N, K) // (N * K) < 200
