"""big-GEMM headroom check (developer script): gemm_nt128 / gemm_tn128 vs the library GEMM torch dispatches to, ViT-B shapes"""
import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib, ops
dev = "cuda:0"
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 512 * 197
for (K, N) in [(768, 2304), (768, 768), (768, 3072), (3072, 768)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02)
    wb = w.bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t = timeit(lambda: torch.matmul(a, wb.t(), out=out)); print(f"fwd  M={M} K={K} N={N}: torch {t:8.1f} us {fl / t / 1e6:7.1f} TF/s", end="  ")
    wp = ops.pack_weight(w.view(N, K, 1, 1), torch.bfloat16, 0)
    t = timeit(lambda: ops._conv_gemm(a, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N)); print(f"hip {t:8.1f} us {fl / t / 1e6:7.1f} TF/s")
    dy = torch.randn(M, N, device=dev).bfloat16()
    dwt = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: torch.matmul(dy.t(), a, out=dwt)); print(f"dW   M={M} K={K} N={N}: torch {t:8.1f} us {fl / t / 1e6:7.1f} TF/s", end="  ")
    n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, K)
    scr = torch.empty(max(n_scr, 1), device=dev); dw = torch.empty(N, K, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: _lib.call("cvh_gemm_dw", 1, dy.data_ptr(), a.data_ptr(), None, K, 0, dw.data_ptr(), M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, K, scr.data_ptr(), n_scr, 0, st))
    print(f"hip {t:8.1f} us {fl / t / 1e6:7.1f} TF/s")
    del a, out, dy, scr
