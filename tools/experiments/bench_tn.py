"""isolated timings of cvh_gemm_dw on the transformer / 3x3 dW shapes of MobileViT-S at batch 1024 (developer script)"""
import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib
dev = "cuda:0"
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(1048576, 432, 144, 1), (1048576, 144, 144, 1), (1048576, 288, 144, 1), (1048576, 144, 288, 1), (262144, 576, 192, 1), (262144, 192, 384, 1),
          (1048576, 96, 96, 3), (1048576, 96, 192, 3), (4194304, 256, 64, 1), (4194304, 64, 256, 1)]
for (M, N, C, k) in shapes:
    Hh = int(round((M // 1024) ** 0.5))
    B = 1024
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, C, device=dev).bfloat16()
    Kt = C * k * k
    n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, Kt)
    scr = torch.empty(max(n_scr, 1), device=dev)
    dw = torch.empty(N * Kt, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.call("cvh_gemm_dw", 1, dy.data_ptr(), x.data_ptr(), None, C, 0, dw.data_ptr(), B, Hh, Hh, Hh, Hh, k, k, 1, k // 2, 1, N, C, scr.data_ptr(), n_scr, 0, st)
    t = timeit(f)
    byts = M * (N + C) * 2
    fl = 2.0 * M * N * Kt
    print(f"M={M} N={N} C={C} k={k}: {t:8.1f} us  {byts / t / 1e3:6.0f} GB/s  {fl / t / 1e6:6.1f} TFLOP/s  rows={n_scr // (N * Kt)}")
    del dy, x, scr, dw
