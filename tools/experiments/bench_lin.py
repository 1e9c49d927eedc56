"""isolated timings of cvh_conv_gemm on the transformer-linear shapes of MobileViT-S at batch 1024 (developer script)"""
import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib, ops
dev = "cuda:0"
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(1048576, 144, 432), (1048576, 144, 144), (1048576, 144, 288), (1048576, 288, 144), (262144, 192, 576), (262144, 384, 192), (65536, 240, 720), (65536, 480, 240),
          (1048576, 96, 144), (1048576, 144, 96), (4194304, 64, 256)]
for (M, K, N) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    wp = ops.pack_weight(w.view(N, K, 1, 1), torch.bfloat16, 0)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).bfloat16()
    byts = M * (K + N) * 2
    t0 = timeit(lambda: ops._conv_gemm(a, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N))
    t1 = timeit(lambda: ops._conv_gemm(a, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, bias=b, act=1))
    t2 = timeit(lambda: ops._conv_gemm(a, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, bias=b, residual=res))
    print(f"M={M} K={K} N={N}: plain {t0:7.1f} us {byts / t0 / 1e3:5.0f} GB/s | bias+silu {t1:7.1f} us {byts / t1 / 1e3:5.0f} GB/s | bias+res {t2:7.1f} us {(byts + M * N * 2) / t2 / 1e3:5.0f} GB/s")
    del a, out, res
