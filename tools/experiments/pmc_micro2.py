import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib, ops
dev, DT = "cuda:0", torch.bfloat16
def rnd(*s): return torch.randn(*s, device=dev).to(DT)
def lin(M, K, N, reps=2, **kw):
    x, w = rnd(M, K), torch.randn(N, K, device=dev) * 0.05
    wp = ops.pack_weight(w, DT, 0); y = torch.empty(M, N, device=dev, dtype=DT)
    for _ in range(reps): ops._conv_gemm(x, None, K, 0, wp, y, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, **kw)
    return x, wp, y
M = 1048576
if len(sys.argv) > 1 and sys.argv[1] == "time":
    for (K, N) in [(144, 128), (144, 144), (144, 192), (144, 256), (144, 288), (128, 128), (128, 256), (96, 96), (96, 192), (160, 160), (192, 192)]:
        x, wp, y = lin(M, K, N, reps=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops._conv_gemm(x, None, K, 0, wp, y, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f"K={K} N={N}: {us:.1f} us  {M * (K + N) * 2 / us / 1e6:.2f} TB/s")
else:
    lin(M, 144, 432); lin(M, 144, 144); lin(M, 144, 288); lin(M, 96, 256)
    torch.cuda.synchronize()
