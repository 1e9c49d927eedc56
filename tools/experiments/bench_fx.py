"""isolated timings of cvh_pw_gemm_bn variants on the dX3 shapes of the fused InvertedResidual backward (developer script)"""
import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib, ops
from cvnets_amd.fused import _pw_gemm, _xf
dev = "cuda:0"
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, K, N) in [(4194304, 64, 128), (4194304, 64, 256), (16777216, 32, 64), (1048576, 96, 256)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev) * 0.1
    wp = ops.pack_weight(w.view(N, K, 1, 1), torch.bfloat16, 0) if hasattr(ops, "pack_weight") else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(M, N, device=dev).bfloat16()
    st = torch.randn(4, N, device=dev).abs() + 0.5
    gb = lambda byts, us: f"{byts / us / 1e3:6.0f} GB/s"
    b0 = M * (K + N) * 2
    b1 = M * (K + 2 * N) * 2
    t = timeit(lambda: _pw_gemm(a, None, K, wp, out, M, N)); print(M, K, N, "plain            ", round(t), "us", gb(b0, t))
    t = timeit(lambda: _pw_gemm(a, None, K, wp, out, M, N, want_stats=True)); print(M, K, N, "plain+stats      ", round(t), "us", gb(b0, t))
    t = timeit(lambda: _pw_gemm(a, None, K, wp, out, M, N, residual=aux)); print(M, K, N, "plain+residual   ", round(t), "us", gb(b1, t))
    t = timeit(lambda: _pw_gemm(a, None, K, wp, out, M, N, e_mode=1, e_aux=aux, e_stats=st, e_act=1)); print(M, K, N, "e_mode1          ", round(t), "us", gb(b1, t))
    t = timeit(lambda: _pw_gemm(a, None, K, wp, out, M, N, e_mode=1, e_aux=aux, e_stats=st, e_act=1, want_stats=True)); print(M, K, N, "e_mode1+stats    ", round(t), "us", gb(b1, t))
    t = timeit(lambda: _pw_gemm(a, None, K, wp, out, M, N, e_mode=1, e_aux=aux, e_stats=st, e_act=0, want_stats=True)); print(M, K, N, "e_mode1+stats act0", round(t), "us", gb(b1, t))
    del a, out, aux
