"""a few representative launches of the step, isolated, for a rocprofv3 --pmc pass (developer aid): one launch each after one warm-up"""
import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib, ops
from cvnets_amd.fused import _pw_gemm
dev, DT = "cuda:0", torch.bfloat16
def rnd(*s): return torch.randn(*s, device=dev).to(DT)
def lin(M, K, N, **kw):
    x, w = rnd(M, K), torch.randn(N, K, device=dev) * 0.05
    wp = ops.pack_weight(w, DT, 0); y = torch.empty(M, N, device=dev, dtype=DT)
    for _ in range(2): ops._conv_gemm(x, None, K, 0, wp, y, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, **kw)
M = 1048576
lin(M, 144, 432)      # l3 qkv  (gemm_nt128)
lin(M, 144, 144)      # l3 out  (<5,64,0,0>)
lin(M, 144, 288)      # l3 fc1  (<3,64,0,0> x3)
lin(M, 288, 144)      # l3 fc2
lin(M, 96, 144)       # l3 1x1in
lin(4194304, 64, 256) # WP plain
# e_mode 1
M2, K2, N2 = 4194304, 64, 256
a = rnd(M2, K2); w = torch.randn(N2, K2, device=dev) * 0.1; wp = ops.pack_weight(w.view(N2, K2, 1, 1), DT, 0)
out = torch.empty(M2, N2, device=dev, dtype=DT); aux = rnd(M2, N2); st = torch.randn(4, N2, device=dev).abs() + 0.5
for _ in range(2): _pw_gemm(a, None, K2, wp, out, M2, N2, e_mode=1, e_aux=aux, e_stats=st, e_act=1, want_stats=True)
for _ in range(2): _pw_gemm(a, None, K2, wp, out, M2, N2, residual=aux)
torch.cuda.synchronize()
