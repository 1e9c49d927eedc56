import sys, torch
sys.path.insert(0, "ml-cvnets_amd")
from cvnets_amd import _lib, ops
dev, DT = "cuda:0", torch.bfloat16
B, H, d, h = 1024, 32, 144, 4
rows_ = B * H * H
qkv = torch.randn(rows_, 3 * d, device=dev).to(DT); o = torch.empty(rows_, d, device=dev, dtype=DT); do = torch.randn(rows_, d, device=dev).to(DT)
nseq, S, c = B * 4, H * H // 4, d // h
lse, dsum = torch.empty(nseq * h * S, device=dev), torch.empty(nseq * h * S, device=dev)
dqkv = torch.empty_like(qkv)
args = (nseq, S, h, c, 2, 2, H // 2, H, H, float(c) ** -0.5, 0)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    _lib.call("cvh_attn_fwd", 1, qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), None, *args, s)
    _lib.call("cvh_attn_bwd", 1, qkv.data_ptr(), o.data_ptr(), do.data_ptr(), dqkv.data_ptr(), lse.data_ptr(), dsum.data_ptr(), None, *args, s)
torch.cuda.synchronize()
