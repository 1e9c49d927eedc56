import sys, torch
sys.path.insert(0, "ml-cvnets_amd"); sys.path.insert(0, "tests")
from cvnets_amd import ops
from test_kernels_gpu import _attn_ref, _rand
def P(*a):
    print(*a, flush=True)
for dtype in (torch.float32, torch.bfloat16):
    for causal in (True, False):
        B, S, h, c = 3, 77, 8, 64
        d = h * c
        qkv = _rand(B * S, 3 * d, seed=1).to(dtype).requires_grad_(True)
        kpm = torch.zeros(B, S, device="cuda"); kpm[:, -9:] = 1
        o = ops.attention(qkv, h, (B, S, 1, 1, S, 1, S), causal=causal, key_padding_mask=kpm)
        torch.cuda.synchronize(); P("fwd ok", dtype, causal)
        qr = qkv.detach().float().requires_grad_(True)
        ref = _attn_ref(qr, B, S, h, causal, kpm)
        go = _rand(B * S, d, seed=2).to(dtype)
        (g,) = torch.autograd.grad(o, [qkv], go)
        torch.cuda.synchronize(); P("bwd ok")
        (r,) = torch.autograd.grad(ref, [qr], go.float())
        g = g.float().view(B, S, 3, d); r = r.view(B, S, 3, d)
        for i, n in enumerate("qkv"):
            e = (g[:, :, i] - r[:, :, i])
            per_s = e.abs().amax(dim=(0, 2))
            bad = (per_s > 1e-2 * r[:, :, i].abs().max()).nonzero().flatten().tolist()
            P(dtype, causal, n, float(e.abs().max()), float(r[:, :, i].abs().max()), "nan" if torch.isnan(g[:, :, i]).any() else "", "bad rows:", bad[:20], len(bad))
