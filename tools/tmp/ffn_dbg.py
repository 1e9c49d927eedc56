import sys, torch, torch.nn.functional as F
sys.path.insert(0, "ml-cvnets_amd"); sys.path.insert(0, "tests")
from util import l2_err
from cvnets_amd import ops
DEV = "cuda:0"
rows, K, Hd = 4096, 768, 3072
g = torch.Generator(device=DEV).manual_seed(0)
x = (torch.randn(rows, K, device=DEV, generator=g)).bfloat16()
w1 = (torch.randn(Hd, K, device=DEV, generator=g) * K ** -0.5).requires_grad_(True)
b1 = (0.1 * torch.randn(Hd, device=DEV, generator=g)).requires_grad_(True)
w2 = (torch.randn(K, Hd, device=DEV, generator=g) * Hd ** -0.5).requires_grad_(True)
b2 = (0.1 * torch.randn(K, device=DEV, generator=g)).requires_grad_(True)
go = torch.randn(rows, K, device=DEV, generator=g).bfloat16()
def run(paired):
    for t in (w1, b1, w2, b2):
        t.grad = None
    xg = x.clone().requires_grad_(True)
    if paired == 1:
        h, dv = ops.linear(xg, w1, b1, act=ops.ACT_GELU_D, expose_pre=True)
        y = ops.linear(h, w2, b2, residual=xg, in_pre=dv, in_act=ops.ACT_DERIV)
    elif paired == 2:
        h, dv = ops.linear(xg, w1, b1, act=ops.ACT_GELU, expose_pre=True)
        y = ops.linear(h, w2, b2, residual=xg, in_pre=dv, in_act=ops.ACT_GELU)
    else:
        y = ops.linear(ops.linear(xg, w1, b1, act=ops.ACT_GELU), w2, b2, residual=xg)
    y.backward(go)
    torch.cuda.synchronize()
    return y.detach().float(), xg.grad.float(), w1.grad.clone(), b1.grad.clone(), w2.grad.clone(), b2.grad.clone()
xr = x.float().requires_grad_(True)
ps = [t.detach().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
h = F.gelu(F.linear(xr, ps[0].bfloat16().float(), ps[1]))
yr = F.linear(h.bfloat16().float() + (h - h.detach()), ps[2].bfloat16().float(), ps[3]) + xr
yr.backward(go.float())
ref = (yr.detach(), xr.grad, ps[0].grad, ps[1].grad, ps[2].grad, ps[3].grad)
for mode in (0, 2, 1):
    got = run(mode)
    print(mode, [(n, round(l2_err(a, c), 5)) for n, a, c in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), got, ref)])
