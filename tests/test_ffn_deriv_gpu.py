"""Transformer-sized FFN pairing with a STORED derivative (include/cvnets_hip.h: CVH_ACT_GELU_D / CVH_ACT_DERIV): fc1's epilogue stores
GELU'(pre-activation) instead of the pre-activation, fc2's dX GEMM multiplies by it (cvnets/modules/transformer.py:140-155 at ViT-B width).
Against the fp32 torch evaluation of fc2(GELU(fc1(x))) + x, forward and every gradient; and against the unpaired HIP path."""
import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,K,Hd", [(4096, 768, 3072), (2500, 512, 2048)])
def test_stored_derivative_ffn_matches_torch(rows, K, Hd):
    from cvnets_amd import ops

    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn(rows, K, device=DEV, generator=g)).bfloat16()
    w1 = (torch.randn(Hd, K, device=DEV, generator=g) * K ** -0.5).requires_grad_(True)
    b1 = (0.1 * torch.randn(Hd, device=DEV, generator=g)).requires_grad_(True)
    w2 = (torch.randn(K, Hd, device=DEV, generator=g) * Hd ** -0.5).requires_grad_(True)
    b2 = (0.1 * torch.randn(K, device=DEV, generator=g)).requires_grad_(True)
    go = torch.randn(rows, K, device=DEV, generator=g).bfloat16()
    assert ops.ffn_stores_derivative(x, w1, ops.ACT_GELU), "the large-tile kernel does not take this shape"

    def run(paired):
        for t in (w1, b1, w2, b2):
            t.grad = None
        xg = x.clone().requires_grad_(True)
        if paired:
            h, dv = ops.linear(xg, w1, b1, act=ops.ACT_GELU_D, expose_pre=True)
            y = ops.linear(h, w2, b2, residual=xg, in_pre=dv, in_act=ops.ACT_DERIV)
        else:
            y = ops.linear(ops.linear(xg, w1, b1, act=ops.ACT_GELU), w2, b2, residual=xg)
        y.backward(go)
        torch.cuda.synchronize()
        return y.detach().float(), xg.grad.float(), w1.grad.clone(), b1.grad.clone(), w2.grad.clone(), b2.grad.clone()

    got = run(True)
    plain = run(False)
    xr = x.float().requires_grad_(True)
    ps = [t.detach().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    # operands rounded where the bf16 path rounds them (weights, hidden activation)
    h = F.gelu(F.linear(xr, ps[0].bfloat16().float(), ps[1]))
    yr = F.linear(h + (h.bfloat16().float() - h).detach(), ps[2].bfloat16().float(), ps[3]) + xr
    yr.backward(go.float())
    ref = (yr.detach(), xr.grad, ps[0].grad, ps[1].grad, ps[2].grad, ps[3].grad)
    for name, a, b, c in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), got, plain, ref):
        assert l2_err(a, c) < 1.5e-2, (name, l2_err(a, c))
        assert l2_err(a, b) < 1.5e-2, (name, "vs unpaired", l2_err(a, b))
