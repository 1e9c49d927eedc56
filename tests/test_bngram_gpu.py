"""cvh_bn_apply_gram (csrc/bngram.hip): the BatchNorm apply that closes a conv -> BatchNorm (-> act) (+ residual) chain
(cvnets/layers/conv_layer.py:254-255; cvnets/modules/mobilenetv2.py:231-235) and, from the same pass, the Gram matrix G = y^T y and the
column sums 1^T y of the stored result — what the next fused InvertedResidual block needs of its input (csrc/dwx.hip).

Kernel level: the stored tensor must equal cvh_bn_apply's bit for bit; G and s against float64 sums over the stored values (fp32 MFMA
accumulation in 512 x waves partial sums, then double: 1e-5 of the largest entry).  Model level: a chain stem-like ConvLayer2d -> three
InvertedResidual blocks trained for three steps with the mechanism on and off (CVH_GRAM_OUT's run-time switch) — the producer learns on the
first step that its consumer wants the Gram matrix and forms it from the second step on; outputs, gradients and running statistics agree
to the bf16 bounds of test_fused_ir_gpu.py (the two paths differ in the summation order of G only)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rows,C,act,res", [
    (1000, 16, 1, False), (70001, 16, 0, False), (33, 16, 1, True),
    (4099, 32, 0, False), (131072 + 17, 32, 1, True),
    (5000, 64, 0, True), (31, 64, 1, False), (262144, 64, 0, True), (100003, 64, 2, False),
])
def test_bn_apply_gram_matches_separate_passes(rows, C, act, res):
    from cvnets_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(rows + C)
    x = torch.randn(rows, C, device=DEV, generator=g).bfloat16()
    r = torch.randn(rows, C, device=DEV, generator=g).bfloat16() if res else None
    sc = torch.rand(C, device=DEV, generator=g) + 0.5
    sh = torch.randn(C, device=DEV, generator=g) * 0.3
    ref = torch.empty_like(x)
    _lib.call("cvh_bn_apply", 1, x.data_ptr(), sc.data_ptr(), sh.data_ptr(), act, r.data_ptr() if res else None, ref.data_ptr(), rows, C, _st())
    R = _lib.query("cvh_bn_apply_gram_rows", rows, C)
    assert R > 0
    n = C * C + C
    part = torch.full((R * n,), float("nan"), device=DEV)
    gs = torch.full((n,), float("nan"), device=DEV)
    y = torch.full((rows, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    _lib.call("cvh_bn_apply_gram", 1, x.data_ptr(), sc.data_ptr(), sh.data_ptr(), act, r.data_ptr() if res else None, y.data_ptr(), rows, C,
              part.data_ptr(), R, gs.data_ptr(), _st())
    torch.cuda.synchronize()
    if act in (0, 1):
        assert torch.equal(y.view(torch.int16), ref.view(torch.int16))
    else:  # the run-time activation branch (erf GELU) is compiled in another translation unit: one bf16 ulp
        assert float(((y.float() - ref.float()).abs() / (ref.float().abs() + 1e-3)).max()) < 2 ** -7
    yd = y.double()
    G = yd.t() @ yd
    s = yd.sum(0)
    Gk = gs[:C * C].reshape(C, C).double()
    assert float((Gk - G).abs().max() / G.abs().max()) < 1e-5
    assert torch.equal(Gk, Gk.t())
    assert float((gs[C * C:].double() - s).abs().max() / (yd.abs().sum(0).max() + 1e-9)) < 1e-5


def test_rows_query_rejects_uncovered_widths():
    from cvnets_amd import _lib
    assert _lib.query("cvh_bn_apply_gram_rows", 4096, 96) == 0
    assert _lib.query("cvh_bn_apply_gram_rows", 4096, 24) == 0
    assert _lib.query("cvh_bn_apply_gram_rows", 4096, 64) > 0


def _chain(seed):
    import torch.nn as nn
    from cvnets_amd.layers import ConvLayer2d, default_opts
    from cvnets_amd.modules import InvertedResidual

    torch.manual_seed(seed)
    opts = default_opts()
    m = nn.Sequential(
        ConvLayer2d(opts, 8, 16, kernel_size=3, stride=1, use_norm=True, use_act=True),  # -> C = 16 (producer: ConvBNAct)
        InvertedResidual(opts, 16, 32, stride=1, expand_ratio=4),                        # consumer of 16, producer of 32
        InvertedResidual(opts, 32, 64, stride=2, expand_ratio=4),                        # consumer of 32, producer of 64
        InvertedResidual(opts, 64, 64, stride=1, expand_ratio=4),                        # residual block
        InvertedResidual(opts, 64, 64, stride=1, expand_ratio=4),
    )
    return m


def _train(m, xs, enabled, monkeypatch):
    import cvnets_amd
    from cvnets_amd import ops

    cvnets_amd.set_compute_dtype(torch.bfloat16)
    monkeypatch.setattr(ops, "_GRAM_OUT", enabled)
    m = m.to(DEV).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)  # same parameters in every step: the runs differ in where G and s come from, nothing else
    outs, tagged = [], []
    for x in xs:
        opt.zero_grad(set_to_none=True)
        xin = ops.to_nhwc(x.to(DEV), torch.bfloat16)
        h = xin
        seen = 0
        for layer in m:
            h = layer(h)
            seen += 1 if getattr(h, "_cvh_gram", None) is not None else 0
        tagged.append(seen)
        loss = (h.float() ** 2).mean()
        loss.backward()
        opt.step()
        outs.append(h.detach().float().clone())
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}
    bufs = {k: v.detach().float().clone() for k, v in m.state_dict().items() if "running" in k}
    return outs, grads, bufs, tagged


def test_chain_learns_and_agrees_with_separate_passes(monkeypatch):
    import copy
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(4, 8, 32, 48, generator=g) for _ in range(3)]
    base = _chain(3)
    o0, g0, b0, t0 = _train(copy.deepcopy(base), xs, False, monkeypatch)
    o1, g1, b1, t1 = _train(copy.deepcopy(base), xs, True, monkeypatch)
    assert t0 == [0, 0, 0]
    # step 1: nobody has asked yet; from step 2 on the first four layers' outputs (16, 32, 64, 64 channels, each feeding a fused block) carry it
    assert t1[0] == 0 and t1[1] == 4 and t1[2] == 4, t1
    for a, b in zip(o0, o1):
        assert float((a - b).abs().max() / a.abs().max()) < 3e-2
    # gradients that are analytically zero (the bias of a BatchNorm whose output feeds a conv -> train-mode BatchNorm) are rounding noise on
    # both sides: deviations are measured against the parameter's own gradient norm plus 2 % of the largest gradient norm of the model
    gmax = max(float(v.norm()) for v in g0.values())
    worst = max((float((g0[k] - g1[k]).norm() / (g0[k].norm() + 0.02 * gmax)), k) for k in g0)
    print("worst gradient deviation", worst)
    assert worst[0] < 3e-2, worst
    for k in b0:
        assert float((b0[k] - b1[k]).abs().max() / (b0[k].abs().max() + 1e-3)) < 3e-2, k  # (a mean that is analytically zero is exactly zero only on the analytic path)
