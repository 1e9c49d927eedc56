"""MobileViTv2 (GroupNorm(1) "layer_norm_2d", separable linear self-attention over unfolded patches, conv FFN)
through the HIP path: the reference's golden fixtures (tests/golden/mobilevitv2_*.npz, oracle/make_golden.py), the live CPU
oracle, and the two new kernel families (csrc/linattn.hip) one by one against plain PyTorch fp32 restatements that DO unfold.

Tolerances: fp32 mode logits rel-L2 <= 1e-4, per-tensor gradients <= 2e-3.  bf16 mode: within BF16_SLACK x the reference's own
bf16-autocast deviation recorded in each fixture (large for these tiny-batch train-mode BatchNorm cases: 6e-2 .. 1.4e-1 in the
logits), and the kernel-level tests bound the bf16 kernels tightly (inputs rounded to bf16 on both sides)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu
BF16_SLACK = 1.5
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("mobilevitv2_w050_64_b2", 0.5, 2, (64, 64)), ("mobilevitv2_w100_224_b2", 1.0, 2, (224, 224)),
         ("mobilevitv2_w075_96x160_b3", 0.75, 3, (96, 160))]


def _build(wm, dtype):
    import cvnets_amd
    from oracle.weights import seeded_state_dict

    model = cvnets_amd.build_mobilevit_v2(wm)
    shapes = json.load(open(os.path.join(GOLD, f"mobilevitv2_w{int(round(wm * 100)):03d}_keys.json")))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    cvnets_amd.set_compute_dtype(dtype)
    return model.to("cuda:0"), sd


def _step(model, x, y):
    model.train()
    model.zero_grad(set_to_none=True)
    logits = model(x)
    loss = F.cross_entropy(logits.float(), y, label_smoothing=0.1)
    loss.backward()
    return logits.detach().float().cpu(), float(loss.detach()), {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}


@pytest.mark.parametrize("name,wm,batch,hw", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_v2_train_step_vs_reference_golden(name, wm, batch, hw, dtype):
    from oracle.weights import seeded_input, seeded_labels

    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model, sd = _build(wm, dtype)
    x = seeded_input((batch, 3) + hw, seed=1).cuda()
    y = seeded_labels(batch, 1000, seed=1).cuda()
    model.eval()
    with torch.no_grad():
        le = model(x).float().cpu()
    logits, loss, grads = _step(model, x, y)
    fp32 = dtype == torch.float32
    ref_bf16 = json.loads(str(gold["ref_bf16_autocast_err"]))
    e_eval = l2_err(le, torch.from_numpy(gold["logits_eval"]))
    e_train = l2_err(logits, torch.from_numpy(gold["logits_train"]))
    print(f"[{name} {dtype}] logits rel-L2 eval {e_eval:.2e} train {e_train:.2e} loss {loss:.5f} vs {float(gold['loss']):.5f}")
    assert e_eval < (1e-4 if fp32 else 3e-2), e_eval
    assert e_train < (1e-4 if fp32 else BF16_SLACK * ref_bf16["logits_train"]), (e_train, ref_bf16)
    # bf16 floor 3e-2 on a loss of ~7: with 2-3 images per batch the train-mode statistics amplify single bf16 roundings (measured 2.05e-2 on
    # the 96x160 case with the stem kernels, whose first conv sums its 27 products in another order than the im2col GEMM did; 1.6e-2
    # before; the result itself is bit-reproducible)
    assert abs(loss - float(gold["loss"])) < (1e-4 if fp32 else max(3e-2, BF16_SLACK * ref_bf16["loss"]))
    names = [str(n) for n in gold["grad_names"]]
    assert names == [k for k, _ in model.named_parameters()]
    gn = torch.tensor([grads[k].norm().item() for k in names], dtype=torch.float64)
    gref = torch.from_numpy(gold["grad_norm"])
    # floor 1e-2 x the largest gradient norm: MobileViTv2 has parameters whose true gradient is exactly zero (a GroupNorm bias in front of
    # conv_proj + BatchNorm, the query bias under softmax): what both implementations return there is cancellation noise that depends
    # on the summation order (BatchNorm statistics are reduced with LDS float atomics), so it must not be compared tightly
    worst = float(((gn - gref).abs() / (gref + 1e-2 * gref.max())).max())
    print(f"[{name} {dtype}] worst per-tensor grad-norm deviation {worst:.2e}")
    assert worst < (2e-3 if fp32 else BF16_SLACK * ref_bf16["grad_norm_worst"]), (worst, ref_bf16)
    for key in gold.files:
        if key.startswith("grad::"):
            e = l2_err(grads[key[6:]], torch.from_numpy(gold[key]))
            print(f"   {key} rel-L2 {e:.2e}")
            assert e < (2e-3 if fp32 else BF16_SLACK * ref_bf16["grad_full_worst"]), (key, e, ref_bf16)
        if key.startswith("bn::"):
            e = l2_err(model.state_dict()[key[4:]].float().cpu(), torch.from_numpy(gold[key]))
            assert e < (1e-4 if fp32 else 3e-2), (key, e)


def test_v2_vs_live_oracle_all_gradients():
    """fresh inputs; 7x7 / 14x14 maps exercise the align_corners=True resize (224 -> layer_5 7x7 -> 8x8): every gradient vs the oracle."""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels

    model, sd = _build(0.5, torch.float32)
    x = seeded_input((2, 3, 112, 144), seed=21)  # layer_4: 7x9 -> 8x10, layer_5: 4x5 -> 4x6
    y = seeded_labels(2, 1000, seed=21)
    logits, loss, grads = _step(model, x.cuda(), y.cuda())
    o_logits, o_loss, o_grads, o_running = orc.generic_train_step(orc.mobilevit_v2_forward, sd, x, y, width_multiplier=0.5)
    assert l2_err(logits, o_logits) < 1e-4
    assert abs(loss - float(o_loss)) < 1e-4
    gmax = max(float(v.norm()) for v in o_grads.values())
    for k, g in o_grads.items():
        assert l2_err(grads[k], g) < 2e-3 or g.norm() < 1e-4 * gmax, (k, l2_err(grads[k], g), float(g.norm()), gmax)
    for k, v in o_running.items():
        assert l2_err(model.state_dict()[k].float().cpu(), v) < 1e-4, k


# ------------------------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------------------------
def _q(t, dtype):
    return t.to(dtype).float()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 64, 8, 8), (3, 192, 14, 10), (1, 128, 32, 32), (130, 96, 4, 4), (2, 512, 6, 6)])
def test_group_norm1(dtype, cfg):
    from cvnets_amd import ops

    B, C, H, W = cfg
    g = torch.Generator().manual_seed(3)
    x = _q(torch.randn(B, C, H, W, generator=g) * 1.5 + 0.7, dtype)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dy = _q(torch.randn(B, C, H, W, generator=g), dtype)
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr = F.group_norm(xr, 1, gr, br, 1e-5)
    yr.backward(dy)
    xg = ops.to_nhwc(x.cuda(), dtype).detach().requires_grad_()
    gg, bg = gamma.cuda().requires_grad_(), beta.cuda().requires_grad_()
    y = ops.group_norm1(xg, gg, bg, 1e-5)
    y.backward(ops.to_nhwc(dy.cuda(), dtype))
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert l2_err(y.float().cpu(), yr.detach()) < tol
    assert l2_err(xg.grad.float().cpu(), xr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert l2_err(gg.grad.cpu(), gr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert l2_err(bg.grad.cpu(), br.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


def _ref_linattn(kvq_nchw, C, ph, pw):
    """the reference formulation ON THE UNFOLDED tensor: F.unfold -> softmax over patches -> context -> relu(v)*cv -> F.fold."""
    B, _, H, W = kvq_nchw.shape
    k, v, q = kvq_nchw[:, :C], kvq_nchw[:, C:2 * C], kvq_nchw[:, 2 * C:2 * C + 1]

    def unf(t):
        return F.unfold(t, (ph, pw), stride=(ph, pw)).reshape(B, t.shape[1], ph * pw, -1)

    s = F.softmax(unf(q), dim=-1)
    cv = torch.sum(unf(k) * s, dim=-1, keepdim=True)
    out = F.relu(unf(v)) * cv
    return F.fold(out.reshape(B, C * ph * pw, -1), (H, W), (ph, pw), stride=(ph, pw))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 64, 8, 8, 2, 2), (3, 192, 14, 10, 2, 2), (1, 128, 32, 32, 2, 2), (2, 256, 4, 4, 2, 2), (2, 72, 12, 9, 4, 3),
                                 (1, 512, 6, 6, 2, 2), (2, 32, 2, 2, 2, 2)])
def test_linear_attention_core(dtype, cfg):
    from cvnets_amd import ops

    B, C, H, W, ph, pw = cfg
    g = torch.Generator().manual_seed(4)
    kvq = torch.randn(B, 2 * C + 8, H, W, generator=g)
    kvq[:, 2 * C] *= 3.0  # sharper softmax
    kvq[:, 2 * C + 1:] = 0
    kvq = _q(kvq, dtype)
    dout = _q(torch.randn(B, C, H, W, generator=g), dtype)
    kr = kvq.clone().requires_grad_()
    outr = _ref_linattn(kr, C, ph, pw)
    outr.backward(dout)
    kg = ops.to_nhwc(kvq.cuda(), dtype).detach().requires_grad_()
    out = ops.linear_attention(kg, C, ph, pw)
    out.backward(ops.to_nhwc(dout.cuda(), dtype))
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert l2_err(out.float().cpu(), outr.detach()) < tol
    gref = kr.grad.clone()
    got = kg.grad.float().cpu()
    assert float(got[:, 2 * C + 1:].abs().max()) == 0.0
    for lo, hi, nm in ((0, C, "dkey"), (C, 2 * C, "dvalue"), (2 * C, 2 * C + 1, "dquery")):
        e = l2_err(got[:, lo:hi], gref[:, lo:hi])
        assert e < (1e-4 if dtype == torch.float32 else 1.5e-2), (nm, e)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_self_attention_layer(dtype):
    """the whole layer (qkv_proj with the permuted weight rows, core, out_proj + residual) vs the unfolded PyTorch restatement,
    including the parameter gradients arriving in the reference's [1+2C, C] row order."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.layers import LinearSelfAttention, default_opts

    cvnets_amd.set_compute_dtype(dtype)
    B, C, H, W = 2, 64, 8, 12
    layer = LinearSelfAttention(default_opts(), C).cuda()
    g = torch.Generator().manual_seed(5)
    x = _q(torch.randn(B, C, H, W, generator=g), dtype)
    dy = _q(torch.randn(B, C, H, W, generator=g), dtype)
    w = {k: v.detach().cpu().clone() for k, v in layer.state_dict().items()}
    wq = {k: (_q(v, dtype) if "weight" in k else v).requires_grad_() for k, v in w.items()}
    xr = x.clone().requires_grad_()
    xu = F.unfold(xr, (2, 2), stride=(2, 2)).reshape(B, C, 4, -1)
    qkv = F.conv2d(xu, wq["qkv_proj.block.conv.weight"], wq["qkv_proj.block.conv.bias"])
    q, k, v = torch.split(qkv, [1, C, C], dim=1)
    o = F.relu(v) * torch.sum(k * F.softmax(q, dim=-1), dim=-1, keepdim=True)
    o = F.conv2d(o, wq["out_proj.block.conv.weight"], wq["out_proj.block.conv.bias"])
    yr = F.fold(o.reshape(B, C * 4, -1), (H, W), (2, 2), stride=(2, 2)) + xr
    yr.backward(dy)
    xg = ops.to_nhwc(x.cuda(), dtype).detach().requires_grad_()
    y = layer(xg, patch_hw=(2, 2), residual=xg)
    y.backward(ops.to_nhwc(dy.cuda(), dtype))
    f32 = dtype == torch.float32
    assert l2_err(y.float().cpu(), yr.detach()) < (1e-5 if f32 else 1e-2)
    assert l2_err(xg.grad.float().cpu(), xr.grad) < (1e-4 if f32 else 2e-2)
    for k_, p in layer.named_parameters():
        assert l2_err(p.grad.float().cpu(), wq[k_].grad) < (1e-4 if f32 else 2e-2), k_


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 16, 7, 7, 8, 8), (1, 24, 7, 9, 8, 10), (2, 8, 5, 4, 6, 4), (1, 8, 1, 3, 2, 4)])
def test_resize_bilinear_align_corners(dtype, cfg):
    from cvnets_amd import ops

    B, C, H, W, Ho, Wo = cfg
    g = torch.Generator().manual_seed(6)
    x = _q(torch.randn(B, C, H, W, generator=g), dtype)
    dy = _q(torch.randn(B, C, Ho, Wo, generator=g), dtype)
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=True)
    yr.backward(dy)
    xg = ops.to_nhwc(x.cuda(), dtype).detach().requires_grad_()
    y = ops.resize_bilinear(xg, Ho, Wo, align_corners=True)
    y.backward(ops.to_nhwc(dy.cuda(), dtype))
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert l2_err(y.float().cpu(), yr.detach()) < tol
    assert l2_err(xg.grad.float().cpu(), xr.grad) < tol


def test_v2_with_dropout_trains_and_is_layout_consistent():
    """mitv2.dropout / ffn_dropout > 0 (not the shipped default): the un-fused dropout + residual-add path on NHWC maps runs forward and
    backward, masks are regenerated consistently in backward (kept positions of dX == kept positions of Y), eval is deterministic."""
    import cvnets_amd
    from cvnets_amd import ops

    cvnets_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    x = ops.to_nhwc(torch.randn(2, 16, 6, 8).cuda()).detach().requires_grad_()
    y = ops.dropout(x, 0.5, True)
    assert y.stride() == x.stride()
    kept = y != 0
    assert 0.3 < float(kept.float().mean()) < 0.7
    assert torch.allclose(y[kept], 2.0 * x.detach()[kept])
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, kept)
    model = cvnets_amd.build_mobilevit_v2(0.5, **{"model.classification.mitv2.dropout": 0.1, "model.classification.mitv2.ffn_dropout": 0.1}).cuda()
    inp = torch.randn(2, 3, 64, 64).cuda()
    model.train()
    out = model(inp)
    out.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(inp), model(inp))


@pytest.mark.parametrize("wm,batch,hw", [(2.0, 2, (64, 64)), (1.5, 1, (96, 64))])
def test_v2_other_widths_vs_live_oracle(wm, batch, hw):
    """width 2.0 is what config/classification/imagenet/mobilevit_v2.yaml ships (attention dims 256 / 384 / 512: the 64-lane-per-row
    variant of the linear-attention kernels); 1.5 with batch 1 on a rectangular input covers the odd sizes in between."""
    import cvnets_amd
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict

    model = cvnets_amd.build_mobilevit_v2(wm)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    cvnets_amd.set_compute_dtype(torch.float32)
    model = model.cuda()
    x, y = seeded_input((batch, 3) + hw, seed=5), seeded_labels(batch, 1000, seed=5)
    if batch == 1:  # train-mode BatchNorm over a 2x2 map of one image is degenerate: compare the eval forward only
        model.eval()
        with torch.no_grad():
            got = model(x.cuda()).float().cpu()
        assert l2_err(got, orc.mobilevit_v2_forward(sd, x, width_multiplier=wm, training=False)) < 1e-4
        return
    logits, loss, grads = _step(model, x.cuda(), y.cuda())
    o_logits, o_loss, o_grads, _ = orc.generic_train_step(orc.mobilevit_v2_forward, sd, x, y, width_multiplier=wm)
    assert l2_err(logits, o_logits) < 1e-4 and abs(loss - float(o_loss)) < 1e-4
    gmax = max(float(v.norm()) for v in o_grads.values())
    for k, g in o_grads.items():
        assert l2_err(grads[k], g) < 2e-3 or g.norm() < 1e-4 * gmax, (k, l2_err(grads[k], g))
