"""End-to-end parity of the HIP path (through the C ABI) for the whole MobileViT hot path: forward logits, loss, every
parameter gradient and BatchNorm running statistics against (a) the committed golden fixtures produced by the reference
itself (tests/golden/, oracle/make_golden.py) and (b) the live CPU oracle on fresh seeded inputs.

Tolerances (stated per BASELINE.json north_star "within stated fp32/bf16 tolerance"):
  fp32 mode : logits rel-L2 <= 1e-4 (north-star target 1e-3; measured ~1e-6), gradients rel-L2 <= 2e-3 per tensor
  bf16 mode : the yardstick is the REFERENCE'S OWN bf16 path: oracle/make_golden.py also runs the reference under
              torch.autocast(bfloat16) and records how far that moves its logits / loss / gradients from its fp32 run
              (fixture key ref_bf16_autocast_err: e.g. logits 1.7e-2 @256^2 b2, 1.1e-1 for xx_small @32^2 where train-mode
              BatchNorm sees 8 values per channel).  The HIP bf16 path must stay within BF16_SLACK x that deviation
              (measured: at or below 1.0x on every case); eval-mode logits (no batch statistics) <= 3e-2.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu
BF16_SLACK = 1.5
BF16_GRAD_NORM_BOUND = {"mobilevit_xxs_32_b8": 0.37, "mobilevit_s_128_b2": 0.27, "mobilevit_s_256_b2": 0.27, "mobilevit_s_160_b2": 0.26,
                        "mobilevit_xxs_128_b2": 0.36}
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("mobilevit_xxs_32_b8", "xx_small", 8, 32), ("mobilevit_s_128_b2", "small", 2, 128), ("mobilevit_s_256_b2", "small", 2, 256),
         ("mobilevit_s_160_b2", "small", 2, 160),
         # 64 patches x 64 channels in layer_3: the reference LayerNorm takes its channel-first branch; reproduced bug-compatibly (cvh_ln_seq_*)
         ("mobilevit_xxs_128_b2", "xx_small", 2, 128)]


def _build(mode, dtype):
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from oracle.weights import seeded_state_dict

    opts = default_opts(**{"model.classification.mit.mode": mode, "model.classification.mit.dropout": 0.0,
                           "model.classification.classifier_dropout": 0.0})
    model = cvnets_amd.MobileViT(opts)
    shapes = json.load(open(os.path.join(GOLD, f"mobilevit_{mode}_keys.json")))
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
    return logits.detach().float().cpu(), float(loss), {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}


@pytest.mark.parametrize("name,mode,batch,res", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_step_vs_reference_golden(name, mode, batch, res, dtype):
    from oracle.weights import seeded_input, seeded_labels

    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model, sd = _build(mode, dtype)
    x = seeded_input((batch, 3, res, res), seed=1).cuda()
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
    # eval mode has no batch statistics to amplify round-off: bf16 measures 3.0e-3 ... 4.4e-3 on these five fixtures (round 5, MI355X);
    # the bound is 2 x the largest value measured — a wrong rounding point or operand shows up an order of magnitude above it
    assert e_eval < (1e-4 if fp32 else 8e-3), e_eval
    assert e_train < (1e-4 if fp32 else BF16_SLACK * ref_bf16["logits_train"]), (e_train, ref_bf16)
    assert abs(loss - float(gold["loss"])) < (1e-4 if fp32 else max(2e-2, BF16_SLACK * ref_bf16["loss"]))
    names = [str(n) for n in gold["grad_names"]]
    assert names == [k for k, _ in model.named_parameters()]
    gn = torch.tensor([grads[k].norm().item() for k in names], dtype=torch.float64)
    gref = torch.from_numpy(gold["grad_norm"])
    worst = float(((gn - gref).abs() / (gref + 1e-3 * gref.max())).max())
    print(f"[{name} {dtype}] worst per-tensor grad-norm deviation {worst:.2e}")
    # bf16: ABSOLUTE bounds (2 x the value measured on MI355X, round 5: 1.87e-1 / 1.34e-1 / 1.38e-1 / 1.31e-1 / 1.81e-1 in CASES order —
    # 2- and 8-image train-mode BatchNorm), and never looser than the reference's own bf16-autocast deviation allows (which on
    # mobilevit_s_256_b2 would have been 0.89: a bound that catches nothing)
    assert worst < (2e-3 if fp32 else min(BF16_GRAD_NORM_BOUND[name], BF16_SLACK * ref_bf16["grad_norm_worst"])), (worst, ref_bf16)
    for key in gold.files:
        if key.startswith("grad::"):
            e = l2_err(grads[key[6:]], torch.from_numpy(gold[key]))
            assert e < (2e-3 if fp32 else BF16_SLACK * ref_bf16["grad_full_worst"]), (key, e, ref_bf16)
        if key.startswith("bn::"):
            got = model.state_dict()[key[4:]].float().cpu()
            e = l2_err(got, torch.from_numpy(gold[key]))
            assert e < (1e-4 if fp32 else 3e-2), (key, e)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_step_vs_live_oracle(dtype):
    """fresh inputs (not in the fixtures): HIP path vs the CPU oracle, all gradients tensor by tensor."""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels

    model, sd = _build("small", dtype)
    x = seeded_input((3, 3, 64, 96), seed=7)
    y = seeded_labels(3, 1000, seed=7)
    logits, loss, grads = _step(model, x.cuda(), y.cuda())
    o_logits, o_loss, o_grads, o_running = orc.train_step(sd, x, y, mode="small")
    fp32 = dtype == torch.float32
    # bf16 bounds: this 64x96 input leaves layer_5 a 2x3 map (12 values per BatchNorm channel), the regime where the
    # reference's own bf16 run deviates by ~1e-1 in logits / ~2.5e-1 in gradients (fixture xxs_32: 1.1e-1 / 2.4e-1)
    assert l2_err(logits, o_logits) < (1e-4 if fp32 else 1e-1)
    assert abs(loss - float(o_loss)) < (1e-4 if fp32 else 5e-2)
    num = sum(float((grads[k].double() - o_grads[k].double()).pow(2).sum()) for k in o_grads)
    den = sum(float(o_grads[k].double().pow(2).sum()) for k in o_grads)
    g_err = (num / den) ** 0.5
    worst = max(((l2_err(grads[k], o_grads[k]), k) for k in o_grads if o_grads[k].norm() > 1e-3 * den ** 0.5), default=(0, ""))
    print(f"[live oracle {dtype}] global grad rel-L2 {g_err:.2e}; worst tensor {worst}")
    assert g_err < (1e-3 if fp32 else 2.5e-1), g_err
    if fp32:
        assert worst[0] < 5e-3, worst
    sd_after = model.state_dict()
    for k, v in o_running.items():
        assert l2_err(sd_after[k].float().cpu(), v) < (1e-4 if fp32 else 3e-2), k


def test_x_small_vs_live_oracle():
    """the third MobileViT size (x_small: expansion 4, 32..96 channels, transformer dims 96/120/144 -> head dims 24/30/36)"""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    opts = default_opts(**{"model.classification.mit.mode": "x_small", "model.classification.mit.dropout": 0.0,
                           "model.classification.classifier_dropout": 0.0})
    model = cvnets_amd.MobileViT(opts)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    cvnets_amd.set_compute_dtype(torch.float32)
    model = model.cuda()
    x, y = seeded_input((2, 3, 96, 96), seed=9), seeded_labels(2, 1000, seed=9)
    logits, loss, grads = _step(model, x.cuda(), y.cuda())
    o_logits, o_loss, o_grads, _ = orc.train_step(sd, x, y, mode="x_small")
    assert l2_err(logits, o_logits) < 1e-4 and abs(loss - float(o_loss)) < 1e-4
    gmax = max(float(v.norm()) for v in o_grads.values())
    for k, g in o_grads.items():
        assert l2_err(grads[k], g) < 2e-3 or g.norm() < 1e-5 * gmax, (k, l2_err(grads[k], g))


def test_rectangular_and_batch1():
    """edge cases: batch 1, non-square input, eval mode (running statistics)."""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input

    model, sd = _build("xx_small", torch.float32)
    model.eval()
    for shape in [(1, 3, 64, 64), (2, 3, 64, 128), (5, 3, 32, 32)]:
        x = seeded_input(shape, seed=3)
        with torch.no_grad():
            got = model(x.cuda()).float().cpu()
        ref = orc.mobilevit_forward(sd, x, mode="xx_small", training=False)
        assert l2_err(got, ref) < 1e-4, shape


def test_unsupported_variants_fail_loudly():
    """no silent fallbacks: configurations without a HIP kernel raise instead of computing something else."""
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    with pytest.raises(NotImplementedError):
        cvnets_amd.MobileViT(default_opts(**{"model.activation.name": "hard_swish"}))
    with pytest.raises(NotImplementedError):
        cvnets_amd.MobileViT(default_opts(**{"model.normalization.name": "group_norm"}))
    # (cross-attention runs since round 5, an additive mask with it since round 6: tests/test_variants_gpu.py, tests/test_attn_mask_gpu.py)


def test_inplace_param_grads_and_pack_plan_match_autograd_path():
    """bench.py's fast path (flat gradient buckets written in place by the backward kernels + one-launch weight packing) must give
    the same gradients as the plain autograd path, and a second step must see re-packed (updated) weights."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.ddp import DistributedDataParallel
    from oracle.weights import seeded_input, seeded_labels

    model, sd = _build("xx_small", torch.float32)
    x = seeded_input((4, 3, 64, 64), seed=5).cuda()
    y = seeded_labels(4, 1000, seed=5).cuda()
    _, loss_ref, g_ref = _step(model, x, y)
    model.load_state_dict(sd)
    ddp = DistributedDataParallel(model, broadcast_buffers=False)
    ops.set_inplace_param_grads(True)
    try:
        for it in range(2):
            ddp.zero_grad()
            model.train()
            logits = model(x)
            loss = F.cross_entropy(logits.float(), y, label_smoothing=0.1)
            loss.backward()
            if it == 0:
                assert abs(float(loss) - loss_ref) < 1e-5
                gmax = max(float(v.norm()) for v in g_ref.values())
                for k, p in model.named_parameters():  # (mathematically-zero gradients, e.g. key biases, are pure round-off)
                    assert l2_err(p.grad.cpu(), g_ref[k]) < 1e-4 or g_ref[k].norm() < 1e-5 * gmax, k
                with torch.no_grad():  # an "optimizer step": the next forward must use the new weights
                    for p in model.parameters():
                        p.mul_(0.5)
            else:
                assert abs(float(loss) - loss_ref) > 1e-3
    finally:
        ops.set_inplace_param_grads(False)


def test_mobilevit_option_variants_vs_live_oracle_structure():
    """constructor options of the reference that change the graph: no_fuse_local_global_features (block returns conv_proj output) and
    head_dim instead of number_heads; both must build, run forward + backward and keep the reference's parameter names."""
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    cvnets_amd.set_compute_dtype(torch.float32)
    x = torch.randn(2, 3, 64, 64).cuda()
    for extra in ({"model.classification.mit.no_fuse_local_global_features": True},
                  {"model.classification.mit.head_dim": 16, "model.classification.mit.number_heads": None}):
        opts = default_opts(**{"model.classification.mit.mode": "xx_small", **extra})
        model = cvnets_amd.MobileViT(opts).cuda().train()
        out = model(x)
        assert out.shape == (2, 1000)
        out.float().square().mean().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        if "model.classification.mit.no_fuse_local_global_features" in extra:
            assert not any("fusion" in k for k in model.state_dict())
        else:
            assert model.layer_3[1].global_rep[0].pre_norm_mha[1].num_heads == 64 // 16
