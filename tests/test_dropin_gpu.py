"""The drop-in boundary EXECUTED on the GPU (north_star: "exposed behind the existing cvnets.layers / cvnets.modules
registry so engine/training_engine.py ... drive it unmodified"):

  (a) a model built by the REFERENCE's own builder (cvnets.get_model(opts), reference YAML), class-swapped by cvnets_amd.dropin and
      pickled in the authoring container (oracle/make_swapped_fixture.py -> tests/golden/swapped_mobilevit_*.pt) is unpickled here
      — with only cvnets_amd importable — and must reproduce the reference's golden logits / loss / gradients: the reference's own
      module objects (its Sequential containers, attribute values, opts namespace) run the HIP kernels;
  (b) the step is driven the way engine/training_engine.py:257-287 drives it: torch.autocast("cuda", bfloat16) + GradScaler,
      no set_compute_dtype pin, the engine's zero_grad(set_to_none=True);
  (c) the benchmarked configuration itself — hipGraph-captured step, in-place parameter gradients in flat buckets, one-launch weight
      packing, cvh_adamw_multi — is followed for 8 steps against the CPU oracle + torch.optim.AdamW trajectory.
"""
import json
import os
import pickle

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load_swapped(tag, mode):
    from oracle.weights import seeded_tensor

    model = pickle.load(open(os.path.join(GOLD, f"swapped_mobilevit_{tag}.pt"), "rb"))
    shapes = json.load(open(os.path.join(GOLD, f"mobilevit_{mode}_keys.json")))
    named = dict(model.named_parameters())
    named.update(dict(model.named_buffers()))
    assert set(named) == set(shapes)
    for k, t in named.items():  # the fixture stores no values (size): the seeded ones of oracle/weights.py, as in the .npz fixtures
        t.data = seeded_tensor(k, tuple(shapes[k]), 0).to(t.dtype)
    # every module of the tree is a cvnets_amd (or torch) class: nothing of the reference is importable here
    assert {type(m).__module__.split(".")[0] for m in model.modules()} <= {"cvnets_amd", "torch"}
    return model.to("cuda:0")


@pytest.mark.parametrize("tag,mode,name,batch,res", [("xxs", "xx_small", "mobilevit_xxs_32_b8", 8, 32), ("s", "small", "mobilevit_s_128_b2", 2, 128)])
def test_reference_built_model_runs_hip_kernels(tag, mode, name, batch, res):
    import cvnets_amd
    from oracle.weights import seeded_input, seeded_labels

    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = _load_swapped(tag, mode)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        x = seeded_input((batch, 3, res, res), seed=1).cuda()
        y = seeded_labels(batch, 1000, seed=1).cuda()
        model.eval()
        with torch.no_grad():
            le = model(x).float().cpu()
        model.train()
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = F.cross_entropy(logits.float(), y, label_smoothing=0.1)
        loss.backward()
        torch.cuda.synchronize()
        assert l2_err(le, torch.from_numpy(gold["logits_eval"])) < 1e-4
        assert l2_err(logits.detach().float().cpu(), torch.from_numpy(gold["logits_train"])) < 1e-4
        assert abs(float(loss) - float(gold["loss"])) < 1e-4
        names = [str(n) for n in gold["grad_names"]]
        assert names == [k for k, _ in model.named_parameters()]
        gn = torch.tensor([p.grad.float().norm().item() for _, p in model.named_parameters()], dtype=torch.float64)
        gref = torch.from_numpy(gold["grad_norm"])
        assert float(((gn - gref).abs() / (gref + 1e-3 * gref.max())).max()) < 2e-3
        grads = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
        for key in gold.files:
            if key.startswith("grad::"):
                assert l2_err(grads[key[6:]], torch.from_numpy(gold[key])) < 2e-3, key
            if key.startswith("bn::"):
                assert l2_err(model.state_dict()[key[4:]].float().cpu(), torch.from_numpy(gold[key])) < 1e-4, key
    finally:
        cvnets_amd.set_compute_dtype(None)


def test_engine_style_autocast_gradscaler_step():
    """engine/training_engine.py:257-287: `with autocast(enabled, dtype=bf16): pred = model(x); loss = criterion(...)`, then
    `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()` (GradScaler is constructed even for bf16,
    main_train.py:114).  The compute dtype follows autocast (no pin); scaled gradients must come back finite fp32 `.grad`s that the
    scaler can unscale, and the step must agree with the same step taken with the dtype pinned and no scaler."""
    import copy

    import cvnets_amd

    model = _load_swapped("xxs", "xx_small").train()
    pinned = copy.deepcopy(model)
    from oracle.weights import seeded_input, seeded_labels
    x = seeded_input((8, 3, 32, 32), seed=1).cuda()
    y = seeded_labels(8, 1000, seed=1).cuda()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=1024.0)
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(x)
        assert pred.dtype == torch.bfloat16
        loss = F.cross_entropy(pred.float(), y, label_smoothing=0.1)
    scaler.scale(loss).backward()
    assert all(p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in model.parameters())
    scaler.unscale_(opt)
    g_amp = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    scaler.step(opt)
    scaler.update()
    assert scaler.get_scale() == 1024.0  # no inf / nan was found
    # the same step with the dtype pinned and no scaler
    cvnets_amd.set_compute_dtype(torch.bfloat16)
    try:
        pinned.zero_grad(set_to_none=True)
        l2 = F.cross_entropy(pinned(x).float(), y, label_smoothing=0.1)
        l2.backward()
    finally:
        cvnets_amd.set_compute_dtype(None)
    assert abs(float(loss) - float(l2)) < 2e-2
    num = sum(float((g_amp[k].double() - p.grad.double()).pow(2).sum()) for k, p in pinned.named_parameters())
    den = sum(float(p.grad.double().pow(2).sum()) for p in pinned.parameters())
    assert (num / den) ** 0.5 < 1.5e-1  # two bf16 runs of this 8-image, 32x32 case differ by round-off alone (the reference's own bf16 deviation here: 2.4e-1); measured 6e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_captured_training_trajectory_matches_oracle(dtype):
    """bench.py's timed path, 8 steps: hipGraph replay of zero-grad + forward + CE + backward (gradients accumulated IN PLACE into flat
    buckets, weights packed by one launch) + cvh_adamw_multi, against the CPU oracle's gradients fed to torch.optim.AdamW."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.ddp import DistributedDataParallel
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict

    mode, B, res, steps = "xx_small", 8, 64, 8
    from cvnets_amd.layers import default_opts
    opts = default_opts(**{"model.classification.mit.mode": mode, "model.classification.mit.dropout": 0.0,
                           "model.classification.classifier_dropout": 0.0})
    model = cvnets_amd.MobileViT(opts)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd)
    model = model.cuda().train()
    x = seeded_input((B, 3, res, res), seed=3)
    y = seeded_labels(B, 1000, seed=3)
    xg, yg = x.cuda(), y.cuda()
    cvnets_amd.set_compute_dtype(dtype)
    ops.set_inplace_param_grads(True)
    try:
        ddp = DistributedDataParallel(model, bucket_cap_mb=25.0, broadcast_buffers=False)
        ddp.hooks_enabled = False
        opt = cvnets_amd.optim.AdamW([p for p in model.parameters()], lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)

        def fwd_bwd():
            loss = ops.cross_entropy(model(xg), yg, 0.1)
            loss.backward()
            return loss

        # the oracle trajectory starts from the SAME initial state: run it first
        ref_params = {k: v.clone().requires_grad_(False) for k, v in sd.items()}
        ref_list = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if k in dict(model.named_parameters())}
        ref_opt = torch.optim.AdamW(list(ref_list.values()), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        ref_losses = []
        state = {k: v.clone() for k, v in sd.items()}
        for _ in range(steps):
            for k, p in ref_list.items():
                state[k] = p.detach().clone()
            _, o_loss, o_grads, o_running = orc.train_step(state, x, y, mode=mode)
            ref_losses.append(float(o_loss))
            for k, v in o_running.items():
                state[k] = v.clone()
            for k, p in ref_list.items():
                p.grad = o_grads[k].clone()
            ref_opt.step()

        # captured step (eager warm-up on a side stream would advance the trajectory: capture first, replay `steps` times)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snapshot = {k: v.detach().clone() for k, v in model.state_dict().items()}
        with torch.cuda.stream(side):  # one eager step to create lazily-built tensors, then restore the initial state
            ddp.zero_grad()
            fwd_bwd()
            opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(snapshot[k])
            opt._plan["m"].zero_()
            opt._plan["v"].zero_()
            opt._plan["step"].zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ddp.zero_grad()
            static_loss = fwd_bwd()
            opt.step(sync_hyperparameters=False)
        with torch.no_grad():  # capture does not execute: still at the initial state
            for k, v in model.state_dict().items():
                v.copy_(snapshot[k])
            opt._plan["m"].zero_()
            opt._plan["v"].zero_()
            opt._plan["step"].zero_()
        losses = []
        for _ in range(steps):
            g.replay()
            losses.append(float(static_loss))
        torch.cuda.synchronize()
        fp32 = dtype == torch.float32
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) < (2e-3 if fp32 else 8e-2), (losses, ref_losses)
        # parameters after 8 AdamW steps.  Adam normalises every gradient to unit scale, so tensors whose true gradient is (numerically)
        # zero drift by +-lr per step in round-off direction on BOTH sides: compare the update direction where the gradient is significant
        num = den = 0.0
        for k, p in model.named_parameters():
            d_hip = (p.detach().float().cpu() - sd[k]).double()
            d_ref = (ref_list[k].detach() - sd[k]).double()
            num += float((d_hip - d_ref).pow(2).sum())
            den += float(d_ref.pow(2).sum())
        rel = (num / den) ** 0.5
        print(f"[captured trajectory {dtype}] losses {losses[0]:.4f} -> {losses[-1]:.4f} (oracle {ref_losses[-1]:.4f}); update rel-L2 {rel:.3e}")
        assert rel < (0.08 if fp32 else 0.35), rel  # measured 2.1e-2 / 1.3e-1
        assert losses[-1] < losses[0]
    finally:
        ops.set_inplace_param_grads(False)
        cvnets_amd.set_compute_dtype(None)
        ops.release_capture_state()  # the seed snapshot was allocated in the captured graph's memory pool, which dies with this test


def test_reference_built_segmentation_model_runs_hip_kernels():
    """The same boundary for the DeepLabv3-on-MobileViT model of config/segmentation/pascal_voc/deeplabv3_mobilevit.yaml: built by the
    reference's builder, class-swapped (SegEncoderDecoder, DeeplabV3, ASPP, ASPPConv2d, ASPPPooling, UpSample, Dropout2d, ReLU ...),
    pickled, unpickled here without the reference, and checked against the reference's own outputs (tests/test_segmentation_gpu.py
    checks the model built by cvnets_amd's constructors against the same fixture)."""
    import cvnets_amd
    from oracle.seg_oracle import seg_loss
    from oracle.weights import seeded_input, seeded_tensor

    gold = np.load(os.path.join(GOLD, "deeplabv3_mobilevit_s_96_b2.npz"))
    shapes = json.load(open(os.path.join(GOLD, "deeplabv3_mobilevit_s_keys.json")))
    model = pickle.load(open(os.path.join(GOLD, "swapped_deeplabv3_s.pt"), "rb"))
    named = dict(model.named_parameters())
    named.update(dict(model.named_buffers()))
    assert set(named) == set(shapes)
    for k, t in named.items():
        t.data = seeded_tensor(k, tuple(shapes[k]), 0).to(t.dtype)
    assert {type(m).__module__.split(".")[0] for m in model.modules()} <= {"cvnets_amd", "torch"}
    model = model.to("cuda:0")
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        x = seeded_input((2, 3, 96, 96), seed=1).cuda()
        target = torch.from_numpy(gold["target"].astype(np.int64)).cuda()
        model.eval()
        with torch.no_grad():
            assert l2_err(model(x).float().cpu(), torch.from_numpy(gold["mask_eval"])) < 1e-4
        model.train()
        model.zero_grad(set_to_none=True)
        mask, aux = model(x)
        loss = seg_loss(mask.float(), aux.float(), target)
        loss.backward()
        torch.cuda.synchronize()
        assert l2_err(mask.detach().float().cpu(), torch.from_numpy(gold["mask_train"])) < 1e-4
        assert l2_err(aux.detach().float().cpu(), torch.from_numpy(gold["aux_train"])) < 1e-4
        assert abs(float(loss.detach()) - float(gold["loss"])) < 1e-4
        names = [str(n) for n in gold["grad_names"]]
        gn = torch.tensor([dict(model.named_parameters())[k].grad.float().norm().item() for k in names], dtype=torch.float64)
        gref = torch.from_numpy(gold["grad_norm"])
        assert float(((gn - gref).abs() / (gref + 1e-3 * gref.max())).max()) < 2e-3
    finally:
        cvnets_amd.set_compute_dtype(None)


def test_reference_built_ssd_model_runs_hip_kernels():
    """The boundary for the SSD detector of config/detection/ssd_coco/mobilevit.yaml: built by the reference's builder, class-swapped
    (SingleShotMaskDetector, SSDHead, SeparableConv2d, SSDAnchorGenerator ...), pickled without its host-side matcher, unpickled here
    without the reference, and checked against the reference's own training outputs."""
    import cvnets_amd
    from oracle.weights import seeded_input, seeded_tensor

    gold = np.load(os.path.join(GOLD, "ssd_mobilevit_s_160_b2.npz"))
    shapes = json.load(open(os.path.join(GOLD, "ssd_mobilevit_s_keys.json")))
    model = pickle.load(open(os.path.join(GOLD, "swapped_ssd_s.pt"), "rb"))
    named = dict(model.named_parameters())
    named.update(dict(model.named_buffers()))
    assert set(named) == set(shapes)
    for k, t in named.items():
        t.data = seeded_tensor(k, tuple(shapes[k]), 0).to(t.dtype)
    assert {type(m).__module__.split(".")[0] for m in model.modules()} <= {"cvnets_amd", "torch"}
    model = model.to("cuda:0").train()
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        x = seeded_input((2, 3, 160, 160), seed=1).cuda()
        out = model(x)
        assert l2_err(out["scores"].detach().float().cpu(), torch.from_numpy(gold["scores_train"])) < 1e-4
        assert l2_err(out["boxes"].detach().float().cpu(), torch.from_numpy(gold["boxes_train"])) < 1e-4
        g = torch.Generator().manual_seed(9)
        t_s, t_b = torch.randn(out["scores"].shape, generator=g).cuda(), torch.randn(out["boxes"].shape, generator=g).cuda()
        loss = F.mse_loss(out["scores"].float(), t_s) + F.mse_loss(out["boxes"].float(), t_b)
        loss.backward()
        assert abs(float(loss.detach()) - float(gold["loss"])) < 1e-4
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    finally:
        cvnets_amd.set_compute_dtype(None)
