"""Row x1 on the GPU: the loop body of the reference's `Trainer.train_epoch` (tests/engine_loop.py — pinned bit for bit against the
reference Trainer on the CPU box by tests/test_launch_cpu.py) drives the reference-BUILT, class-swapped, pickled MobileViT through the
HIP kernels for several iterations — scheduler hook, autocast, criterion call convention, GradScaler, gradient clipping, accumulation,
zero_grad(set_to_none=True) — and the trajectory is checked against the CPU oracle driven by the same optimizer."""
import copy
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import engine_loop  # noqa: E402
from test_dropin_gpu import _load_swapped  # noqa: E402

pytestmark = pytest.mark.gpu


class _ConstLR:
    def update_lr(self, optimizer, epoch, curr_iter):
        for g in optimizer.param_groups:
            g["lr"] = 0.05 * (0.5 ** curr_iter)   # the engine's scheduler rewrites the group rates before every iteration (:246-248)
        return optimizer


def _batches(n, B=8, res=32):
    from oracle.weights import seeded_input, seeded_labels
    return [{"samples": seeded_input((B, 3, res, res), seed=10 + i), "targets": seeded_labels(B, 1000, seed=10 + i)} for i in range(n)]


def test_engine_loop_fp32_three_iterations_vs_oracle():
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from oracle import mobilevit_oracle as orc

    model = _load_swapped("xxs", "xx_small")
    for m in model.modules():  # no RNG in the step: the oracle has no dropout stream to match
        if isinstance(m, cvnets_amd.layers.Dropout):
            m.p = 0.0
    sd0 = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
    crit = cvnets_amd.CrossEntropy(default_opts(**{"loss.classification.cross_entropy.label_smoothing": 0.1}))
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    batches = _batches(3)
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        n, losses = engine_loop.train_iterations(model, crit, opt, _ConstLR(), torch.amp.GradScaler("cuda", enabled=False), batches,
                                                 device="cuda:0", max_norm=5.0)
    finally:
        cvnets_amd.set_compute_dtype(None)
    assert n == 3
    # CPU oracle + the same torch optimizer / clipping / schedule
    names = [k for k, _ in model.named_parameters()]
    ref = {k: torch.nn.Parameter(sd0[k].clone()) for k in names}
    ropt = torch.optim.SGD(list(ref.values()), lr=0.05, momentum=0.9, weight_decay=1e-4)
    state = {k: v.clone() for k, v in sd0.items()}
    ref_losses = []
    for i, b in enumerate(batches):
        _ConstLR().update_lr(ropt, 0, i)
        for k in names:
            state[k] = ref[k].detach().clone()
        _, l, grads, running = orc.train_step(state, b["samples"], b["targets"], mode="xx_small")
        ref_losses.append(float(l))
        state.update({k: v.clone() for k, v in running.items()})
        for k in names:
            ref[k].grad = grads[k].clone()
        torch.nn.utils.clip_grad_norm_(list(ref.values()), max_norm=5.0)
        ropt.step()
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-4, (losses, ref_losses)
    num = den = 0.0
    for k, p in model.named_parameters():
        d_hip, d_ref = (p.detach().float().cpu() - sd0[k]).double(), (ref[k].detach() - sd0[k]).double()
        num += float((d_hip - d_ref).pow(2).sum())
        den += float(d_ref.pow(2).sum())
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5   # update after 3 clipped SGD-momentum steps, relative L2
    for k, b in model.named_buffers():
        if b.dtype.is_floating_point:
            assert torch.allclose(b.float().cpu(), state[k], rtol=1e-4, atol=1e-5), k  # BatchNorm running statistics followed too


def test_engine_loop_bf16_autocast_scaler_accumulation():
    """the engine as configured for BASELINE configs[1]: bf16 autocast + GradScaler (always constructed, main_train.py:114),
    accum_freq = 2, clipping; 4 batches -> 2 updates; finite fp32 gradients, scale untouched, and the same trajectory when the model is
    wrapped in cvnets_amd.ddp (hooks on, no process group: the wrapper only provides the flat gradient storage)."""
    import cvnets_amd
    from cvnets_amd.ddp import DistributedDataParallel
    from cvnets_amd.layers import default_opts

    base = _load_swapped("xxs", "xx_small")
    for m in base.modules():
        if isinstance(m, cvnets_amd.layers.Dropout):
            m.p = 0.0
    crit = cvnets_amd.CrossEntropy(default_opts(**{"loss.classification.cross_entropy.label_smoothing": 0.1}))
    batches = _batches(4)
    out = []
    for wrap in (False, True):
        model = copy.deepcopy(base)
        net = DistributedDataParallel(model) if wrap else model
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
        scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=256.0)
        n, losses = engine_loop.train_iterations(net, crit, opt, _ConstLR(), scaler, batches, device="cuda:0", accum_freq=2, max_norm=5.0,
                                                 amp_dtype=torch.bfloat16)
        assert n == 2 and scaler.get_scale() == 256.0 and all(l == l for l in losses)
        assert all(torch.isfinite(p).all() for p in model.parameters())
        out.append(([p.detach().float().cpu().clone() for p in model.parameters()], losses))
    # the two runs differ only by the run-to-run order of fp32 atomics inside the kernels, i.e. by bf16 round-off — which on this 8-image
    # 32x32 case is large (train-mode BatchNorm over a handful of values: the reference's own bf16 run deviates from fp32 by 2.4e-1 here);
    # measured 1.5e-1
    p0 = [p.detach().float().cpu() for p in base.parameters()]
    num = sum(float(((a - c) - (b - c)).double().pow(2).sum()) for a, b, c in zip(out[0][0], out[1][0], p0))
    den = sum(float((a - c).double().pow(2).sum()) for a, c in zip(out[0][0], p0))
    assert (num / den) ** 0.5 < 3.5e-1, (num / den) ** 0.5
    # losses of the 4 batches: the first two are equal to round-off, the last two come after an update each and drift with it; over ~20
    # runs the largest gap seen was 4.0e-2 (on a loss of 7.5)
    assert all(abs(a - b) < 8e-2 for a, b in zip(out[0][1], out[1][1])), (out[0][1], out[1][1])


def test_engine_loop_with_the_launchers_fused_adamw_follows_torch_adamw():
    """substitution 4 of cvnets_amd/launch.py: the reference's Trainer loop (zero_grad(set_to_none=True), GradScaler, clipping, a scheduler that
    rewrites the group rates, accumulation) stepping `cvnets_amd.optim.AdamW.from_torch(torch AdamW)` — gradients in one flat buffer the
    kernels add into, one cvh_adamw_multi launch per update — must follow the same loop with torch.optim.AdamW itself (fp32, two parameter
    groups with and without weight decay as optim/base_optim builds them)."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.layers import default_opts

    base = _load_swapped("xxs", "xx_small")
    for m in base.modules():
        if isinstance(m, cvnets_amd.layers.Dropout):
            m.p = 0.0
    crit = cvnets_amd.CrossEntropy(default_opts(**{"loss.classification.cross_entropy.label_smoothing": 0.1}))
    batches = _batches(4)

    class _Sched:
        def update_lr(self, optimizer, epoch, curr_iter):
            for g in optimizer.param_groups:
                g["lr"] = 1e-3 * (0.7 ** curr_iter)
            return optimizer

    def groups(model):
        decay = [p for p in model.parameters() if p.dim() > 1]
        no_decay = [p for p in model.parameters() if p.dim() <= 1]
        return [{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}]

    cvnets_amd.set_compute_dtype(torch.float32)
    res = []
    try:
        for fused in (False, True):
            model = copy.deepcopy(base)
            opt = torch.optim.AdamW(groups(model), lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
            if fused:
                opt = cvnets_amd.optim.AdamW.from_torch(opt, flat_grads=True)
                ops.set_inplace_param_grads(True)
                assert all(p.grad is not None for p in model.parameters())
                ptrs = [p.grad.data_ptr() for p in model.parameters()]
            scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=128.0)
            n, losses = engine_loop.train_iterations(model, crit, opt, _Sched(), scaler, batches, device="cuda:0", accum_freq=2, max_norm=5.0)
            assert n == 2
            if fused:
                assert ptrs == [p.grad.data_ptr() for p in model.parameters()]     # zero_grad(set_to_none=True) did not move a gradient
                assert opt._rebuilds == 0                                          # ... so the one-launch plan was built once
                assert set(opt.state_dict()["state"].keys()) == set(range(len(ptrs)))
            res.append(([p.detach().float().cpu().clone() for p in model.parameters()], losses))
    finally:
        ops.set_inplace_param_grads(False)
        cvnets_amd.set_compute_dtype(None)
    p0 = [p.detach().float().cpu() for p in base.parameters()]
    num = sum(float(((a - c) - (b - c)).double().pow(2).sum()) for a, b, c in zip(res[0][0], res[1][0], p0))
    den = sum(float((a - c).double().pow(2).sum()) for a, c in zip(res[0][0], p0))
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5     # AdamW normalises each gradient: compare the accumulated update
    assert all(abs(a - b) < 2e-3 for a, b in zip(res[0][1], res[1][1])), (res[0][1], res[1][1])


def test_engine_loop_under_the_reference_default_float16_autocast():
    """the shipped YAMLs leave `common.mixed_precision_dtype` at float16 (options/opts.py:126-131): the engine then wraps the step in
    torch.autocast(float16) with a GradScaler at its default scale (65536).  There are no float16 kernels on this path — such regions are
    computed in bfloat16 storage / fp32 accumulation, said once in a warning — so the reference configs run unmodified: same trajectory as
    the bfloat16 run up to the scale handling, finite gradients, the scaler's scale untouched."""
    import warnings
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.layers import default_opts

    base = _load_swapped("xxs", "xx_small")
    for m in base.modules():
        if isinstance(m, cvnets_amd.layers.Dropout):
            m.p = 0.0
    crit = cvnets_amd.CrossEntropy(default_opts(**{"loss.classification.cross_entropy.label_smoothing": 0.1}))
    batches = _batches(2)
    res = {}
    ops._WARNED_FP16 = False
    for dt in (torch.bfloat16, torch.float16):
        model = copy.deepcopy(base)
        opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        scaler = torch.amp.GradScaler("cuda", enabled=True)   # default init_scale 65536, as main_train.py:114 constructs it
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            n, losses = engine_loop.train_iterations(model, crit, opt, _ConstLR(), scaler, batches, device="cuda:0", amp_dtype=dt)
        assert n == 2 and scaler.get_scale() == 65536.0 and all(l == l for l in losses)
        assert all(torch.isfinite(p).all() for p in model.parameters())
        said = [x for x in w if "float16 autocast regions are computed in bfloat16" in str(x.message)]
        assert (len(said) == 1) == (dt == torch.float16), [str(x.message) for x in w]
        res[dt] = losses
    # both runs compute in bf16: identical kernels, identical losses
    assert res[torch.float16] == res[torch.bfloat16], res
    os.environ["CVH_STRICT_AUTOCAST"] = "1"
    try:
        with pytest.raises(RuntimeError), torch.autocast("cuda", dtype=torch.float16):
            ops.compute_dtype()
    finally:
        os.environ.pop("CVH_STRICT_AUTOCAST", None)
