"""SURVEY §8f "next" rows built so far, to the same parity bar as the path itself:
  row 1 — fused AdamW (+EMA) step over all parameters in one launch (cvnets_amd.optim.AdamW vs torch.optim.AdamW, the class the
          reference's optim/adamw.py wraps; EMA vs cvnets/misc/averaging_utils.py:43-55 restated inline),
  row 2 — label-smoothed cross-entropy kernel (cvnets_amd.CrossEntropy vs F.cross_entropy as called by
          loss_fn/classification/cross_entropy.py:65-92)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(8, 1000, 0.1, False), (128, 1000, 0.1, False), (5, 10, 0.0, False), (33, 257, 0.2, True)])
def test_cross_entropy_kernel(dtype, cfg):
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    N, M, eps, with_ignored = cfg
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(N, M, generator=g) * 3).to(dtype)
    labels = torch.randint(0, M, (N,), generator=g)
    if with_ignored:
        labels[::4] = -1
    crit = cvnets_amd.CrossEntropy(default_opts(**{"loss.classification.cross_entropy.label_smoothing": eps,
                                                   "loss.classification.cross_entropy.ignore_index": -1})).train()
    lr = logits.float().clone().requires_grad_()
    ref = F.cross_entropy(lr, labels, ignore_index=-1, label_smoothing=eps)
    ref.backward()
    lg = logits.cuda().requires_grad_()
    loss = crit(None, lg, labels.cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-5 * max(1.0, abs(float(ref.detach())))
    assert l2_err(lg.grad.float().cpu(), lr.grad) < (1e-5 if dtype == torch.float32 else 8e-3)
    crit.eval()  # evaluation: no smoothing (cross_entropy.py:82)
    assert abs(float(crit(None, {"logits": lg.detach()}, labels.cuda())) - float(F.cross_entropy(logits.float(), labels, ignore_index=-1))) < 2e-5 * 10


def test_fused_adamw_and_ema_match_torch():
    import cvnets_amd
    from cvnets_amd.optim import AdamW, EMABuffers

    torch.manual_seed(0)
    # (no conv bias: a bias in front of BatchNorm has a mathematically zero gradient, and Adam turns its round-off into +-lr steps)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 10)).cuda()
    ref = copy.deepcopy(net)
    ema, ema_ref = copy.deepcopy(net), copy.deepcopy(net)

    def groups(m):  # reference grouping: no weight decay for 1-D tensors (cvnets/misc/common.py:136-160), own lr multiplier per group
        decay = [p for p in m.parameters() if p.dim() > 1]
        no_decay = [p for p in m.parameters() if p.dim() <= 1]
        return [{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0, "lr": 3e-3}]

    opt = AdamW(groups(net), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, ema=(net, ema), ema_momentum=0.01)
    emab = EMABuffers(net, ema, momentum=0.01)
    opt_ref = torch.optim.AdamW(groups(ref), lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    x = torch.randn(4, 3, 8, 8, device="cuda")
    for it in range(4):
        for m, o in ((net, opt), (ref, opt_ref)):
            o.zero_grad(set_to_none=False)
            m(x).square().mean().backward()
        if it == 2:  # a scheduler rewrites the rates between steps
            for o in (opt, opt_ref):
                o.param_groups[0]["lr"] = 5e-4
        opt.step()
        emab.update()
        opt_ref.step()
        with torch.no_grad():  # averaging_utils.py:47-55
            msd = ref.state_dict()
            for k, ev in ema_ref.state_dict().items():
                ev.copy_((ev * (1.0 - 0.01)) + (0.01 * msd[k].detach()))
    for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert l2_err(a.float().cpu(), b.float().cpu()) < 2e-6, k
    for (k, a), (_, b) in zip(ema.state_dict().items(), ema_ref.state_dict().items()):
        assert l2_err(a.float().cpu(), b.float().cpu()) < 2e-6, k


def test_fused_adamw_inside_a_captured_step():
    """the whole optimizer step is hipGraph-capturable (device-side step counter and rates): replays advance the state exactly like
    eager torch.optim.AdamW steps on the same (static) gradient."""
    from cvnets_amd.optim import AdamW

    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(1000, device="cuda"))
    w_ref = torch.nn.Parameter(w.detach().clone())
    gr = torch.randn(1000, device="cuda")
    w.grad, w_ref.grad = gr.clone(), gr.clone()
    opt, opt_ref = AdamW([w], lr=1e-2, weight_decay=0.1), torch.optim.AdamW([w_ref], lr=1e-2, weight_decay=0.1)
    opt.step()  # eager warm-up step builds the pointer table
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            opt.step(sync_hyperparameters=False)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    for _ in range(4):  # 1 eager + 3 replays (capture records, it does not execute)
        opt_ref.step()
    torch.cuda.synchronize()
    assert float(opt._plan["step"].item()) == 4.0
    assert l2_err(w.detach().cpu(), w_ref.detach().cpu()) < 2e-6


# ------------------------------------------------------------------------------------------------
# advisor findings of round 1 (packed-weight cache, optimizer state, set_to_none gradients, dropout seed)
# ------------------------------------------------------------------------------------------------
def _tiny_mobilevit(seed=0):
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    torch.manual_seed(seed)
    opts = default_opts(**{"model.classification.mit.mode": "xx_small", "model.classification.mit.dropout": 0.0,
                           "model.classification.classifier_dropout": 0.0})
    return cvnets_amd.MobileViT(opts).cuda()


def test_forward_after_fused_step_uses_fresh_weights():
    """The fused AdamW / EMA kernels write parameters through raw pointers (no autograd version bump): the next forward — train mode of
    the stepped model, eval mode of the EMA model — must run on the UPDATED weights, not on a packed copy of the old ones."""
    import cvnets_amd
    from cvnets_amd.optim import AdamW

    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        net = _tiny_mobilevit().train()
        ema = copy.deepcopy(net).eval()
        x = torch.randn(4, 3, 32, 32, device="cuda")
        with torch.no_grad():
            e0 = ema(x).clone()  # fills the EMA model's packed-weight cache
            net(x)               # ... and the model's
        opt = AdamW(net.parameters(), lr=5e-2, ema=(net, ema), ema_momentum=0.5)
        opt.zero_grad(set_to_none=False)
        net(x).square().mean().backward()
        opt.step()
        # references: fresh copies of the UPDATED parameters (new tensors, nothing cached for them)
        net_ref, ema_ref = copy.deepcopy(net), copy.deepcopy(ema)
        with torch.no_grad():
            y, y_ref = net(x), net_ref(x)      # train-mode forward after the step
            e1, e1_ref = ema(x), ema_ref(x)    # eval-mode forward of the EMA model after its parameters moved
        assert l2_err(y, y_ref) < 1e-3  # (stale packs give >= 1e-2; run-to-run round-off through train-mode BatchNorm at batch 4 is ~4e-5)
        assert l2_err(e1, e1_ref) < 1e-3
        assert l2_err(e1, e0) > 1e-3, "the EMA model's forward did not change although its parameters did: stale packed weights"
        # standalone layer (no model-level forward): its per-call packed-weight cache must notice the raw-pointer update as well
        from cvnets_amd import ops
        lin_w = torch.randn(16, 8, device="cuda").requires_grad_()
        xin = torch.randn(4, 8, device="cuda")
        y0 = ops.linear(xin, lin_w).clone()
        o2 = AdamW([lin_w], lr=0.5)
        lin_w.grad = torch.ones_like(lin_w)
        o2.step()
        assert l2_err(ops.linear(xin, lin_w), xin @ lin_w.detach().t()) < 1e-5 and l2_err(ops.linear(xin, lin_w), y0) > 1e-2
    finally:
        cvnets_amd.set_compute_dtype(None)


def test_fused_adamw_state_dict_roundtrip_and_set_to_none():
    """checkpoint / resume (the reference saves optimizer.state_dict()) and the default zero_grad(set_to_none=True) of torch >= 2"""
    from cvnets_amd.optim import AdamW

    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4)).cuda()
    unused = torch.nn.Parameter(torch.randn(5, device="cuda"))  # never receives a gradient: torch skips it, so must we
    ref, unused_ref = copy.deepcopy(net), torch.nn.Parameter(unused.detach().clone())
    opt = AdamW(list(net.parameters()) + [unused], lr=1e-2, weight_decay=0.1)
    opt_ref = torch.optim.AdamW(list(ref.parameters()) + [unused_ref], lr=1e-2, weight_decay=0.1)
    x = torch.randn(8, 12, device="cuda")

    def run(m, o, steps):
        for _ in range(steps):
            o.zero_grad()  # set_to_none=True: fresh gradient tensors every step
            m(x).square().mean().backward()
            o.step()

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run(net, opt, 3)
        run(ref, opt_ref, 3)
    assert torch.equal(unused.detach(), unused_ref.detach())
    sd = opt.state_dict()
    sd_ref = opt_ref.state_dict()
    assert set(sd["state"]) == set(sd_ref["state"])  # the unused parameter has no state in either
    for k, st in sd["state"].items():
        assert float(st["step"]) == float(sd_ref["state"][k]["step"]) == 3.0
        assert l2_err(st["exp_avg"].cpu(), sd_ref["state"][k]["exp_avg"].cpu()) < 1e-4
        assert l2_err(st["exp_avg_sq"].cpu(), sd_ref["state"][k]["exp_avg_sq"].cpu()) < 1e-4
    # resume in a fresh optimizer, continue, compare with the uninterrupted torch run
    net2 = copy.deepcopy(net)
    opt2 = AdamW(list(net2.parameters()) + [torch.nn.Parameter(unused.detach().clone())], lr=1e-2, weight_decay=0.1)
    opt2.load_state_dict(sd)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run(net2, opt2, 2)
        run(ref, opt_ref, 2)
    for a, b in zip(net2.parameters(), ref.parameters()):
        assert l2_err(a.detach().cpu(), b.detach().cpu()) < 2e-5


def test_dropout_mask_survives_a_second_forward():
    """forward(a), forward(b), backward(a): the mask regenerated in a's backward must be a's, not b's (per-forward seed snapshot)"""
    import cvnets_amd
    from cvnets_amd import ops

    x = torch.randn(64, 128, device="cuda", requires_grad=True)
    w = torch.randn(128, 128, device="cuda") * 0.1
    ops.advance_dropout_seed(x.device)
    ya = ops.linear(x, w, drop_p=0.5)
    keep_a = (ya.detach() != 0)
    ops.advance_dropout_seed(x.device)      # a second training forward starts (different masks)
    yb = ops.linear(x.detach(), w, drop_p=0.5)
    assert (keep_a != (yb != 0)).any()
    ya.backward(torch.ones_like(ya))
    # d/dx sum(dropout(x W^T)) = (mask * 2) W: compare with the mask observed in a's forward
    ref = (keep_a.float() * 2.0) @ w
    assert l2_err(x.grad, ref) < 1e-4
