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
