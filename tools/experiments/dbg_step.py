import os, sys, copy
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, cvnets_amd
from cvnets_amd.layers import default_opts
from cvnets_amd.optim import AdamW
from util import l2_err
cvnets_amd.set_compute_dtype(torch.float32)
torch.manual_seed(0)
opts = default_opts(**{"model.classification.mit.mode": "xx_small", "model.classification.mit.dropout": 0.0, "model.classification.classifier_dropout": 0.0})
net = cvnets_amd.MobileViT(opts).cuda().train()
ref = copy.deepcopy(net)
x = torch.randn(4, 3, 32, 32, device="cuda")
for lr in (5e-2, 1e-3):
    a, b = copy.deepcopy(net), copy.deepcopy(ref)
    oa, ob = AdamW(a.parameters(), lr=lr), torch.optim.AdamW(b.parameters(), lr=lr)
    for m, o in ((a, oa), (b, ob)):
        o.zero_grad(set_to_none=False); m(x).square().mean().backward(); 
    gerr = max(l2_err(p.grad, q.grad) for p, q in zip(a.parameters(), b.parameters()))
    oa.step(); ob.step()
    perr = [(k, l2_err(p, q)) for (k, p), q in zip(a.named_parameters(), b.parameters())]
    perr.sort(key=lambda t: -t[1])
    with torch.no_grad():
        ya, yb = a(x), b(x)
    print("lr", lr, "grad err", gerr, "worst param err", perr[:3], "fwd err", l2_err(ya, yb), float(ya.abs().mean()), float(yb.abs().mean()))
