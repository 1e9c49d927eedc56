import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-cvnets_amd"))
import cvnets_amd
torch.manual_seed(0)
m = cvnets_amd.build_mobilevit("small", **{"model.classification.mit.dropout": 0.1}).cuda().train()
opt = cvnets_amd.optim.AdamW(m.parameters(), lr=1e-3)
x = torch.randn(32, 3, 256, 256, device="cuda"); y = torch.randint(0, 1000, (32,), device="cuda")
cvnets_amd.set_compute_dtype(torch.bfloat16)
def step():
    opt.zero_grad()
    loss = cvnets_amd.ops.cross_entropy(m(x), y, 0.1)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
cnt = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        st = [f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}" for fr in traceback.extract_stack()[:-1] if "cvnets_amd" in fr.filename or "autograd" in fr.filename][-3:]
        cnt[(name, tuple(st))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for n in ("zeros", "ones", "full", "zeros_like", "ones_like"):
    wrap(torch, n)
for n in ("zero_", "fill_", "new_zeros"):
    wrap(torch.Tensor, n)
step()
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k[0], " <- ", " | ".join(k[1])[:300])
