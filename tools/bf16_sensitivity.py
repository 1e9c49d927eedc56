"""How sensitive is a bf16-rounded evaluation of MobileViT to fp32 round-off?  (CPU only; test infrastructure: uses oracle/.)
The fp32 oracle and its rounding-points evaluation (oracle/bf16_points.py) are run on the same weights and batch, then again on an input
perturbed by 1e-7 .. 1e-5 RELATIVE (one fp32 ulp and up) — a stand-in for "the same mathematics with another fp32 summation order".
    python tools/bf16_sensitivity.py small 16 256      # MobileViT-S, the fixture shape of tests/test_bf16_parity_gpu.py
Measured (EPYC, this container): the fp32 oracle moves by 1.4e-6 (logits) / 4.8e-6 (gradients) under a 1e-7 perturbation, the rounded
evaluation by 1.5e-2 / 5.1e-2 — nearly its whole distance from fp32 (2.2e-2 / 7.5e-2).  Rounding decisions flip and the flips propagate: two
correct bf16 implementations cannot agree with each other (or with this emulation) to 1e-3, only to the bf16 noise level itself."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import bf16_points, mobilevit_oracle as orc
from oracle.weights import seeded_input, seeded_labels, seeded_state_dict
gold=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
mode, B, res = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sd = seeded_state_dict(json.load(open(os.path.join(gold, f"mobilevit_{mode}_keys.json"))), seed=0)
x, y = seeded_input((B, 3, res, res), seed=1), seeded_labels(B, 1000, seed=1)
def rel(a,b): return float((a.double()-b.double()).norm()/b.double().norm())
def grel(ga,gb):
    num=sum(float((ga[k].double()-gb[k].double()).pow(2).sum()) for k in gb); den=sum(float(gb[k].double().pow(2).sum()) for k in gb); return (num/den)**0.5
torch.set_num_threads(8)
t=time.time(); l32,_,g32,_ = orc.train_step(sd,x,y,mode=mode); print('fp32',time.time()-t,flush=True)
l_a,_,g_a,_ = bf16_points.train_step(sd,x,y,mode=mode)
# the same emulation with the batch in reversed order: identical mathematics, every batch reduction (BatchNorm statistics, weight gradients) sums in another order
perm = torch.arange(B-1,-1,-1)
l_b,_,g_b,_ = bf16_points.train_step(sd,x[perm].contiguous(),y[perm].contiguous(),mode=mode)
l_b = l_b[perm]
print('emulation vs fp32: logits %.3e grads %.3e' % (rel(l_a,l32), grel(g_a,g32)))
print('emulation (batch reversed) vs fp32: logits %.3e grads %.3e' % (rel(l_b,l32), grel(g_b,g32)))
print('emulation vs emulation (batch reversed): logits %.3e grads %.3e' % (rel(l_a,l_b), grel(g_a,g_b)))
l32b,_,g32b,_ = orc.train_step(sd,x[perm].contiguous(),y[perm].contiguous(),mode=mode)
print('fp32 vs fp32 (batch reversed): logits %.3e grads %.3e' % (rel(l32b[perm],l32), grel(g32b,g32)))
g = torch.Generator().manual_seed(5)
for eps in (1e-7, 1e-6, 1e-5):
    xp = x * (1 + eps * torch.randn(x.shape, generator=g))
    l_p,_,g_p,_ = bf16_points.train_step(sd,xp,y,mode=mode)
    l32p,_,g32p,_ = orc.train_step(sd,xp,y,mode=mode)
    print('input perturbed by %.0e relative: emulation moves logits %.3e grads %.3e | fp32 oracle moves logits %.3e grads %.3e' % (eps, rel(l_p,l_a), grel(g_p,g_a), rel(l32p,l32), grel(g32p,g32)), flush=True)
