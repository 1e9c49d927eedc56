#!/usr/bin/env python
"""Repeats the fp32 MobileViTv2 golden cases N times in ONE process and reports every run whose error margins leave the test's
tolerances (one unexplained single failure of such a case was seen in ~25 full-suite runs in round 2).

    CVH_ASYNC_DW=1 python tools/stress_v2.py 300      # parameter-gradient side stream on (default)
    CVH_ASYNC_DW=0 python tools/stress_v2.py 300      # everything on one stream

Prints one line per case: runs, failures, worst logits / loss / per-tensor gradient-norm deviation seen."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "ml-cvnets_amd"), os.path.join(REPO, "tests")]
import test_mobilevitv2_gpu as t  # noqa: E402
from util import l2_err  # noqa: E402
from oracle.weights import seeded_input, seeded_labels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name, wm, batch, hw in t.CASES:
    gold = np.load(os.path.join(t.GOLD, name + ".npz"))
    model, sd = t._build(wm, torch.float32)
    x, y = seeded_input((batch, 3) + hw, seed=1).cuda(), seeded_labels(batch, 1000, seed=1).cuda()
    names = [str(k) for k in gold["grad_names"]]
    gref = torch.from_numpy(gold["grad_norm"])
    worst = [0.0, 0.0, 0.0, 0.0]
    fails = []
    for it in range(n):
        model.load_state_dict(sd)  # BatchNorm running statistics back to the start
        logits, loss, grads = t._step(model, x, y)
        e_l = l2_err(logits, torch.from_numpy(gold["logits_train"]))
        e_loss = abs(loss - float(gold["loss"]))
        gn = torch.tensor([grads[k].norm().item() for k in names], dtype=torch.float64)
        e_g = float(((gn - gref).abs() / (gref + 1e-2 * gref.max())).max())
        e_full = max(l2_err(grads[k[6:]], torch.from_numpy(gold[k])) for k in gold.files if k.startswith("grad::"))
        worst = [max(a, b) for a, b in zip(worst, (e_l, e_loss, e_g, e_full))]
        if e_l >= 1e-4 or e_loss >= 1e-4 or e_g >= 2e-3 or e_full >= 2e-3 or not np.isfinite(loss):
            fails.append((it, e_l, e_loss, e_g, e_full))
    print(f"{name}: runs {n}, failures {len(fails)}, worst logits {worst[0]:.2e} loss {worst[1]:.2e} grad-norm {worst[2]:.2e} full-grad {worst[3]:.2e}"
          f" (tolerances 1e-4 / 1e-4 / 2e-3 / 2e-3) ASYNC_DW={os.environ.get('CVH_ASYNC_DW', '1')}", flush=True)
    for f in fails[:5]:
        print("   FAIL run %d: logits %.2e loss %.2e grad-norm %.2e full-grad %.2e" % f, flush=True)
