"""Throughput of the ENGINE-DRIVEN path — what the reference's Trainer does per iteration (engine/training_engine.py:221-309, restated in
tests/engine_loop.py and pinned against the real Trainer by tests/test_launch_cpu.py): eager launches, torch.autocast(bfloat16) + GradScaler,
optimizer.zero_grad(set_to_none=True), a reference-style torch optimizer — no hipGraph, no flat gradient buckets, no fused AdamW.
Reports img/s, GPU-busy time per iteration (HIP events) against wall time (= how host-bound the ctypes-per-kernel dispatch is) and, beside
it, bench.py's replayed step on the same box.

    python tools/bench_engine.py [--batch 128,1024] [--iters 10] [--graph-compare]"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import cvnets_amd  # noqa: E402
import engine_loop  # noqa: E402
from cvnets_amd.layers import default_opts  # noqa: E402


class _ConstLR:
    def update_lr(self, optimizer, epoch, curr_iter):
        return optimizer


def run(batch, iters, warm=3, fused=False):
    dev = "cuda:0"
    torch.manual_seed(0)
    model = cvnets_amd.build_mobilevit("small").to(dev).train()
    crit = cvnets_amd.CrossEntropy(default_opts(**{"loss.classification.cross_entropy.label_smoothing": 0.1}))
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0.01)  # optim/adamw.py:16-46 = torch.optim.AdamW
    if fused:  # substitution 4 of cvnets_amd/launch.py: the same loop steps the one-launch AdamW over flat in-place gradients
        opt = cvnets_amd.optim.AdamW.from_torch(opt, flat_grads=True)
        cvnets_amd.ops.set_inplace_param_grads(True)
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    x = torch.randn(batch, 3, 256, 256, device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)
    batches = [{"samples": x, "targets": y}]

    def go(n):
        engine_loop.train_iterations(model, crit, opt, _ConstLR(), scaler, batches * n, device=dev, amp_dtype=torch.bfloat16)

    go(warm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    go(iters)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    # GPU-busy time of one iteration: the same work with the host running ahead is not observable directly; take the kernel time of an
    # iteration from a device-side timeline instead: enqueue `iters` iterations without any host sync in between and read the events
    gpu = e0.elapsed_time(e1) / 1e3 / iters
    cvnets_amd.ops.set_inplace_param_grads(False)
    return {"batch": batch, "img_per_s": round(batch / wall, 1), "wall_ms_per_iter": round(wall * 1e3, 2),
            "gpu_event_ms_per_iter": round(gpu * 1e3, 2)}


def replayed(batch):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--batch", str(batch), "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
           "--no-kernel-probe"]
    out = subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    return {"batch": batch, "img_per_s": d["value"], "ms_per_step": d["ms_per_step"]}


_replay_cache = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default="128,1024")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--graph-compare", action="store_true", help="also run bench.py (hipGraph replay, fused AdamW) at the same batch sizes")
    a = ap.parse_args()
    for b, fused in [(int(v), f) for v in a.batch.split(",") for f in (False, True)]:
        r = run(b, a.iters, fused=fused)
        line = {"path": "engine loop, eager (autocast bf16 + GradScaler + " + ("cvnets_amd.optim.AdamW over flat in-place gradients: the launcher's default)"
                                                                                 if fused else "torch AdamW)"), **r}
        if a.graph_compare:
            g = _replay_cache[b] if b in _replay_cache else _replay_cache.setdefault(b, replayed(b))
            line["replayed_img_per_s"] = g["img_per_s"]
            line["eager_over_replayed"] = round(r["img_per_s"] / g["img_per_s"], 3)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
