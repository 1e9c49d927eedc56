#!/usr/bin/env python
"""Throughput of the widened SURVEY §8 rows (a12 ViT, a13 MobileViTv2, a14 CLIP) measured exactly like bench.py measures the
headline MobileViT-S config: synthetic data resident in HBM, zero_grad + fwd + loss + bwd + fused AdamW captured in one hipGraph,
K timed replays between HIP events.  One JSON line per model, with the roofline that bounds it:

  python tools/bench_models.py --models vit_base,mobilevitv2,clip [--steps 10 --warmup 3]

These are self-measurement lines for DESIGN.md / profiles/, not the driver's bench contract (bench.py)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
sys.path.insert(0, REPO)

HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15
# (fwd+bwd GFLOP/img, ideal fwd+bwd MB/img, binding roof)
ALGO = {"vit_base": (106.25, 190.9, "mfma"), "vit_tiny": (None, None, "mfma"), "mobilevitv2": (24.46, 362.6, "hbm"), "clip": (124.0, None, "mfma"),
        "mobilevit_s": (12.0, 176.1, "hbm"), "deeplabv3": (None, None, "hbm")}


def build(name, batch, dev):
    import cvnets_amd

    if name.endswith("_ckpt"):  # as shipped: gradient_checkpointing: true (config/classification/imagenet/vit.yaml:81, clip_vit.yaml:94)
        m, loss_of, desc = build(name[:-5], batch, dev)
        for mod in m.modules():
            if hasattr(mod, "gradient_checkpointing"):
                mod.gradient_checkpointing = True
        return m, loss_of, desc + ", gradient checkpointing"

    def seeded_caption_tokens(batch: int, ctx: int, vocab: int, seed: int) -> torch.Tensor:
        """synthetic captions: ids in [1, vocab-2], one EOT (= vocab-1) at a random position >= 4, padding (0) after it"""
        g = torch.Generator().manual_seed(seed)
        tok = torch.randint(1, vocab - 1, (batch, ctx), generator=g)
        eot = torch.randint(4, ctx, (batch,), generator=g)
        pos = torch.arange(ctx)[None, :]
        tok = torch.where(pos == eot[:, None], torch.full_like(tok, vocab - 1), tok)
        return torch.where(pos > eot[:, None], torch.zeros_like(tok), tok)

    if name in ("vit_base", "vit_tiny"):
        m = cvnets_amd.build_vit(name.split("_")[1], **{"model.classification.vit.dropout": 0.2 if name == "vit_base" else 0.0}).to(dev).train()
        x = torch.randn(batch, 3, 224, 224, device=dev)
        y = torch.randint(0, 1000, (batch,), device=dev)
        return m, lambda: cvnets_amd.ops.cross_entropy(m(x), y, 0.1), "224x224"
    if name == "mobilevitv2":
        m = cvnets_amd.build_mobilevit_v2(1.0).to(dev).train()
        x = torch.randn(batch, 3, 384, 384, device=dev)
        y = torch.randint(0, 1000, (batch,), device=dev)
        return m, lambda: cvnets_amd.ops.cross_entropy(m(x), y, 0.1), "384x384 width 1.0"
    if name == "mobilevit_s":
        m = cvnets_amd.build_mobilevit("small").to(dev).train()
        x = torch.randn(batch, 3, 256, 256, device=dev)
        y = torch.randint(0, 1000, (batch,), device=dev)
        return m, lambda: cvnets_amd.ops.cross_entropy(m(x), y, 0.1), "256x256"
    if name == "clip":
        from cvnets_amd.layers import default_opts
        m = cvnets_amd.build_clip().to(dev).train()
        loss_fn = cvnets_amd.ContrastiveLossClip(default_opts()).train()
        x = torch.randn(batch, 3, 224, 224, device=dev)
        tok = seeded_caption_tokens(batch, 77, 49408, seed=0).to(dev)
        return m, lambda: loss_fn(None, m({"image": x, "text": tok}))["total_loss"], "ViT-B/16 224x224 + 12x512 text, ctx 77"
    if name == "deeplabv3":
        # config/segmentation/pascal_voc/deeplabv3_mobilevit.yaml: MobileViT-S encoder at output stride 8, ASPP 512 channels, rates 12/24/36,
        # auxiliary head, 512 x 512 crops; loss of loss_fn/segmentation/cross_entropy.py (torch: adjacent to the path)
        import torch.nn.functional as F
        from cvnets_amd.layers import default_opts
        opts = default_opts(**{"model.classification.mit.mode": "small", "model.segmentation.output_stride": 8, "model.segmentation.n_classes": 21,
                               "model.segmentation.use_aux_head": True, "model.segmentation.use_level5_exp": False,
                               "model.segmentation.deeplabv3.aspp_out_channels": 512, "model.segmentation.deeplabv3.aspp_rates": (12, 24, 36)})
        m = cvnets_amd.build_deeplabv3_mobilevit(opts).to(dev).train()
        x = torch.randn(batch, 3, 512, 512, device=dev)
        y = torch.randint(0, 21, (batch, 512, 512), device=dev)

        def seg_loss():
            mask, aux = m(x)
            aux = F.interpolate(aux.float(), size=y.shape[-2:], mode="bilinear", align_corners=True)
            return F.cross_entropy(mask.float(), y, ignore_index=255) + 0.4 * F.cross_entropy(aux, y, ignore_index=255)
        return m, seg_loss, "DeepLabv3 + MobileViT-S, 512x512, output stride 8"
    raise SystemExit(f"unknown model {name}")


def run(name, batch, steps, warmup, dtype, use_graph):
    import cvnets_amd
    from cvnets_amd.ddp import DistributedDataParallel

    dev = torch.device("cuda", 0)
    cvnets_amd.set_compute_dtype(dtype)
    torch.manual_seed(1234)
    model, loss_of, desc = build(name, batch, dev)
    ddp = DistributedDataParallel(model, bucket_cap_mb=25.0, broadcast_buffers=False)
    ddp.hooks_enabled = False
    cvnets_amd.ops.set_inplace_param_grads(True)
    opt = cvnets_amd.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01)

    def one():
        ddp.zero_grad()
        loss = loss_of()
        loss.backward()
        opt.step(sync_hyperparameters=not capturing[0])
        return loss

    capturing = [False]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            loss = one()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph, err = None, None
    if use_graph:
        try:
            g = torch.cuda.CUDAGraph()
            capturing[0] = True
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = one()
            graph = g
        except Exception as e:
            err = f"{type(e).__name__}: {e}"[:200]
            torch.cuda.synchronize()
    step = graph.replay if graph is not None else one
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ips = batch * steps / wall
    gf, mb, bound = ALGO[name[:-5] if name.endswith("_ckpt") else name]
    out = {"model": name, "workload": f"{desc}, {batch} img/GPU", "dtype": str(dtype).split(".")[-1], "images_per_sec": round(ips, 1),
           "ms_per_step": round(wall * 1e3 / steps, 3), "gpu_ms_per_step_hip_events": round(e0.elapsed_time(e1) / steps, 3), "hipgraph": graph is not None,
           "loss": round(float(loss), 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "bound": bound}
    if gf:
        out["mfma_frac"] = round(gf * 1e9 * ips / MFMA_PEAK, 4)
        out["achieved_TFLOPs"] = round(gf * 1e9 * ips / 1e12, 1)
    if mb:
        out["hbm_frac"] = round(mb * 1e6 * ips / HBM_PEAK, 4)
    if err:
        out["hipgraph_error"] = err
    cvnets_amd.ops.set_inplace_param_grads(False)
    return out


def run_vbs(steps, warmup, dtype):
    """BASELINE config 5: MobileViTv2-1.0 under the variable-batch-sampler schedule around 384^2 — the (H = W, B) pairs of
    data/sampler/utils.py:13-67 image_batch_pairs(384, 384, 128, min 256, max 512, 5 scales, divisor 32), drawn per step with
    random.seed(epoch) as data/sampler/variable_batch_sampler.py:315-320 does.  One hipGraph per shape, captured on first use
    (capture time is inside the timed region: no per-shape warm-up credit)."""
    import random

    import cvnets_amd
    from cvnets_amd.ddp import DistributedDataParallel

    from cvnets_amd.schedule import image_batch_pairs, vbs_sequence

    pairs = [(h, b) for h, w, b in image_batch_pairs(384, 384, 128, 5, 32, 256, 512, 256, 512)]  # [(256, 288), (320, 184), (384, 128), (448, 94), (512, 72)]
    dev = torch.device("cuda", 0)
    cvnets_amd.set_compute_dtype(dtype)
    torch.manual_seed(1234)
    m = cvnets_amd.build_mobilevit_v2(1.0).to(dev).train()
    ddp = DistributedDataParallel(m, bucket_cap_mb=25.0, broadcast_buffers=False)
    ddp.hooks_enabled = False
    cvnets_amd.ops.set_inplace_param_grads(True)
    opt = cvnets_amd.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01)
    data = {hw: (torch.randn(b, 3, hw, hw, device=dev), torch.randint(0, 1000, (b,), device=dev)) for hw, b in pairs}
    graphs = {}

    def one(hw):
        x, y = data[hw]
        ddp.zero_grad()
        loss = cvnets_amd.ops.cross_entropy(m(x), y, 0.1)
        loss.backward()
        opt.step(sync_hyperparameters=False)
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ddp.zero_grad()
        cvnets_amd.ops.cross_entropy(m(data[256][0][:8]), data[256][1][:8], 0.1).backward()
        opt.step()  # builds the optimizer tables (eager, tiny batch)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    seq = vbs_sequence(pairs, warmup + steps, epoch=0)

    def step(hw):
        if hw not in graphs:
            s2 = torch.cuda.Stream()
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                one(hw)  # allocator warm-up for this shape
            torch.cuda.current_stream().wait_stream(s2)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                one(hw)
            graphs[hw] = g
        graphs[hw].replay()

    for hw, _ in seq[:warmup]:
        step(hw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    imgs = 0
    for hw, b in seq[warmup:]:
        step(hw)
        imgs += b
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    cvnets_amd.ops.set_inplace_param_grads(False)
    return {"model": "mobilevitv2_vbs", "workload": f"MobileViTv2-1.0, VBS schedule {pairs}, {steps} steps after {warmup} (graphs captured on first use of a shape)",
            "dtype": str(dtype).split(".")[-1], "images_per_sec": round(imgs / wall, 1), "ms_per_step": round(wall * 1e3 / steps, 3),
            "shapes_seen": sorted({hw for hw, _ in seq}), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "bound": "hbm"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="vit_base,mobilevitv2,clip")
    ap.add_argument("--batch", default="vit_base=128,vit_tiny=256,mobilevitv2=128,clip=128,mobilevit_s=128,deeplabv3=32")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    batches = dict(kv.split("=") for kv in a.batch.split(","))
    for name in a.models.split(","):
        if name == "mobilevitv2_vbs":
            try:
                print(json.dumps(run_vbs(a.steps, a.warmup, torch.bfloat16 if a.dtype == "bf16" else torch.float32)), flush=True)
            except Exception as e:
                print(json.dumps({"model": name, "error": f"{type(e).__name__}: {e}"[:400]}), flush=True)
            continue
        try:
            print(json.dumps(run(name, int(batches[name[:-5] if name.endswith("_ckpt") else name]), a.steps, a.warmup, torch.bfloat16 if a.dtype == "bf16" else torch.float32, not a.no_graph)), flush=True)
        except Exception as e:  # keep going: one model failing must not hide the others
            print(json.dumps({"model": name, "error": f"{type(e).__name__}: {e}"[:400]}), flush=True)
        torch.cuda.empty_cache()
