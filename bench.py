#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec (fwd+bwd) of a MobileViT-S 256x256 bf16 training step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = zero-grad + forward + label-smoothed cross-entropy + backward (+ gradient all-reduce when N > 1) + AdamW update
of MobileViT-S (random init, dropout as in config/classification/imagenet/mobilevit.yaml) on a synthetic batch of
128 images/GPU that is already resident in HBM.  The step is captured once into a hipGraph and replayed.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# algorithmic work per image of MobileViT-S @256^2, fwd+bwd
ALGO_BYTES_PER_IMG = 176.1e6   # fused-ideal bf16 activation traffic  2 B x (5 x sum|op outputs| + |input|)
ALGO_FLOP_PER_IMG = 12.00e9
HBM_PEAK = 8.0e12              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_BF16 = 2.5e15


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (mobilevit.yaml train_batch_size0)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--mode", default="small")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true", help="torch.optim.AdamW(fused=True) instead of the one-launch cvh_adamw_multi step")
    ap.add_argument("--torch-loss", action="store_true", help="F.cross_entropy instead of the cvh_ce_* kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path, fp32 — the reference refuses AMP without CUDA,
    engine/utils.py:31-32) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict
    import cvnets_amd

    cores = min(os.cpu_count() or 1, 32)  # PyTorch's CPU conv/GEMM stop scaling (and regress) past ~32 threads at this size
    torch.set_num_threads(cores)
    m = cvnets_amd.build_mobilevit(args.mode)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    x = seeded_input((args.cpu_batch, 3, args.res, args.res), seed=1)
    y = seeded_labels(args.cpu_batch, 1000, seed=1)
    orc.train_step(sd, x, y, mode=args.mode)  # warm-up
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        orc.train_step(sd, x, y, mode=args.mode)
    dt = (time.perf_counter() - t0) / args.cpu_steps
    return {"value": round(args.cpu_batch / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 fwd+loss+bwd, batch {args.cpu_batch} @ {args.res}x{args.res}, {args.cpu_steps} timed steps after 1 warm-up"}


def dominant_kernel_probe(dtype, batch):
    """Times the model's largest single kernel class live with HIP events on the launch stream: the pointwise-conv
    implicit GEMM of InvertedResidual layer_2.0.exp_1x1 (M = B*128*128 pixels, K = 32 -> N = 128).
    Algorithmic bytes per launch = M*(K+N)*2 (bf16 read of the input map + write of the output map); weights negligible."""
    from cvnets_amd import ops
    M, K, N = batch * 128 * 128, 32, 128
    x = torch.randn(M, K, device="cuda").to(dtype)
    w = torch.randn(N, K, device="cuda") * 0.1
    wp = ops.pack_weight(w, dtype, 0)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    st = torch.cuda.current_stream()
    for _ in range(3):
        ops._conv_gemm(x, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(st)
    for _ in range(reps):
        ops._conv_gemm(x, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N)
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = M * (K + N) * x.element_size()
    return {"kernel": "conv_gemm_kernel (pointwise 32->128 @128x128, layer_2.0.exp_1x1)", "avg_ms": round(ms, 4),
            "algorithmic_bytes": nbytes, "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # CVH_BENCH_SHARE_GPU=1 + CVH_DIST_BACKEND=gloo: developer rehearsal of the N > 1 control flow on a 1-GPU box (all ranks on cuda:0,
    # host-staged collectives); the real runs use one GPU per rank and RCCL.
    dev_index = 0 if os.environ.get("CVH_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import cvnets_amd
    from cvnets_amd.ddp import DistributedDataParallel, distributed_init

    if world > 1:
        distributed_init(os.environ.get("CVH_DIST_BACKEND", "nccl"), dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cvnets_amd.set_compute_dtype(dtype)
    torch.manual_seed(1234 + rank)
    model = cvnets_amd.build_mobilevit(args.mode).to(dev).train()
    # gradients live in flat fp32 buckets (one zero-fill per step, one RCCL message per bucket); with world == 1 the wrapper
    # only provides the flat storage.  Backward kernels add parameter gradients straight into those buffers.
    ddp = DistributedDataParallel(model, bucket_cap_mb=25.0, broadcast_buffers=False)
    ddp.hooks_enabled = False  # hipGraph replay does not run autograd hooks: buckets are reduced right after the replay
    cvnets_amd.ops.set_inplace_param_grads(True)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = None
    if not args.no_optimizer:
        if args.torch_optimizer:
            opt = torch.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01, fused=True, capturable=True)
        else:  # every parameter stepped by one kernel launch, rates and step counter on the device
            opt = cvnets_amd.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01)

    x = torch.randn(args.batch, 3, args.res, args.res, device=dev)
    y = torch.randint(0, 1000, (args.batch,), device=dev)

    def zero_grads():
        ddp.zero_grad()

    def fwd_bwd():
        logits = model(x)
        if args.torch_loss:
            loss = F.cross_entropy(logits.float(), y, label_smoothing=0.1)
        else:  # loss_fn/classification/cross_entropy.py:65-92 in one HIP kernel per direction
            loss = cvnets_amd.ops.cross_entropy(logits, y, 0.1)
        loss.backward()
        return loss

    use_graph = not args.no_graph
    graph = None
    static_loss = None

    def step():
        nonlocal static_loss
        if graph is not None:
            graph.replay()
            if world > 1:
                ddp.allreduce_flat()
        else:
            zero_grads()
            static_loss = fwd_bwd()
            if world > 1:
                ddp.allreduce_flat()
        if opt is not None and (graph is None or world > 1):
            opt_step()

    # eager warm-up (also creates every lazily-built tensor before capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            zero_grads()
            static_loss = fwd_bwd()
            if world > 1:
                ddp.allreduce_flat()  # replicas stay identical through the warm-up as well
            if opt is not None:
                opt.step()
    torch.cuda.current_stream().wait_stream(side)

    def opt_step():
        if args.torch_optimizer:
            opt.step()
        else:
            opt.step(sync_hyperparameters=False)  # constant rate in this benchmark: the device-side table was filled by the warm-up steps
    torch.cuda.synchronize()

    graph_err = None
    if use_graph:
        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events while we capture; only this thread's calls belong to the graph
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                ddp.zero_grad()
                static_loss = fwd_bwd()
                if opt is not None and world == 1:
                    opt_step()
            graph = g
        except Exception as e:  # pragma: no cover - reported in the JSON line
            graph_err = f"{type(e).__name__}: {e}"[:300]
            graph = None
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for _ in range(args.steps):
        step()
    e1.record(st)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    gpu_ms_per_step = e0.elapsed_time(e1) / args.steps
    loss_val = float(static_loss.item())

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / wall
        per_gpu = value / world
        t_img = 1.0 / per_gpu
        frac_hbm = (ALGO_BYTES_PER_IMG / HBM_PEAK) / t_img
        frac_mfma = (ALGO_FLOP_PER_IMG / MFMA_PEAK_BF16) / t_img
        roofline = {
            "bound": "hbm",
            "kernel": "whole train step (all kernels; MobileViT-S is HBM-bound at AI = 68 FLOP/B, SURVEY.md 8d)",
            "achieved": round(ALGO_BYTES_PER_IMG * per_gpu / 1e9, 1),
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": round(frac_hbm, 4),
            "traffic": None,
            "mfma_frac": round(frac_mfma, 4),
            "algorithmic_bytes_per_image": ALGO_BYTES_PER_IMG,
            "gpu_ms_per_step_hip_events": round(gpu_ms_per_step, 3),
        }
        pmc = None
        try:  # HBM traffic cannot be counted from inside the process: it comes from the committed rocprofv3 PMC passes of this command
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
            if args.batch == pmc["step"]["images"] and args.mode == "small" and args.res == 256 and args.dtype == "bf16":
                roofline["traffic"] = pmc["step"]["total_bytes"] / pmc["step"]["images"]  # bytes per image, like algorithmic_bytes_per_image
                roofline["traffic_source"] = pmc["source"]
        except Exception:
            pmc = None
        try:
            roofline["dominant_kernel"] = dominant_kernel_probe(dtype, args.batch)
            if pmc is not None and args.batch == 128:
                roofline["dominant_kernel"]["traffic"] = pmc["dominant_kernel"]["read_bytes_per_launch"] + pmc["dominant_kernel"]["write_bytes_per_launch"]
            roofline["dominant_kernel"]["frac"] = round(roofline["dominant_kernel"]["achieved_GBps"] * 1e9 / HBM_PEAK, 4)
        except Exception as e:  # pragma: no cover
            roofline["dominant_kernel"] = {"error": str(e)[:200]}
        out = {
            "metric": "images/sec (fwd+bwd) MobileViT-S 256x256 bf16" if (args.mode == "small" and args.res == 256 and args.dtype == "bf16")
            else f"images/sec (fwd+bwd) MobileViT-{args.mode} {args.res}x{args.res} {args.dtype}",
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (randn images, random-init weights)",
            "config": {"workload": f"MobileViT-{args.mode} {args.res}x{args.res}, {args.batch} img/GPU, global batch {args.batch * world}",
                       "step": "zero_grad+fwd+CE(ls=0.1)+bwd" + ("+allreduce" if world > 1 else "") + ("" if opt is None else ("+AdamW(torch fused)" if args.torch_optimizer else "+AdamW(cvh_adamw_multi)")),
                       "parallelism": f"dp{world}", "hipgraph": graph is not None, "dropout": 0.1, "loss": round(loss_val, 4)},
            "images_per_sec_per_gpu": round(per_gpu, 2),
            "roofline": roofline,
        }
        if graph_err:
            out["config"]["hipgraph_error"] = graph_err
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
