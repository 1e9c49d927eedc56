#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec (fwd+bwd) of a MobileViT-S 256x256 bf16 training step on MI355X.

    python bench.py                      # 1 GPU, 1024 images per step (BASELINE configs[1]: "global batch 1024, 1xMI355X")
    python bench.py --gpus N             # spawns N ranks itself (one per GPU, RCCL), like the reference's main_train.py:261-265
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = zero-grad + forward + label-smoothed cross-entropy + backward (+ gradient all-reduce when N > 1) + AdamW update
of MobileViT-S (random init, dropout as in config/classification/imagenet/mobilevit.yaml) on a synthetic batch of `--batch` images
per GPU (weak scaling) that is already resident in HBM.  The step is captured once into a hipGraph and replayed.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import socket
import statistics
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per step (BASELINE configs[1]: 1024 on one MI355X; "
                    "the reference recipe's 128/GPU of mobilevit.yaml is --batch 128)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--mode", default="small")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true", help="torch.optim.AdamW(fused=True) instead of the one-launch cvh_adamw_multi step")
    ap.add_argument("--torch-loss", action="store_true", help="F.cross_entropy instead of the cvh_ce_* kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-probe", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--dry-run", action="store_true", help="control-flow rehearsal on CPU (gloo, no model, no HIP): launch / rendezvous / "
                    "timing / reduction / JSON only — NOT a measurement (tests/test_bench_cpu.py)")
    return ap.parse_args(argv)


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path, fp32 — the reference refuses AMP without CUDA, engine/utils.py:31-32) timed on
    this box's host cores, SURVEY.md §8d protocol: batch 16 at 256x256, 1 warm-up + >= 3 timed fwd+loss+bwd steps, best and median."""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict
    import cvnets_amd

    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 64)  # PyTorch's CPU conv / GEMM stop scaling (and regress) well before the full socket count at this size
    torch.set_num_threads(cores)
    m = cvnets_amd.build_mobilevit(args.mode)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    x = seeded_input((args.cpu_batch, 3, args.res, args.res), seed=1)
    y = seeded_labels(args.cpu_batch, 1000, seed=1)
    orc.train_step(sd, x, y, mode=args.mode)  # warm-up
    times = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        orc.train_step(sd, x, y, mode=args.mode)
        times.append(time.perf_counter() - t0)
    best, med = min(times), statistics.median(times)
    return {"value": round(args.cpu_batch / best, 3), "median": round(args.cpu_batch / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "host_cpus": ncpu, "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"oracle fp32 fwd+loss+bwd, batch {args.cpu_batch} @ {args.res}x{args.res}, {args.cpu_steps} timed steps after 1 warm-up "
                      f"(value = best, median also given)"}


# ------------------------------------------------------------------------------------------------
# dominant kernel (by total time in the committed rocprofv3 CSV): the dW GEMM gemm_tn_kernel<bf16_t, 0, 0>
# ------------------------------------------------------------------------------------------------
class _DwShapeLog:
    """records every (M, N, K, conv geometry) the backward pass hands to cvh_gemm_dw during one eager step"""

    def __init__(self):
        self.shapes = []

    def __enter__(self):
        from cvnets_amd import ops
        self._ops, self._orig = ops, ops._weight_grad

        def logged(dy, x, x2, C1, C2, weight, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real, **kw):
            self.shapes.append((B, H, W, Ho, Wo, C1, C2, KH, KW, stride, pad, dil, N, Cin_real))
            return self._orig(dy, x, x2, C1, C2, weight, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real, **kw)

        ops._weight_grad = logged
        from cvnets_amd import fused
        fused.DW_SHAPE_LOG = self.shapes  # the Gram / g^T x GEMMs of the fused InvertedResidual blocks
        return self

    def __exit__(self, *a):
        from cvnets_amd import fused
        fused.DW_SHAPE_LOG = None
        self._ops._weight_grad = self._orig


def dominant_kernel_probe(dtype, shapes):
    """Times the step's dominant kernel class live with HIP events on the launch stream: every dW GEMM (cvh_gemm_dw ->
    gemm_tn_kernel + split reduction) of one training step, on synthetic operands of the recorded shapes, back to back.
    Algorithmic bytes per launch = M*(N + K)*sizeof(T): dY and the (implicit) im2col input are each read once; dW is negligible."""
    from cvnets_amd import _lib, ops
    dev = torch.device("cuda")
    st = torch.cuda.current_stream()
    esz = 2 if dtype == torch.bfloat16 else 4
    total_ms, total_bytes, n = 0.0, 0, 0
    worst = None
    for (B, H, W, Ho, Wo, C1, C2, KH, KW, stride, pad, dil, N, Cin_real) in shapes:
        M, Ktot = B * Ho * Wo, KH * KW * (C1 + C2)
        dy = torch.randn(M, N, device=dev).to(dtype)
        x = torch.randn(B * H * W, C1, device=dev).to(dtype)
        x2 = torch.randn(B * H * W, C2, device=dev).to(dtype) if C2 else None
        dw = torch.empty(N * Cin_real * KH * KW, device=dev)
        n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, Ktot)
        scr = torch.empty(max(n_scr, 1), device=dev)

        def run():
            _lib.call("cvh_gemm_dw", 1 if dtype == torch.bfloat16 else 0, dy.data_ptr(), x.data_ptr(), None if x2 is None else x2.data_ptr(), C1, C2,
                      dw.data_ptr(), B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real, scr.data_ptr(), n_scr, 0, st.cuda_stream)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record(st)
        for _ in range(reps):
            run()
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = M * (N + (C1 + C2)) * esz  # the KH*KW taps re-read the same input rows (L2): count the tensor once
        total_ms += ms
        total_bytes += nbytes
        n += 1
        if worst is None or ms > worst[0]:
            worst = (ms, f"M={M} N={N} K={Ktot}")
        del dy, x, x2, dw, scr
    return {"kernel": "gemm_tn_kernel<bf16_t, *, 0> / gemm_tn_skinny_kernel / gemm_tn128_kernel (+ split reduction): every plain dW GEMM dY^T x im2col(X) of one training step",
            "launches_per_step": n, "avg_ms": round(total_ms / max(n, 1), 4), "total_ms_per_step": round(total_ms, 3),
            "algorithmic_bytes": int(total_bytes / max(n, 1)), "achieved_GBps": round(total_bytes / (total_ms * 1e-3) / 1e9, 1),
            "longest_launch": {"ms": round(worst[0], 4), "shape": worst[1]} if worst else None}


# ------------------------------------------------------------------------------------------------
def _dry_run_step(rank, world, device):
    """stand-in "training step" of the CPU control-flow rehearsal: a fixed amount of host work + the gradient all-reduce"""
    g = torch.full((1 << 16,), float(rank + 1))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.002:
        pass
    if world > 1:
        dist.all_reduce(g)
        g /= world
    return g


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a different number of GPUs")
    if args.dry_run:
        return run_dry(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # CVH_BENCH_SHARE_GPU=1 + CVH_DIST_BACKEND=gloo: developer rehearsal of the N > 1 control flow on a 1-GPU box (all ranks on cuda:0,
    # host-staged collectives); the real runs use one GPU per rank and RCCL.
    dev_index = 0 if os.environ.get("CVH_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import cvnets_amd
    from cvnets_amd.ddp import DistributedDataParallel, distributed_init

    # CVH_DDP_FORCE_COLLECTIVES=1 on ONE GPU: a single-rank RCCL group whose (identity) collectives are issued anyway — the N > 1 code path
    # of this file (rendezvous, in-graph all-reduce, barriers, MAX reduction) executed on the hardware a developer has
    force = os.environ.get("CVH_DDP_FORCE_COLLECTIVES", "0") == "1"
    multi = world > 1 or force
    if multi:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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

    def opt_step():
        if args.torch_optimizer:
            opt.step()
        else:
            opt.step(sync_hyperparameters=False)  # constant rate in this benchmark: the device-side table was filled by the warm-up steps

    graph_has_allreduce = False  # True: the bucket all-reduces (side stream fork / join) and the optimizer are nodes of the hipGraph

    def step():
        nonlocal static_loss
        if graph is not None:
            graph.replay()
            if graph_has_allreduce:
                return
            if multi:
                ddp.allreduce_flat()
        else:
            zero_grads()
            static_loss = fwd_bwd()
            if multi:
                ddp.allreduce_flat()
        if opt is not None and (graph is None or multi):
            opt_step()

    # eager warm-up (also creates every lazily-built tensor before capture); the first step logs the dW GEMM shapes for the kernel probe
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    dw_log = _DwShapeLog()
    with torch.cuda.stream(side):
        for it in range(2):
            zero_grads()
            if it == 0:
                with dw_log:
                    static_loss = fwd_bwd()
            else:
                static_loss = fwd_bwd()
            if multi:
                ddp.allreduce_flat()  # replicas stay identical through the warm-up as well
            if opt is not None:
                opt.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    graph_err = None
    if use_graph:
        # N > 1: first try the whole step as ONE graph — zero-grad, forward, loss, backward, the bucket all-reduces (RCCL kernels captured
        # on the side stream: a fork after the last gradient kernel, a join before the optimizer) and the fused AdamW.  If RCCL refuses
        # to be captured on this stack, fall back to replay + eager all-reduce + eager optimizer (the round-2 path).
        attempts = [True, False] if (multi and os.environ.get("CVH_GRAPH_ALLREDUCE", "1") != "0") else [False]
        # CVH_MAIN_PRIO=1 captures the dX chain on a HIGH-priority stream (the parameter-gradient side stream keeps the default priority).
        # Measured and left off: 92.3 -> 102.2 ms per step — with priorities the two branches of the graph serialise instead of sharing the CUs.
        cap_stream = torch.cuda.Stream(priority=-1) if os.environ.get("CVH_MAIN_PRIO", "0") == "1" else None
        for in_graph in attempts:
            try:
                g = torch.cuda.CUDAGraph()
                # thread_local: RCCL's watchdog thread polls events while we capture; only this thread's calls belong to the graph
                with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local"):
                    ddp.zero_grad()
                    static_loss = fwd_bwd()
                    if in_graph:
                        ddp.allreduce_flat()
                    if opt is not None and (not multi or in_graph):
                        opt_step()
                graph, graph_has_allreduce, graph_err = g, in_graph, None
                break
            except Exception as e:  # pragma: no cover - reported in the JSON line
                graph_err = f"{type(e).__name__}: {e}"[:300]
                graph = None
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if multi:
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
    if multi:
        dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    gpu_ms_per_step = e0.elapsed_time(e1) / args.steps
    loss_val = float(static_loss.item())

    if rank == 0:
        out = report(args, world, wall, gpu_ms_per_step)
        headline = args.mode == "small" and args.res == 256 and args.dtype == "bf16"
        out["config"].update({"hipgraph": graph is not None, "dropout": 0.1, "loss": round(loss_val, 4),
                              "allreduce": ("in-graph (RCCL, side stream)" if graph_has_allreduce else "after replay (RCCL, side stream)") if multi else None,
                              "step": "zero_grad+fwd+CE(ls=0.1)+bwd" + ("+allreduce" if multi else "") +
                                      ("" if opt is None else ("+AdamW(torch fused)" if args.torch_optimizer else "+AdamW(cvh_adamw_multi)"))})
        if graph_err:
            out["config"]["hipgraph_error"] = graph_err
        roofline = out["roofline"]
        pmc = None
        try:  # HBM traffic cannot be counted from inside the process: it comes from the committed rocprofv3 PMC passes of this command
            pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
            if args.batch == pmc["step"]["images"] and headline:
                roofline["traffic"] = pmc["step"]["total_bytes"] / pmc["step"]["images"]  # bytes per image, like algorithmic_bytes_per_image
                roofline["traffic_source"] = pmc["source"]
        except Exception:
            pmc = None
        if not args.no_kernel_probe:
            try:
                dk = dominant_kernel_probe(dtype, dw_log.shapes)
                dk["frac"] = round(dk["achieved_GBps"] * 1e9 / HBM_PEAK, 4)
                if pmc is not None and args.batch == pmc["step"]["images"] and "dominant_kernel" in pmc:
                    dk["traffic"] = pmc["dominant_kernel"].get("bytes_per_launch")
                    dk["rocprof_avg_ms"] = pmc["dominant_kernel"].get("rocprof_avg_ms")
                roofline["dominant_kernel"] = dk
            except Exception as e:  # pragma: no cover
                roofline["dominant_kernel"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def report(args, world, wall, gpu_ms_per_step=None):
    """the JSON line (rank 0) from the max-over-ranks wall time of the timed region"""
    imgs = args.batch * world * args.steps
    value = imgs / wall
    per_gpu = value / world
    t_img = 1.0 / per_gpu
    frac_hbm = (ALGO_BYTES_PER_IMG / HBM_PEAK) / t_img
    frac_mfma = (ALGO_FLOP_PER_IMG / MFMA_PEAK_BF16) / t_img
    headline = args.mode == "small" and args.res == 256 and args.dtype == "bf16"
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
    }
    if gpu_ms_per_step is not None:
        roofline["gpu_ms_per_step_hip_events"] = round(gpu_ms_per_step, 3)
    return {
        "metric": "images/sec (fwd+bwd) MobileViT-S 256x256 bf16" if headline else f"images/sec (fwd+bwd) MobileViT-{args.mode} {args.res}x{args.res} {args.dtype}",
        "value": round(value, 2),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3 / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (randn images, random-init weights)" if not args.dry_run else "dry-run (control-flow rehearsal on CPU: NOT a measurement)",
        "config": {"workload": f"MobileViT-{args.mode} {args.res}x{args.res}, {args.batch} img/GPU, global batch {args.batch * world}",
                   "parallelism": f"dp{world}"},
        "images_per_sec_per_gpu": round(per_gpu, 2),
        "roofline": roofline,
    }


def run_dry(args, rank, world):
    """CPU rehearsal of exactly the launch / rendezvous / barrier / timing / max-over-ranks / JSON control flow of `run` (gloo)"""
    from cvnets_amd.ddp import distributed_init
    if world > 1:
        distributed_init("gloo", torch.device("cpu"))
    for _ in range(args.warmup):
        _dry_run_step(rank, world, None)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = _dry_run_step(rank, world, None)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        out = report(args, world, float(t.item()))
        out["config"].update({"step": "dry-run" + ("+allreduce" if world > 1 else ""), "allreduce_mean": float(g[0])})
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _spawn_entry(local_rank, argv, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(parse(argv).gpus), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    run(parse(argv))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: spawn one rank per GPU ourselves (the reference does the same in main_train.py:261-265)
        import torch.multiprocessing as mp
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_spawn_entry, args=(sys.argv[1:], port), nprocs=args.gpus, join=True)
        return
    run(args)


if __name__ == "__main__":
    main()
