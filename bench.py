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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch images per GPU at every N (global batch N x 1024).  strong: --batch is the GLOBAL batch and every "
                         "GPU gets --batch / N of it — at the default 1024 this is the north-star question: the recipe's global batch 1024 sharded "
                         "over 8 GPUs = 128 img/GPU")
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


def per_gpu_batch(args, world: int) -> int:
    """images per GPU per step: --batch (weak scaling) or --batch / N (strong scaling: --batch is the global batch)"""
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit(f"--scaling strong: global batch {args.batch} is not divisible by {world} GPUs")
        return args.batch // world
    return args.batch


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
# dominant kernel family
# ------------------------------------------------------------------------------------------------
# WHICH family dominates, its in-step average launch duration and its HBM traffic come from profiles/step_profile.json — written by
# tools/make_step_profile.py from rocprofv3 runs (kernel trace + two PMC passes) of THIS command, stamped with a hash of the sources it was
# measured on; a profile of another source tree is refused (the line then says so and carries the live measurements only).
# Live, on every run: the family's launches per step and their algorithmic bytes (tallied by the library itself during one eager step), and an
# isolated re-timing of exactly those launches with HIP events on the launch stream.
def _src_hash():
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(REPO, "ml-cvnets_amd", "csrc", "*")) + glob.glob(os.path.join(REPO, "ml-cvnets_amd", "cvnets_amd", "*.py")) +
                   [os.path.join(REPO, "bench.py"), os.path.join(REPO, "include", "cvnets_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


FAMILIES = {  # kernel families the library tallies (cvh_stream_counters): name in the rocprofv3 kernel table -> (index, description)
    "gemm_stream_kernel": (0, "gemm_stream_kernel<FW, NK, EM> (csrc/gemm_stream.hip): every short-K pointwise GEMM of the step — token linears, 1x1 "
                              "convolutions, BatchNorm-link expansion / projection GEMMs"),
    "conv_gemm_kernel": (1, "conv_gemm_kernel<T, NF, BK, FX, WP> (csrc/conv_gemm.hpp): the implicit-GEMM family — conv_1, operand-transform (BatchNorm-link) "
                            "projections, two-source expansion dX, K > 320 linears, classifier"),
    # cvh_family_counters: the fused InvertedResidual kernels (indices 2 + family)
    "dwx_fwd_kernel": (2, "dwx_fwd_kernel<S, CIN> (csrc/dwx.hip): expansion 1x1 + BatchNorm + SiLU + depthwise 3x3 of an InvertedResidual block, the 4x-wide "
                          "expansion output recomputed from the narrow input; algorithmic bytes = x + y2"),
    "dwx_bwd_kernel": (3, "dwx_bwd_kernel<S, CIN> (csrc/dwx.hip): backward of the same (depthwise dX and dW, SiLU', BatchNorm statistics) with y1 recomputed; "
                          "algorithmic bytes = x + g2 + y2 + g1 (no halo)"),
    "ir_pb_kernel": (4, "ir_pb_kernel<NA> (csrc/ir_pb.hip): projection backward of an InvertedResidual block from one pass over y2; algorithmic bytes = "
                        "dout + y3 + y2 + g2"),
    "ir_exp_bwd_kernel": (5, "ir_exp_bwd_kernel<HBK, CB> (csrc/ir_bwd.hip): expansion dX + raw dW from one pass over g1; algorithmic bytes = g1 + x + dx (+ residual)"),
    "ir_red_fwd_kernel": (6, "ir_red_fwd_kernel<HBK, NB> (csrc/ir_fwd.hip): projection forward with BatchNorm + SiLU on load; algorithmic bytes = y2 + y3"),
}
N_FAMILY_COUNTERS = 5


class _StreamGemmLog:
    """records the (M, K, N, epilogue operands) of every launch that lands on gemm_stream_kernel during one eager step, by watching the
    library's own tally (cvh_stream_counters) around the two entry points that can dispatch to it"""

    def __init__(self):
        self.calls = []
        self.dwx_bwd_calls = []
        self.launches = 0
        self.alg_bytes = 0
        self.tallies = [0] * (4 + 2 * N_FAMILY_COUNTERS)

    def __enter__(self):
        import ctypes
        from cvnets_amd import _lib
        self._lib = _lib
        self._orig = _lib.call
        buf = (ctypes.c_longlong * 4)()
        self._buf = buf

        def tally():
            _lib.load().cvh_stream_counters(0, buf)
            return buf[0], buf[1]

        _lib.load().cvh_stream_counters(1, None)
        for f in range(N_FAMILY_COUNTERS):
            _lib.load().cvh_family_counters(f, 1, None)
        self._tally = tally

        def call(name, *a):
            if name == "cvh_dwx_bwd":  # (dtype, x, w1, in_stats, act1, g_out, y_out, ca, cb, cc, wd, g_in, stats, dw, B, H, W, Ho, Wo, Cin, hid, stride, st)
                self.dwx_bwd_calls.append(tuple(a[14:22]))
            if name not in ("cvh_conv_gemm", "cvh_pw_gemm_bn"):
                return self._orig(name, *a)
            n0, _ = tally()
            rc = self._orig(name, *a)
            n1, _ = tally()
            if n1 > n0:
                if name == "cvh_conv_gemm":  # (dtype, src1, src2, C1, C2, wgt, out, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, bias, act, save_pre, aux, aact, res, p, seed, sid, stats, st)
                    self.calls.append(("g", a[7] * a[10] * a[11], a[3], a[17], dict(bias=a[18] is not None, act=a[19], save_pre=a[20] is not None,
                                                                                actgrad=a[21] is not None, aact=a[22], residual=a[23] is not None, drop=a[24])))
                else:  # (dtype, a, a_xf, K, wgt, out, M, N, residual, e_mode, e_aux, e_stats, e_act, stats_part, st)
                    self.calls.append(("f", a[6], a[3], a[7], dict(residual=a[8] is not None, e_mode=a[9], e_act=a[12], stats=a[13] is not None)))
            return rc

        _lib.call = call  # ops.py / fused.py call through the module attribute, so this patches them too
        return self

    def __exit__(self, *exc):
        self._lib.call = self._orig
        import ctypes
        self.launches, self.alg_bytes = self._tally()
        self.tallies = list(self._buf)
        b2 = (ctypes.c_longlong * 2)()
        for f in range(N_FAMILY_COUNTERS):
            self._lib.load().cvh_family_counters(f, 0, b2)
            self.tallies += [b2[0], b2[1]]


def dwx_bwd_probe(log):
    """isolated re-timing of the step's dwx_bwd_kernel launches (synthetic operands of the recorded geometries), HIP events on the launch
    stream; returns (total ms per step's worth of launches, launches)"""
    from cvnets_amd import _lib
    dev = torch.device("cuda")
    st = torch.cuda.current_stream()
    total_ms, n = 0.0, 0
    for (B, H, W, Ho, Wo, Cin, hid, stride) in log.dwx_bwd_calls:
        x = torch.randn(B * H * W, Cin, device=dev).bfloat16()
        w1 = (torch.randn(hid, Cin, device=dev) * Cin ** -0.5).bfloat16()
        wd = (torch.randn(9, hid, device=dev) * 0.4).bfloat16()
        stats = torch.stack([torch.zeros(hid, device=dev), torch.ones(hid, device=dev), torch.ones(hid, device=dev), torch.zeros(hid, device=dev)]).contiguous()
        g2 = torch.randn(B * Ho * Wo, hid, device=dev).bfloat16()
        y2 = torch.randn(B * Ho * Wo, hid, device=dev).bfloat16()
        g1 = torch.empty(B * H * W, hid, device=dev, dtype=torch.bfloat16)
        ca, cb, cc = torch.ones(hid, device=dev), torch.full((hid,), 0.1, device=dev), torch.zeros(hid, device=dev)
        R = _lib.query("cvh_dwx_rows", B, Ho, Wo, hid, stride)
        part, dwp = torch.empty(R * 2 * hid, device=dev), torch.empty(R * hid * 9, device=dev)

        def go():
            _lib.call("cvh_dwx_bwd", 1, x.data_ptr(), w1.data_ptr(), stats.data_ptr(), 1, g2.data_ptr(), y2.data_ptr(), ca.data_ptr(), cb.data_ptr(),
                      cc.data_ptr(), wd.data_ptr(), g1.data_ptr(), part.data_ptr(), dwp.data_ptr(), B, H, W, Ho, Wo, Cin, hid, stride, st.cuda_stream)

        go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            go()
        e1.record(st)
        torch.cuda.synchronize()
        total_ms += e0.elapsed_time(e1) / 3
        n += 1
    return total_ms, n


def stream_gemm_probe(log):
    """isolated re-timing of the step's gemm_stream_kernel launches (synthetic operands of the recorded shapes, same epilogues), HIP events on
    the launch stream; returns (total ms per step's worth of launches, launches)"""
    from cvnets_amd import _lib, ops
    from cvnets_amd.fused import _pw_gemm
    dev = torch.device("cuda")
    st = torch.cuda.current_stream()
    total_ms, n = 0.0, 0
    seed = torch.tensor([12345], dtype=torch.int64, device=dev)
    for kind, M, K, N, o in log.calls:
        x = torch.randn(M, K, device=dev).bfloat16()
        wp = ops.pack_weight(torch.randn(N, K, 1, 1, device=dev) * 0.05, torch.bfloat16, 0)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.randn(M, N, device=dev).bfloat16() if (o.get("actgrad") or o.get("residual") or o.get("e_mode")) else None
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if o.get("save_pre") else None
        if kind == "g":
            bias = torch.zeros(N, device=dev) if o["bias"] else None

            def run():
                ops._conv_gemm(x, None, K, 0, wp, out, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, bias=bias, act=o["act"], save_pre=pre,
                               actgrad_aux=aux if o["actgrad"] else None, actgrad_act=o["aact"], residual=aux if o["residual"] else None,
                               drop_p=o["drop"], seed=seed if o["drop"] > 0 else None, stream_id=1)
        else:
            stats = (torch.rand(4, N, device=dev) + 0.5) if o["e_mode"] else None

            def run():
                _pw_gemm(x, None, K, wp, out, M, N, residual=aux if o["residual"] else None, e_mode=o["e_mode"], e_aux=aux if o["e_mode"] else None,
                         e_stats=stats, e_act=o["e_act"], want_stats=o["stats"])
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            run()
        e1.record(st)
        e1.synchronize()
        total_ms += e0.elapsed_time(e1) / 3
        n += 1
        del x, wp, out, aux, pre
    return total_ms, n


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
    # (buckets: 1 MB first, 8 MB after — MobileViT-S = 4 messages; cvnets_amd/ddp.py)
    # In-place parameter gradients never reach autograd, so per-parameter hooks cannot drive the exchange; the wrapper's BOUNDARY hooks do:
    # when backward passes the input of the earliest top-level child of a bucket, the bucket's all-reduce forks onto the side stream.  Python
    # hooks do not run on a hipGraph REPLAY, but they do run while the step is CAPTURED: the capture below enables them, so the forks behind
    # the last gradient kernel of every bucket and the join at the end of backward become graph edges and the replayed step overlaps the
    # all-reduces with the rest of backward.  Eager steps (warm-up, --no-graph) reduce after backward.
    ddp = DistributedDataParallel(model, broadcast_buffers=False, boundary_overlap=True)
    ddp.hooks_enabled = False
    ddp.boundary_enabled = False
    cvnets_amd.ops.set_inplace_param_grads(True)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = None
    if not args.no_optimizer:
        if args.torch_optimizer:
            opt = torch.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01, fused=True, capturable=True)
        else:  # every parameter stepped by one kernel launch, rates and step counter on the device
            opt = cvnets_amd.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01)

    pgb = per_gpu_batch(args, world)
    x = torch.randn(pgb, 3, args.res, args.res, device=dev)
    y = torch.randint(0, 1000, (pgb,), device=dev)

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
    graph_overlapped = 0         # buckets whose all-reduce node starts before the last backward kernel in the captured graph

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
    dw_log = _StreamGemmLog()
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
                early0, fin0 = ddp.early_launches, ddp.finish_count
                with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local"):
                    ddp.zero_grad()
                    ddp.boundary_enabled = bool(in_graph) and os.environ.get("CVH_GRAPH_OVERLAP", "1") != "0"
                    static_loss = fwd_bwd()  # boundaries on: every complete bucket forks onto the side stream; `finish` joins at the end of backward
                    ddp.boundary_enabled = False
                    if in_graph and ddp.finish_count == fin0:  # no boundary fired (switched off / unsupported model): one exchange after backward
                        ddp.allreduce_flat()
                    if opt is not None and (not multi or in_graph):
                        opt_step()
                graph, graph_has_allreduce, graph_err = g, in_graph, None
                graph_overlapped = ddp.early_launches - early0
                break
            except Exception as e:  # pragma: no cover - reported in the JSON line
                graph_err = f"{type(e).__name__}: {e}"[:300]
                graph = None
                ddp.boundary_enabled = False
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
                              "communicator": (("cvh_comm (own RCCL communicator, cvnets_amd/comm.py)" if ddp.comm is not None else "torch.distributed (fall-back)")
                                               if multi else None),
                              "allreduce_buckets": ddp.overlap_report()["bucket_mb"] if multi else None,
                              "allreduce_buckets_started_inside_backward": graph_overlapped if (multi and graph_has_allreduce) else None,
                              "step": "zero_grad+fwd+CE(ls=0.1)+bwd" + ("+allreduce" if multi else "") +
                                      ("" if opt is None else ("+AdamW(torch fused)" if args.torch_optimizer else "+AdamW(cvh_adamw_multi)"))})
        if graph_err:
            out["config"]["hipgraph_error"] = graph_err
        roofline = out["roofline"]
        prof, stale = None, None
        try:  # HBM traffic and the in-step kernel table cannot be produced from inside the process: they come from the rocprofv3 runs of this command
            prof = json.load(open(os.path.join(REPO, "profiles", "step_profile.json")))
            stale = prof.get("src_hash") != _src_hash()
            if args.batch != prof["step"]["images"] or not headline:
                prof = None
        except Exception:
            prof = None
        if prof is not None and not stale:
            roofline["traffic"] = prof["step"]["total_bytes"] / prof["step"]["images"]  # bytes per image, like algorithmic_bytes_per_image
            roofline["traffic_source"] = prof["source"]
        elif prof is not None:
            roofline["traffic_note"] = "profiles/step_profile.json was measured on a different source tree (src_hash mismatch): not quoted"
        # the dominant family = the one with the largest share of kernel time in the rocprofv3 table of THIS source tree (without a valid
        # profile: the streaming GEMM, which the live probe below re-times)
        dom = prof["dominant"] if (prof is not None and not stale and prof["dominant"] in FAMILIES) else "gemm_stream_kernel"
        fi, fdesc = FAMILIES[dom]
        n_l, n_b = dw_log.tallies[2 * fi], dw_log.tallies[2 * fi + 1]
        dk = {"kernel": fdesc, "family": dom, "launches_per_step": int(n_l), "algorithmic_bytes": int(n_b / max(n_l, 1)),
              "algorithmic_bytes_note": "per launch, averaged over the step's launches: every operand and result tensor of the launch once (input tensor(s) + "
                                        "out + every [M][N] epilogue operand; no halo re-reads), tallied by the library during one eager step "
                                        "(cvh_stream_counters / cvh_family_counters)"}
        if prof is not None and not stale:
            fam = {f["family"]: f for f in prof["families"]}
            f = fam.get(dom)
            if f is not None:
                dk.update({"share_of_kernel_time": f["share_of_kernel_time"], "avg_ms": f["avg_ms"], "ms_per_step": f["ms_per_step"],
                           "profile_launches_per_step": f["launches_per_step"], "traffic": f["hbm_bytes_per_step"] / max(f["launches_per_step"], 1),
                           "avg_ms_source": "in-step average from the rocprofv3 kernel trace of this command (profiles/step_profile.json)"})
                dk["achieved_GBps"] = round(n_b / max(n_l, 1) / (f["avg_ms"] * 1e-3) / 1e9, 1)
                dk["frac"] = round(dk["achieved_GBps"] * 1e9 / HBM_PEAK, 4)
            if prof["dominant"] not in FAMILIES:
                d0 = fam[prof["dominant"]]
                dk["note"] = f"the profile's largest family is {prof['dominant']} ({d0['share_of_kernel_time']:.3f} of kernel time, {d0['hbm_GBps']} GB/s of counter traffic): not tallied"
            dk["runner_up"] = {k: {"share_of_kernel_time": fam[k]["share_of_kernel_time"], "avg_ms": fam[k]["avg_ms"], "hbm_GBps": fam[k]["hbm_GBps"]}
                               for k in list(fam)[:4] if k != dom}
        if not args.no_kernel_probe:
            try:
                if dom == "dwx_bwd_kernel":
                    iso_ms, n_iso = dwx_bwd_probe(dw_log)
                    iso_bytes = n_b
                else:
                    iso_ms, n_iso = stream_gemm_probe(dw_log)
                    iso_bytes = dw_log.alg_bytes
                iso_name = dom if dom == "dwx_bwd_kernel" else "gemm_stream_kernel"
                iso = {"kernel": iso_name, "launches": n_iso, "total_ms_per_step": round(iso_ms, 3), "avg_ms": round(iso_ms / max(n_iso, 1), 5),
                       "achieved_GBps": round(iso_bytes / (iso_ms * 1e-3) / 1e9, 1),
                       "what": f"the step's {iso_name} launches re-issued back to back on synthetic operands, HIP events on the launch stream"}
                dk["isolated_probe"] = iso
                if iso_name == dom:  # (a live probe of another family than the one the profile ranks first stays apart)
                    dk["isolated_avg_ms"], dk["isolated_achieved_GBps"] = iso["avg_ms"], iso["achieved_GBps"]
                    if "avg_ms" not in dk:  # no valid profile: the live numbers are all there is
                        dk.update({"avg_ms": dk["isolated_avg_ms"], "avg_ms_source": "isolated HIP-event probe (no valid profiles/step_profile.json for this source tree)",
                                   "achieved_GBps": dk["isolated_achieved_GBps"], "frac": round(dk["isolated_achieved_GBps"] * 1e9 / HBM_PEAK, 4)})
            except Exception as e:  # pragma: no cover
                dk["probe_error"] = str(e)[:200]
        roofline["dominant_kernel"] = dk
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if multi:
        from cvnets_amd import comm as hip_comm
        torch.cuda.synchronize()
        hip_comm.destroy_default()
        dist.destroy_process_group()


def report(args, world, wall, gpu_ms_per_step=None):
    """the JSON line (rank 0) from the max-over-ranks wall time of the timed region"""
    pgb = per_gpu_batch(args, world)
    imgs = pgb * world * args.steps
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
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (randn images, random-init weights)" if not args.dry_run else "dry-run (control-flow rehearsal on CPU: NOT a measurement)",
        "config": {"workload": f"MobileViT-{args.mode} {args.res}x{args.res}, {pgb} img/GPU, global batch {pgb * world}"
                               + (" (strong scaling: the global batch is fixed, sharded over the GPUs)" if args.scaling == "strong" else ""),
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
