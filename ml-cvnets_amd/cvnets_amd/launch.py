"""Launcher: the reference's training entry point with the HIP path dropped in.

    python -m cvnets_amd.launch --common.config-file config/classification/imagenet/mobilevit.yaml [reference CLI options ...]

Mirrors ``main_train.py`` of the reference (``main`` :33-173, ``distributed_worker`` :176-188, ``main_worker`` :191-262) with exactly
three substitutions, so that ``engine/training_engine.py`` (``Trainer``), ``options/``, ``data/``, ``loss_fn/``, ``optim/`` and every
``config/*.yaml`` of the reference run UNMODIFIED ("≈140 lines of glue"):

  1. after ``cvnets.get_model(opts)`` (:67) the model is class-swapped onto the HIP kernels — ``dropin.swap_to_hip(model, strict=True)``;
  2. ``torch.nn.parallel.DistributedDataParallel`` (:90-96) -> ``cvnets_amd.ddp.DistributedDataParallel`` (flat fp32 buckets, RCCL
     all-reduce on a side HIP stream overlapped with backward);
  3. ``utils.ddp_utils.distributed_init`` (ddp_utils.py:47-89) -> ``distributed_init`` below: same option keys (``ddp.dist_url``,
     ``ddp.dist_port``, ``ddp.backend``, ``ddp.rank``, ``ddp.world_size``), same dummy all-reduce, but the process group is bound to its
     GPU (``device_id``) and the default rendezvous host is 127.0.0.1 instead of ``socket.gethostname()`` (one node, xGMI).

The reference tree must be importable (``PYTHONPATH=/path/to/ml-cvnets``); this module imports nothing from it at import time.
``swap`` / ``loader_factory`` exist for the tests: the CPU plumbing test (BASELINE.json configs[0]) runs this very code path with
``swap=False`` — no GPU, no HIP library — and a dummy loader in the style of the reference's ``tests/dummy_loader.py``.
"""
from __future__ import annotations

import argparse
import os
import math
from typing import Callable, List, Optional

import torch


def distributed_init(opts) -> int:
    """utils/ddp_utils.py:47-89 with the process group bound to this rank's GPU.  Returns the rank (also stored in ``ddp.rank``)."""
    import torch.distributed as dist

    ddp_url = getattr(opts, "ddp.dist_url", None)
    if ddp_url is None:
        ddp_url = "tcp://127.0.0.1:{}".format(getattr(opts, "ddp.dist_port", 6006))
        setattr(opts, "ddp.dist_url", ddp_url)
    node_rank = getattr(opts, "ddp.rank", 0)
    world_size = getattr(opts, "ddp.world_size", 0)
    if not dist.is_initialized():
        backend = getattr(opts, "ddp.backend", "nccl")
        if backend is None:
            backend = "nccl" if (dist.is_nccl_available() and torch.cuda.is_available()) else "gloo"
        kwargs = {}
        device = getattr(opts, "dev.device", None)
        if backend == "nccl" and isinstance(device, torch.device) and device.type == "cuda" and device.index is not None:
            kwargs["device_id"] = device  # eager communicator creation on THIS GPU; no device guessing inside RCCL
        dist.init_process_group(backend=backend, init_method=ddp_url, world_size=world_size, rank=node_rank, **kwargs)
        if torch.cuda.is_available() and backend == "nccl":
            # ddp_utils.py:84-85 creates the communicator with a dummy all-reduce.  Here the hot path gets a communicator of its own: the
            # unique id travels through the TCP store this rendezvous just opened, every rank joins, one self-test all-reduce + broadcast
            # (cvnets_amd/comm.py).  The torch group stays for what the engine does off the path (barriers, metric reductions).
            from . import comm as hip_comm

            dev = device if kwargs else torch.device("cuda", torch.cuda.current_device())
            with torch.cuda.device(dev):
                if hip_comm.init_default(dev) is None:
                    dist.all_reduce(torch.zeros(1, device=dev))
    node_rank = dist.get_rank()
    setattr(opts, "ddp.rank", node_rank)
    return node_rank


def main(opts: argparse.Namespace, *, swap: Optional[bool] = None, loader_factory: Optional[Callable] = None, **kwargs):
    """main_train.py:33-173.  Returns the Trainer after ``run`` (the reference returns None; the tests inspect the trained model)."""
    from common import DEFAULT_EPOCHS, DEFAULT_ITERATIONS, DEFAULT_MAX_EPOCHS, DEFAULT_MAX_ITERATIONS
    from cvnets import EMA, get_model
    from engine import Trainer
    from loss_fn import build_loss_fn
    from optim import build_optimizer
    from optim.scheduler import build_scheduler
    from torch.cuda.amp import GradScaler
    from utils import logger
    from utils.checkpoint_utils import load_checkpoint, load_model_state
    from utils.ddp_utils import is_master

    from . import ddp as hip_ddp
    from . import dropin

    dev_id = getattr(opts, "dev.device_id", torch.device("cpu"))
    device = getattr(opts, "dev.device", torch.device("cpu"))
    use_distributed = getattr(opts, "ddp.use_distributed")
    is_master_node = is_master(opts)

    if loader_factory is None:
        from data import create_train_val_loader as loader_factory
    train_loader, val_loader, train_sampler = loader_factory(opts)

    # iteration / epoch budget (main_train.py:46-65)
    if getattr(opts, "scheduler.is_iteration_based"):
        max_iter = getattr(opts, "scheduler.max_iterations", DEFAULT_ITERATIONS)
        if max_iter is None or max_iter <= 0:
            setattr(opts, "scheduler.max_iterations", DEFAULT_ITERATIONS)
        setattr(opts, "scheduler.max_epochs", DEFAULT_MAX_EPOCHS)
    else:
        max_epochs = getattr(opts, "scheduler.max_epochs", DEFAULT_EPOCHS)
        if max_epochs is None or max_epochs <= 0:
            setattr(opts, "scheduler.max_epochs", DEFAULT_EPOCHS)
        setattr(opts, "scheduler.max_iterations", DEFAULT_MAX_ITERATIONS)

    model = get_model(opts)
    if is_master_node:
        model.info()

    # substitution 1: the HIP path.  The kernels exist for GPU tensors only — there is no CPU fallback, so the swap follows the device.
    if swap is None:
        swap = isinstance(device, torch.device) and device.type == "cuda"
    if swap:
        counts, _ = dropin.swap_to_hip(model, strict=True)
        if is_master_node:
            logger.log("cvnets_amd: class-swapped {} modules onto libcvnets_hip.so ({} classes)".format(sum(counts.values()), len(counts)))

    memory_format = torch.channels_last if getattr(opts, "common.channels_last") else torch.contiguous_format
    model = model.to(device=device, memory_format=memory_format)

    if getattr(opts, "ddp.use_deprecated_data_parallel"):
        logger.error("cvnets_amd.launch: DataParallel is not supported (one process per GPU only)")
    elif use_distributed:
        # substitution 2: flat-bucket RCCL data parallelism.  Wrap BEFORE EMA / optimizer construction: the wrapper re-points gradient
        # and buffer storage into its flat tensors.
        # boundary_overlap: with the fused optimizer below the kernels add parameter gradients in place (no per-parameter autograd hooks);
        # the buckets then start when backward passes the inputs of the model's top-level children, the rest at the end of backward
        model = hip_ddp.DistributedDataParallel(model, boundary_overlap=True)
        if is_master_node:
            logger.log("Using cvnets_amd.ddp.DistributedDataParallel (RCCL, {} bucket(s), {:.1f} MB of gradients)".format(
                len(model.buckets), model.grad_bytes() / 1e6))

    criteria = build_loss_fn(opts).to(device=device)
    optimizer = build_optimizer(model, opts=opts)
    # substitution 4: when the YAML asks for AdamW on the GPU, the reference's Trainer steps cvnets_amd.optim.AdamW — one
    # cvh_adamw_multi launch over every parameter instead of torch's per-tensor / foreach kernels — and the gradients live in one flat
    # buffer the backward kernels add into (no per-parameter AccumulateGrad work; `zero_grad(set_to_none=True)` becomes one memset).
    # Same param_groups, so the reference's scheduler and checkpointing see what they expect.  CVH_ENGINE_FUSED_OPT=0 keeps torch's.
    if swap and isinstance(device, torch.device) and device.type == "cuda" and os.environ.get("CVH_ENGINE_FUSED_OPT", "1") != "0" \
            and isinstance(optimizer, torch.optim.AdamW) and not any(g.get("amsgrad", False) for g in optimizer.param_groups):
        from . import ops as hip_ops
        from . import optim as hip_optim

        optimizer = hip_optim.AdamW.from_torch(optimizer, flat_grads=True)
        hip_ops.set_inplace_param_grads(True)
        if is_master_node:
            logger.log("cvnets_amd: optimizer AdamW -> cvnets_amd.optim.AdamW (one launch per step, flat in-place gradients)")
    gradient_scaler = GradScaler(enabled=getattr(opts, "common.mixed_precision"))
    scheduler = build_scheduler(opts=opts)

    model_ema = None
    if getattr(opts, "ema.enable"):
        model_ema = EMA(model=model, ema_momentum=getattr(opts, "ema.momentum"), device=device)

    best_metric = 0.0 if getattr(opts, "stats.checkpoint_metric_max") else math.inf
    start_epoch = start_iteration = 0
    resume_loc, finetune_loc = getattr(opts, "common.resume"), getattr(opts, "common.finetune")
    if resume_loc is not None or getattr(opts, "common.auto_resume"):
        (model, optimizer, gradient_scaler, start_epoch, start_iteration, best_metric, model_ema) = load_checkpoint(
            opts=opts, model=model, optimizer=optimizer, model_ema=model_ema, gradient_scaler=gradient_scaler)
    elif finetune_loc is not None:
        model, model_ema = load_model_state(opts=opts, model=model, model_ema=model_ema)

    training_engine = Trainer(opts=opts, model=model, validation_loader=val_loader, training_loader=train_loader, optimizer=optimizer,
                              criterion=criteria, scheduler=scheduler, start_epoch=start_epoch, start_iteration=start_iteration,
                              best_metric=best_metric, model_ema=model_ema, gradient_scaler=gradient_scaler)
    training_engine.run(train_sampler=train_sampler)
    return training_engine


def distributed_worker(i: int, main_fn, opts, kwargs):
    """main_train.py:176-188, one process per GPU."""
    setattr(opts, "dev.device_id", i)
    torch.cuda.set_device(i)
    setattr(opts, "dev.device", torch.device(f"cuda:{i}"))
    ddp_rank = getattr(opts, "ddp.rank", None)
    if ddp_rank is None:  # torch.multiprocessing.spawn
        ddp_rank = kwargs.get("start_rank", 0) + i
        setattr(opts, "ddp.rank", ddp_rank)
    node_rank = distributed_init(opts)  # substitution 3
    setattr(opts, "ddp.rank", node_rank)
    main_fn(opts, **{k: v for k, v in kwargs.items() if k != "start_rank"})


def main_worker(args: Optional[List[str]] = None, **kwargs):
    """main_train.py:191-262: option parsing, device set-up, batch-size scaling, spawn of one worker per GPU."""
    from options.opts import get_training_arguments
    from utils import logger, resources
    from utils.common_utils import create_directories, device_setup
    from utils.ddp_utils import is_master

    opts = get_training_arguments(args=args)
    opts = device_setup(opts)
    if getattr(opts, "ddp.rank") < 0:
        logger.error("--rank should be >=0. Got {}".format(getattr(opts, "ddp.rank")))
    exp_dir = "{}/{}".format(getattr(opts, "common.results_loc"), getattr(opts, "common.run_label"))
    setattr(opts, "common.exp_loc", exp_dir)
    create_directories(dir_path=exp_dir, is_master_node=is_master(opts))

    num_gpus = getattr(opts, "dev.num_gpus")
    world_size = getattr(opts, "ddp.world_size")
    setattr(opts, "ddp.use_distributed", num_gpus > 1)
    if num_gpus > 0:
        assert torch.cuda.is_available(), "We need a GPU (ROCm: device type 'cuda') for training on GPUs."
    n_cpus = resources.cpu_count()
    dataset_workers = getattr(opts, "dataset.workers", -1)

    if num_gpus <= 1:
        if dataset_workers == -1:
            setattr(opts, "dataset.workers", n_cpus)
        setattr(opts, "dataset.train_batch_size0", getattr(opts, "dataset.train_batch_size0") * max(1, num_gpus))
        setattr(opts, "dataset.val_batch_size0", getattr(opts, "dataset.val_batch_size0") * max(1, num_gpus))
        setattr(opts, "dev.device_id", None)
        return main(opts=opts, **kwargs)

    setattr(opts, "dev.device_id", getattr(opts, "ddp.device_id"))
    if world_size == -1:
        world_size = num_gpus
        setattr(opts, "ddp.world_size", world_size)
    if dataset_workers == -1 or dataset_workers is None:
        setattr(opts, "dataset.workers", n_cpus // num_gpus)
    start_rank = getattr(opts, "ddp.rank")
    setattr(opts, "ddp.rank", None)  # set inside distributed_worker
    kwargs["start_rank"] = start_rank
    setattr(opts, "ddp.start_rank", start_rank)
    torch.multiprocessing.spawn(fn=distributed_worker, args=(main, opts, kwargs), nprocs=num_gpus)


if __name__ == "__main__":
    main_worker()
