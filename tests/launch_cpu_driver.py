"""Driver for tests/test_launch_cpu.py (runs in its own process: the reference tree has top-level packages named `tests`, `utils`,
`data`, `common` that must not leak into the pytest process).

BASELINE.json configs[0]: MobileViT-XXS 32x32 fp32, batch 8, config/classification, PyTorch CPU, world_size = 1 — driven by the
reference's UNMODIFIED engine/training_engine.py `Trainer` through cvnets_amd.launch.main (swap disabled: there is no GPU here and
the HIP path has no CPU fallback).  Prints one JSON line."""
import json
import os
import sys

REF = os.environ.get("CVNETS_REFERENCE_ROOT", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "oracle", "ref_shim"), REF, os.path.join(REPO, "ml-cvnets_amd")]
os.chdir(REF)

import autostub  # noqa: E402

autostub.install()

import torch  # noqa: E402
import torch.utils.data as tdata  # noqa: E402


class SeededClassificationDataset(tdata.Dataset):
    """the pattern of the reference's tests/dummy_datasets/classification.py:38-55, but seeded per sample index"""

    def __init__(self, opts, n=32):
        self.n = n
        setattr(opts, "model.classification.n_classes", 1000)
        for k in ("train", "val", "test"):
            setattr(opts, f"dataset.collate_fn_name_{k}", "image_classification_data_collate_fn")

    def __getitem__(self, tup):
        h, w, idx = tup
        g = torch.Generator().manual_seed(1234 + int(idx))
        return {"samples": torch.randn(3, h, w, generator=g), "targets": torch.randint(0, 1000, (1,), generator=g).long(),
                "sample_id": torch.tensor([int(idx)]).long()}

    def __len__(self):
        return self.n


def loader_factory(opts):
    from functools import partial
    from data.collate_fns import build_collate_fn
    from data.loader.dataloader import CVNetsDataLoader
    from data.sampler import build_sampler

    setattr(opts, "stats.train", ["loss"])
    setattr(opts, "stats.val", ["loss"])
    setattr(opts, "stats.checkpoint_metric", "loss")
    setattr(opts, "stats.checkpoint_metric_max", False)
    tr, va = SeededClassificationDataset(opts), SeededClassificationDataset(opts, 16)
    s_tr = build_sampler(opts=opts, n_data_samples=len(tr), is_training=True)
    s_va = build_sampler(opts=opts, n_data_samples=len(va), is_training=False)
    c_tr, c_va = build_collate_fn(opts=opts)
    mk = lambda ds, s, c: CVNetsDataLoader(dataset=ds, batch_size=1, num_workers=0, pin_memory=False, batch_sampler=s,  # noqa: E731
                                            persistent_workers=False, collate_fn=partial(c, opts=opts) if c is not None else None, prefetch_factor=None)
    return mk(tr, s_tr, c_tr), mk(va, s_va, c_va), s_tr


def main(rank=0, world=1, port=0):
    from options.opts import get_training_arguments
    from options.utils import load_config_file
    from utils.common_utils import create_directories, device_setup
    from cvnets_amd import launch

    out_dir = sys.argv[1]
    parser = get_training_arguments(parse_args=False)
    opts = parser.parse_args([])
    setattr(opts, "common.config_file", "config/classification/imagenet/mobilevit.yaml")
    opts = load_config_file(opts)
    opts = device_setup(opts)
    for k, v in {
        "model.classification.mit.mode": "xx_small", "common.mixed_precision": False, "ddp.use_distributed": False, "ddp.rank": 0,
        "dev.num_gpus": 0, "dev.device_id": None, "dev.device": torch.device("cpu"), "dataset.workers": 0,
        "dataset.train_batch_size0": 8, "dataset.val_batch_size0": 8, "common.results_loc": out_dir, "common.exp_loc": out_dir + "/run_1",
        "scheduler.is_iteration_based": False, "scheduler.max_epochs": 1, "ema.enable": True,
        "sampler.name": "batch_sampler", "sampler.bs.crop_size_width": 32, "sampler.bs.crop_size_height": 32,
        "image_augmentation.mixup.enable": False, "image_augmentation.cutmix.enable": False,
    }.items():
        setattr(opts, k, v)
    if world > 1:  # the launcher's DDP branch (main_train.py:90-96 replaced) under a 2-rank gloo group, rendezvous by launch.distributed_init
        for k, v in {"ddp.use_distributed": True, "ddp.rank": rank, "ddp.world_size": world, "ddp.backend": "gloo",
                     "ddp.dist_url": f"tcp://127.0.0.1:{port}", "ddp.start_rank": 0}.items():
            setattr(opts, k, v)
        assert launch.distributed_init(opts) == rank
    create_directories(dir_path=out_dir + "/run_1", is_master_node=rank == 0)
    loop_check = os.environ.get("LAUNCH_LOOP_CHECK") == "1"
    if loop_check:  # no RNG inside the step, no EMA deep copy: the two runs below must agree bit for bit
        for k in ("model.classification.mit.dropout", "model.classification.mit.ffn_dropout", "model.classification.mit.attn_dropout",
                  "model.classification.classifier_dropout"):
            setattr(opts, k, 0.0)
        setattr(opts, "ema.enable", False)
        setattr(opts, "common.grad_clip", 10.0)
    torch.manual_seed(0)
    eng = launch.main(opts, swap=False, loader_factory=loader_factory)
    m = eng.model
    if loop_check:
        # the same model / loader / optimizer / scheduler, trained by the restated loop body (tests/engine_loop.py) instead of the Trainer
        import copy
        from cvnets import get_model
        from loss_fn import build_loss_fn
        from optim import build_optimizer
        from optim.scheduler import build_scheduler
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import engine_loop
        opts2 = copy.deepcopy(opts)
        torch.manual_seed(0)
        train_loader, _, _ = loader_factory(opts2)
        m2 = get_model(opts2)
        crit, opt2, sched = build_loss_fn(opts2), build_optimizer(m2, opts=opts2), build_scheduler(opts=opts2)
        n, losses = engine_loop.train_iterations(m2, crit, opt2, sched, torch.cuda.amp.GradScaler(enabled=False), train_loader, device="cpu",
                                                 max_norm=10.0)
        diff = max(float((a - b).abs().max()) for a, b in zip(m.state_dict().values(), m2.state_dict().values()) if a.dtype.is_floating_point)
        print("LOOP_JSON " + json.dumps({"updates": n, "losses": losses, "max_abs_param_diff": diff}))
    if world > 1:
        assert type(m).__module__ == "cvnets_amd.ddp", type(m)
        # Trainer.run tears the process group down; the test process compares the replicas (same averaged gradients -> bit-identical)
        torch.save(torch.cat([p.detach().reshape(-1) for p in m.parameters()]), f"{out_dir}/flat_{rank}.pt")
        m = m.module
        if rank != 0:
            return
    finite = all(torch.isfinite(p).all().item() for p in m.parameters())
    print("LAUNCH_JSON " + json.dumps({
        "engine": type(eng).__module__ + "." + type(eng).__name__, "model": type(m).__name__, "train_iterations": int(eng.train_iterations),
        "params_finite": bool(finite), "ema": eng.model_ema is not None,
        "checkpoints": sorted(f for f in os.listdir(out_dir + "/run_1") if f.endswith(".pt"))}))


def _worker(rank, world, port, out_dir):
    sys.argv = [sys.argv[0], out_dir]
    main(rank, world, port)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--ranks2":
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_worker, args=(2, port, sys.argv[1]), nprocs=2)
    else:
        main()
