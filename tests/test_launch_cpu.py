"""SURVEY.md §8b / BASELINE.json configs[0]: the reference's UNMODIFIED engine (`engine/training_engine.py` Trainer, options parser,
samplers, collate functions, optimizer / scheduler / loss builders, checkpointing, EMA) drives a model through `cvnets_amd.launch`
— the launcher that replaces main_train.py — on CPU: MobileViT-XXS 32x32 fp32, batch 8, world size 1 (and 2 over gloo).

The swap is disabled here (no GPU in this container, and the HIP path has no CPU fallback): what is tested is the launcher's
control flow around the reference engine.  The reference tree is imported from /root/reference through the test-side import shims
(oracle/ref_shim); every test runs in its own process so the reference's top-level packages (`tests`, `utils`, `data` ...) do not leak
into pytest.  Skipped where the reference tree does not exist (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest
import torch

REF = os.environ.get("CVNETS_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "engine")), reason="reference tree not present")


def _run(tmp_path, *extra, env=None):
    e = dict(os.environ)
    e.pop("PYTHONPATH", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, "launch_cpu_driver.py"), str(tmp_path), *extra], capture_output=True, text=True,
                       timeout=900, env=e)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = {}
    for line in r.stdout.splitlines():
        for tag in ("LAUNCH_JSON ", "LOOP_JSON "):
            if line.startswith(tag):
                out[tag.strip()] = json.loads(line[len(tag):])
    return out


def test_reference_trainer_runs_through_launcher(tmp_path):
    res = _run(tmp_path)["LAUNCH_JSON"]
    assert res["engine"] == "engine.training_engine.Trainer" and res["model"] == "MobileViT"
    assert res["train_iterations"] == 4          # 32 samples / batch 8, one epoch
    assert res["params_finite"] and res["ema"]
    assert {"checkpoint_last.pt", "checkpoint_ema_last.pt", "training_checkpoint_last.pt"} <= set(res["checkpoints"])


def test_launcher_ddp_branch_two_ranks_gloo(tmp_path):
    """main_train.py:90-96 replaced by cvnets_amd.ddp.DistributedDataParallel, rendezvous by launch.distributed_init (gloo, 2 ranks)."""
    res = _run(tmp_path, "--ranks2")["LAUNCH_JSON"]
    assert res["train_iterations"] == 2          # 32 samples / (2 ranks x batch 8)
    a, b = (torch.load(os.path.join(tmp_path, f"flat_{r}.pt")) for r in (0, 1))
    assert a.numel() > 1_000_000 and torch.equal(a, b)  # averaged gradients -> bit-identical replicas


def test_restated_engine_loop_equals_reference_trainer(tmp_path):
    """tests/engine_loop.py (the loop the GPU tests drive, the GPU box having no reference tree) == Trainer.train_epoch, bit for bit."""
    res = _run(tmp_path, env={"LAUNCH_LOOP_CHECK": "1"})["LOOP_JSON"]
    assert res["updates"] == 4 and res["max_abs_param_diff"] == 0.0
