import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "ml-cvnets_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests call the HIP library; skip them (not fail) when no GPU is visible so that a plain `pytest tests`
    on the CPU container stays green."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


if os.environ.get("CVH_GUARD_ALLOC"):
    # debugging aid (tools/guard_alloc.cpp): every device allocation in its own address range with unmapped pages on both sides and never
    # reused after free — an out-of-bounds access or a launch through a stale pointer faults at the launch that does it
    import torch

    torch.cuda.memory.change_current_allocator(
        torch.cuda.memory.CUDAPluggableAllocator(os.path.abspath(os.environ["CVH_GUARD_ALLOC"]), "guard_malloc", "guard_free"))

    def _no_capture(*args, **kwargs):
        pytest.skip("no hipGraph capture under the guard allocator")

    torch.cuda.CUDAGraph = _no_capture


if os.environ.get("CVH_TEST_NAN_FILL"):
    # debugging aid: every torch.empty() buffer starts as NaN (floats) / max-int, so a kernel that reads memory it was supposed to
    # write first poisons the result deterministically instead of depending on what the caching allocator handed out
    import torch

    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
