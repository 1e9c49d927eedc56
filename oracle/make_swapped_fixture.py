"""ORACLE helper — test infrastructure, NOT product code.

Executes the drop-in boundary the way INTEGRATION.md prescribes it, as far as this (GPU-less) container can: the REFERENCE builder
(`cvnets.get_model(opts)`, reference YAML) constructs the model, `cvnets_amd.dropin.swap_to_hip` class-swaps every module, and the
swapped object is pickled into tests/golden/ — so the GPU box (which has no /root/reference) unpickles a *reference-built* model
with only cvnets_amd importable and runs it through the HIP kernels (tests/test_dropin_gpu.py).  The pickle holds the reference's
own module tree, attribute values and opts namespace; parameters are the seeded values of oracle/weights.py (the same the golden
.npz fixtures were generated with), stored as zeros here and re-seeded after loading to keep the file small.

    python oracle/make_swapped_fixture.py            # writes tests/golden/swapped_mobilevit_{xxs,s}.pt, swapped_deeplabv3_s.pt, swapped_ssd_s.pt
"""
import os
import pickle
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))

from oracle.make_golden import build_reference_model  # noqa: E402  (puts the shim + /root/reference on sys.path)


def foreign_classes(obj, seen=None, path="model"):
    """every class reachable from the pickled object graph that lives in the reference tree (must be empty)"""
    out = set()
    for name, m in obj.named_modules():
        mod = m.__class__.__module__
        if mod.startswith("cvnets.") or mod.startswith("options") or mod.startswith("utils"):
            out.add(f"{name}:{m.__class__.__module__}.{m.__class__.__name__}")
    return out


def main():
    from cvnets_amd import dropin
    cwd = os.getcwd()
    from oracle.make_golden import build_reference_segmentation, build_reference_ssd
    for tag, mode in (("xxs", "xx_small"), ("s", "small"), ("deeplabv3_s", None), ("ssd_s", None)):
        model = build_reference_model(mode) if mode else (build_reference_ssd() if tag == "ssd_s" else build_reference_segmentation())
        if tag == "ssd_s":
            model.match_prior = None  # host-side box matcher (loss / eval post-processing): not a module, stays on the reference side
            model.opts = None
        os.chdir(cwd)
        counts, left = dropin.swap_to_hip(model, strict=True)
        bad = foreign_classes(model)
        assert not bad, bad
        for p in model.parameters():  # values are re-seeded after loading (oracle.weights): keep the fixture tiny
            p.data = torch.zeros(0)
        for b in model.buffers():
            b.data = torch.zeros(0, dtype=b.dtype)
        shapes = {}
        path = os.path.join(REPO, "tests", "golden", f"swapped_mobilevit_{tag}.pt" if mode else f"swapped_{tag}.pt")
        blob = pickle.dumps(model)
        assert b"cvnets." not in blob.replace(b"cvnets_amd.", b""), "a reference class leaked into the pickle"
        open(path, "wb").write(blob)
        print(path, len(blob), "bytes", counts)


if __name__ == "__main__":
    main()
