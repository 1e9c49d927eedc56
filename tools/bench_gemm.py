"""TFLOP/s of the transformer-sized linear layers (ViT-B / CLIP shapes) through ops.linear: forward GEMM per kernel choice
(CVH_TUNE key 5: 1 = 256 x 256 four-stage, 3 = 128 x 128 two-stage, 2 = 256 x 128) and the same products through torch.matmul (the vendor
library) on the same box.   python tools/bench_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-cvnets_amd"))
from cvnets_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [(100864, 768, 2304), (100864, 768, 768), (100864, 768, 3072), (100864, 3072, 768), (19712, 512, 1536), (19712, 512, 2048), (19712, 2048, 512)]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ops.set_compute_dtype(torch.bfloat16)
    for (M, K, N) in SHAPES:
        x = torch.randn(M, K, device=DEV).bfloat16()
        w = torch.randn(N, K, device=DEV) * K ** -0.5
        wb = w.bfloat16()
        b = torch.randn(N, device=DEV)
        fl = 2.0 * M * K * N
        line = f"M{M} K{K} N{N}: "
        with torch.no_grad():
            timed(lambda: ops.linear(x, w, b), reps=20)  # the first candidate of a row otherwise pays the clock / cache ramp of the new shape (measured: -15 %)
            for knob, name in ((1, "nt256"), (3, "nt128"), (2, "nt256x128")):
                _lib.call("cvh_set_tuning", 5, knob)
                t = timed(lambda: ops.linear(x, w, b))
                line += f"{name} {fl / t / 1e12:6.0f} TF  "
            _lib.call("cvh_set_tuning", 5, 1)
            t = timed(lambda: torch.nn.functional.linear(x, wb, b.bfloat16()))
            line += f"| torch {fl / t / 1e12:6.0f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
