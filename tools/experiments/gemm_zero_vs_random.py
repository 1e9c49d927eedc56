"""Is the transformer-sized GEMM limited by the chip's power budget?  The same kernel on zero-filled and on random operands (MI355X_MICROARCH.md:
a power-limited kernel clocks higher on zeros).   python tools/experiments/gemm_zero_vs_random.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "ml-cvnets_amd"))
from cvnets_amd import ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


ops.set_compute_dtype(torch.bfloat16)
for (M, K, N) in [(100864, 768, 2304), (100864, 3072, 768), (100864, 768, 768)]:
    fl = 2.0 * M * K * N
    line = f"M{M} K{K} N{N}: "
    for kind in ("random", "zeros", "random", "zeros"):
        x = (torch.randn(M, K, device=DEV) if kind == "random" else torch.zeros(M, K, device=DEV)).bfloat16()
        w = torch.randn(N, K, device=DEV) * K ** -0.5 if kind == "random" else torch.zeros(N, K, device=DEV)
        wb = w.bfloat16()
        with torch.no_grad():
            t = timed(lambda: ops.linear(x, w, None))
            tt = timed(lambda: torch.nn.functional.linear(x, wb))
        line += f"{kind} {fl / t / 1e12:5.0f} (torch {fl / tt / 1e12:5.0f})  "
    print(line, flush=True)
