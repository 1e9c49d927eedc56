"""Race screen for the dW kernels (their LDS reads are ordered by the kernels' own waits since round 6, not by compiler-placed drains): the weight
and bias gradient of the same linear layer, repeated, must be bit-identical (fixed-order reductions) and equal to the fp32 matmul within bf16
rounding.   python tools/experiments/dw_determinism.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "ml-cvnets_amd"))
from cvnets_amd import ops  # noqa: E402

DEV = "cuda:0"
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SHAPES = [(100864, 768, 2304), (100864, 3072, 768), (100864, 768, 768), (19712, 512, 2048),  # gemm_tn256
          (262144, 192, 576), (65536, 240, 720), (1048576, 144, 432),  # gemm_tn128
          (1048576, 144, 144), (1048576, 144, 288), (262144, 192, 192), (65536, 240, 480), (50000, 256, 256)]  # gemm_tn_rows (incl. the 4 x 4 rectangle)
bad = 0
for (M, K, N) in SHAPES:
    g = torch.Generator(device=DEV).manual_seed(M % 1000 + K + N)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).requires_grad_(True)
    b = torch.zeros(N, device=DEV).requires_grad_(True)
    go = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    ref_w = ref_b = None
    worst = 0.0
    for r in range(REPS):
        w.grad = None
        b.grad = None
        y = ops.linear(x, w, b)
        y.backward(go)
        torch.cuda.synchronize()
        if ref_w is None:
            ref_w, ref_b = w.grad.clone(), b.grad.clone()
            chunk = 65536  # fp32 reference of dW on a slice of the rows would not equal the full sum: use the full product in fp32, chunked
            acc = torch.zeros(N, K, device=DEV)
            for i in range(0, M, chunk):
                acc += go[i:i + chunk].float().t() @ x[i:i + chunk].float()
            worst = float((ref_w - acc).norm() / acc.norm())
        else:
            if not (torch.equal(ref_w, w.grad) and torch.equal(ref_b, b.grad)):
                bad += 1
                print(f"  M{M} K{K} N{N}: repeat {r} differs: dW max abs {float((ref_w - w.grad).abs().max()):.3e}")
    print(f"M{M} K{K} N{N}: {REPS} repeats bit-identical: {bad == 0}; rel-L2 of dW against the fp32 product {worst:.2e}", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
