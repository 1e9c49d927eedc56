"""microseconds and unique-operand TB/s of the weight-gradient product dW = dY^T X (cvh_gemm_dw_bias) on the token-linear shapes of
MobileViT-S at 1024 images, per kernel choice (CVH_TUNE key 22: 1 = the 128 x 128 tiles of gemm_big.hip, 0 = whole-row workgroups of
gemm_rows.hip; key 23 = cap on its LDS stages).   python tools/bench_dw.py [key=value ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-cvnets_amd"))
from cvnets_amd import _lib  # noqa: E402

DEV = "cuda:0"
SHAPES = [(1048576, 144, 144), (1048576, 288, 144), (1048576, 144, 288), (1048576, 432, 144), (262144, 192, 192), (262144, 384, 192), (262144, 576, 192)]


def main():
    global SHAPES
    if os.environ.get("DW_SHAPES"):
        SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["DW_SHAPES"].split(",")]
    sets = [a for a in sys.argv[1:]] or ["22=1", "22=0"]
    for (M, N, K) in SHAPES:
        dy = torch.randn(M, N, device=DEV).bfloat16()
        x = torch.randn(M, K, device=DEV).bfloat16()
        line = f"M{M} N{N} K{K}: "
        for kv in sets:
            for one in kv.split(","):
                k, v = one.split("=")
                _lib.call("cvh_set_tuning", int(k), int(v))
            n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, K)
            scr = torch.empty(n_scr, device=DEV)
            bp = torch.empty(n_scr // (N * K) * N, device=DEV)
            st = torch.cuda.current_stream().cuda_stream

            def run():
                _lib.call("cvh_gemm_dw_bias", 1, dy.data_ptr(), x.data_ptr(), None, K, 0, None, bp.data_ptr(), M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, K,
                          scr.data_ptr(), n_scr, 0, st)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            line += f"[{kv}] {us:7.1f} us {M * (N + K) * 2 / us / 1e6:5.2f} TB/s rows {n_scr // (N * K):3d} | "
        print(line, flush=True)


if __name__ == "__main__":
    main()
