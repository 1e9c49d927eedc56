"""Micro-benchmark of the dwx kernels (csrc/dwx.hip) at the InvertedResidual shapes of the 1024-image MobileViT-S step, next to the kernels
they replace (expansion GEMM + cvh_dwconv_bn_fwd / cvh_dwconv_bn_bwd): us per launch and GB/s of algorithmic traffic.
    python tools/bench_dwx.py [--batch 1024] [--reps 5] [--only new|old] [--shape i] [--dbg bits]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-cvnets_amd"))
from cvnets_amd import _lib, fused, ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # (H, W, Cin, hid, stride)   MobileViT-S @256: layer_1, layer_2 (3 blocks), layer_3, layer_4, layer_5
    (128, 128, 16, 64, 1), (128, 128, 32, 128, 2), (64, 64, 64, 256, 1), (64, 64, 64, 256, 2), (32, 32, 96, 384, 2), (16, 16, 128, 512, 2)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="both")
    ap.add_argument("--dbg", type=int, default=0, help="CVH_TUNE key 17: phase-skip bits of the dwx kernels (timing experiments; results wrong)")
    ap.add_argument("--shape", type=int, default=-1, help="index into SHAPES (-1: all)")
    ap.add_argument("--tune", default="", help="CVH_TUNE settings key=value[,key=value] (20=1: tile kernel of dwx.hip instead of the strip kernel)")
    a = ap.parse_args()
    B = a.batch
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("cvh_set_tuning", 17, a.dbg)
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        _lib.call("cvh_set_tuning", int(k), int(v))
    for (H, W, Cin, hid, s) in (SHAPES if a.shape < 0 else [SHAPES[a.shape]]):
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(B, H, W, Cin, device=DEV, generator=g).bfloat16()
        w1 = (torch.randn(hid, Cin, device=DEV, generator=g) * Cin ** -0.5).bfloat16()
        wd = (torch.randn(9, hid, device=DEV, generator=g) * 0.4).bfloat16()
        stats = torch.stack([torch.zeros(hid, device=DEV), torch.ones(hid, device=DEV), torch.ones(hid, device=DEV), torch.zeros(hid, device=DEV)]).contiguous()
        y2 = torch.empty(B, Ho, Wo, hid, device=DEV, dtype=torch.bfloat16)
        g2 = torch.randn(B, Ho, Wo, hid, device=DEV, generator=g).bfloat16()
        g1 = torch.empty(B, H, W, hid, device=DEV, dtype=torch.bfloat16)
        ca = torch.ones(hid, device=DEV); cb = torch.zeros(hid, device=DEV) + 0.1; cc = torch.zeros(hid, device=DEV)
        M1, M2 = B * H * W, B * Ho * Wo
        line = f"{H}x{W} Cin{Cin} hid{hid} s{s}: "
        if a.only in ("both", "new"):
            R = _lib.query("cvh_dwx_rows", B, Ho, Wo, hid, s)
            Rf = _lib.query("cvh_dwx_fwd_rows", B, H, W, Cin, hid, s)
            part = torch.empty(max(R, Rf), 2, hid, device=DEV); dwp = torch.empty(R, hid * 9, device=DEV)
            tf = timed(lambda: _lib.call("cvh_dwx_fwd", 1, x.data_ptr(), w1.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), 1, wd.data_ptr(),
                                         y2.data_ptr(), part.data_ptr(), B, H, W, Ho, Wo, Cin, hid, s, st), a.reps)
            tb = timed(lambda: _lib.call("cvh_dwx_bwd", 1, x.data_ptr(), w1.data_ptr(), stats.data_ptr(), 1, g2.data_ptr(), y2.data_ptr(), ca.data_ptr(),
                                         cb.data_ptr(), cc.data_ptr(), wd.data_ptr(), g1.data_ptr(), part.data_ptr(), dwp.data_ptr(), B, H, W, Ho, Wo,
                                         Cin, hid, s, st), a.reps)
            bf = (M1 * Cin + M2 * hid) * 2
            bb = (M1 * Cin + 2 * M2 * hid + M1 * hid) * 2
            line += f"dwx fwd {tf:7.0f} us ({bf / tf / 1e6:5.2f} TB/s)  bwd {tb:7.0f} us ({bb / tb / 1e6:5.2f} TB/s)  | "
        if a.only in ("both", "old"):
            y1 = torch.empty(B, H, W, hid, device=DEV, dtype=torch.bfloat16)
            R = _lib.query("cvh_dwconv_bn_rows", B, Ho, Wo, hid, s)
            part = torch.empty(R, 2, hid, device=DEV); dwp = torch.empty(R, hid * 9, device=DEV)
            tg = timed(lambda: fused._pw_gemm(x, None, Cin, w1, y1, M1, hid, want_stats=True), a.reps)
            tf = timed(lambda: _lib.call("cvh_dwconv_bn_fwd", 1, y1.data_ptr(), fused._xf(1, None, stats[2], stats[3], None, 1), wd.data_ptr(), y2.data_ptr(),
                                         B, H, W, Ho, Wo, hid, s, part.data_ptr(), st), a.reps)
            tb = timed(lambda: _lib.call("cvh_dwconv_bn_bwd", 1, g2.data_ptr(), fused._xf(2, y2, ca, cb, cc), y1.data_ptr(), stats.data_ptr(), 1,
                                         wd.data_ptr(), g1.data_ptr(), part.data_ptr(), dwp.data_ptr(), B, H, W, Ho, Wo, hid, s, st), a.reps)
            line += f"old gemm {tg:6.0f} + dwf fwd {tf:6.0f} us, dwf bwd {tb:6.0f} us"
        print(line, flush=True)


if __name__ == "__main__":
    ops.set_compute_dtype(torch.bfloat16)
    main()
