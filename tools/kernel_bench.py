#!/usr/bin/env python
"""Per-kernel roofline table at the layer shapes of MobileViT-S 256x256, B images per GPU, bf16 (the bench.py workload).

    python tools/kernel_bench.py [--batch 128] [--reps 10] [--tune KEY=v1,v2 ...] [--only gemm,dw,tn,attn,bn]

Every row: one kernel launch through the C ABI at one real layer shape, timed with HIP events on the launch stream;
algorithmic bytes = the tensors the launch must read + write once (weights/statistics neglected); GB/s = bytes / time.
`--tune KEY=a,b` re-times everything under each value of a cvh_set_tuning knob in the SAME process (interleaved A/B), which is
the only trustworthy way to compare kernel variants: box-to-box variance on the pool is larger than most deltas.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))

import torch  # noqa: E402

from cvnets_amd import _lib, ops  # noqa: E402

DT = torch.bfloat16
ES = 2

# (name, Cin, Cout, k, stride, H_in) for groups == 1 convs ; token linears as (name, rows_per_img, K, N)
CONVS = [
    ("conv_1 3x3s2", 8, 16, 3, 2, 256),
    ("l1 exp", 16, 64, 1, 1, 128), ("l1 red", 64, 32, 1, 1, 128),
    ("l2.0 exp", 32, 128, 1, 1, 128), ("l2.0 red", 128, 64, 1, 1, 64),
    ("l2.1 exp", 64, 256, 1, 1, 64), ("l2.1 red", 256, 64, 1, 1, 64),
    ("l3.0 red", 256, 96, 1, 1, 32),
    ("l3 3x3", 96, 96, 3, 1, 32), ("l3 1x1in", 96, 144, 1, 1, 32), ("l3 proj", 144, 96, 1, 1, 32), ("l3 fusion", 192, 96, 3, 1, 32),
    ("l4.0 exp", 96, 384, 1, 1, 32), ("l4.0 red", 384, 128, 1, 1, 16),
    ("l4 3x3", 128, 128, 3, 1, 16), ("l4 fusion", 256, 128, 3, 1, 16),
    ("l5.0 exp", 128, 512, 1, 1, 16), ("l5 fusion", 320, 160, 3, 1, 8), ("exp 640", 160, 640, 1, 1, 8),
]
LINEARS = [
    ("l3 qkv", 1024, 144, 432), ("l3 out", 1024, 144, 144), ("l3 fc1", 1024, 144, 288), ("l3 fc2", 1024, 288, 144),
    ("l4 qkv", 256, 192, 576), ("l4 fc1", 256, 192, 384), ("l5 qkv", 64, 240, 720), ("l5 fc2", 64, 480, 240), ("fc", 1, 640, 1000),
]
DWS = [("l1 dw", 64, 128, 1), ("l2.0 dw", 128, 128, 2), ("l2.1 dw", 256, 64, 1), ("l3.0 dw", 256, 64, 2), ("l4.0 dw", 384, 32, 2), ("l5.0 dw", 512, 16, 2)]
ATTN = [("l3 attn", 32, 144, 4), ("l4 attn", 16, 192, 4), ("l5 attn", 8, 240, 4)]  # (name, H=W of the map, d, heads)


def timeit(fn, reps):
    st = torch.cuda.current_stream()
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def rnd(*shape):
    return torch.randn(*shape, device="cuda").to(DT)


def bench_all(B, reps, only):
    rows = []
    s = ops._stream

    def add(kind, name, shape, us, nbytes):
        rows.append((kind, name, shape, us, nbytes))

    if "gemm" in only or "tn" in only:
        for name, Cin, Cout, k, stride, H in CONVS:
            Ho = H // stride
            pad = (k - 1) // 2
            x = rnd(B * H * H, Cin)
            w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
            wp = ops.pack_weight(w, DT, 0)
            y = torch.empty(B * Ho * Ho, Cout, device="cuda", dtype=DT)
            M = B * Ho * Ho
            R = _lib.query("cvh_conv_gemm_grid_rows", M, Cout)
            part = torch.empty(R * 2 * Cout, device="cuda")
            if "gemm" in only:
                us = timeit(lambda: ops._conv_gemm(x, None, Cin, 0, wp, y, B, H, H, Ho, Ho, k, k, stride, pad, 1, Cout, stats_part=part), reps)
                add("conv_gemm fwd", name, f"M={M} K={k * k * Cin} N={Cout}", us, (x.numel() + y.numel()) * ES)
                if stride == 1:
                    wpt = ops.pack_weight(w, DT, 1)
                    dx = torch.empty_like(x)
                    us = timeit(lambda: ops._conv_gemm(y, None, Cout, 0, wpt, dx, B, Ho, Ho, H, H, k, k, 1, k - 1 - pad, 1, Cin), reps)
                    add("conv_gemm dX", name, f"M={M} K={k * k * Cout} N={Cin}", us, (x.numel() + y.numel()) * ES)
            if "tn" in only:
                dw = torch.zeros(Cout, Cin, k, k, device="cuda")
                n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, Cout, k * k * Cin)
                scr = torch.empty(max(n_scr, 1), device="cuda")
                us = timeit(lambda: _lib.call("cvh_gemm_dw", 1, y.data_ptr(), x.data_ptr(), None, Cin, 0, dw.data_ptr(), B, H, H, Ho, Ho, k, k,
                                               stride, pad, 1, Cout, Cin, scr.data_ptr(), n_scr, 0, s()), reps)
                add("gemm_tn dW", name, f"M={M} N={Cout} K={k * k * Cin}", us, (x.numel() + y.numel()) * ES)
        for name, rpi, K, N in LINEARS:
            M = B * rpi
            x, w = rnd(M, K), torch.randn(N, K, device="cuda") * 0.05
            wp = ops.pack_weight(w, DT, 0)
            y = torch.empty(M, N, device="cuda", dtype=DT)
            if "gemm" in only:
                us = timeit(lambda: ops._conv_gemm(x, None, K, 0, wp, y, M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N), reps)
                add("conv_gemm fwd", name, f"M={M} K={K} N={N}", us, (x.numel() + y.numel()) * ES)
            if "tn" in only:
                dw = torch.zeros(N, K, device="cuda")
                n_scr = _lib.query("cvh_gemm_dw_scratch_elems", M, N, K)
                scr = torch.empty(max(n_scr, 1), device="cuda")
                us = timeit(lambda: _lib.call("cvh_gemm_dw", 1, y.data_ptr(), x.data_ptr(), None, K, 0, dw.data_ptr(), M, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, K,
                                               scr.data_ptr(), n_scr, 0, s()), reps)
                add("gemm_tn dW", name, f"M={M} N={N} K={K}", us, (x.numel() + y.numel()) * ES)
    if "dw" in only:
        for name, C, H, stride in DWS:
            Ho = H // stride
            x, y = rnd(B * H * H, C), rnd(B * Ho * Ho, C)
            w = torch.randn(C, 1, 3, 3, device="cuda") * 0.3
            wp = ops.pack_weight(w, DT, 2)
            R = _lib.query("cvh_dwconv_rows", B, Ho, Ho, C, 3, stride, 1, 1)
            part = torch.empty(R * 2 * C, device="cuda")
            Rw = _lib.query("cvh_dwconv_bwd_w_rows", B, Ho, Ho, C, 3, stride, 1, 1)
            partw = torch.empty(Rw * 9 * C, device="cuda")
            nb = (x.numel() + y.numel()) * ES
            us = timeit(lambda: _lib.call("cvh_dwconv_fwd", 1, x.data_ptr(), wp.data_ptr(), y.data_ptr(), B, H, H, Ho, Ho, C, 3, stride, 1, 1,
                                           part.data_ptr(), s()), reps)
            add("dwconv fwd", name, f"C={C} {H}->{Ho}", us, nb)
            us = timeit(lambda: _lib.call("cvh_dwconv_bwd_x", 1, y.data_ptr(), wp.data_ptr(), x.data_ptr(), B, H, H, Ho, Ho, C, 3, stride, 1, 1, s()), reps)
            add("dwconv dX", name, f"C={C} {H}->{Ho}", us, nb)
            us = timeit(lambda: _lib.call("cvh_dwconv_bwd_w", 1, x.data_ptr(), y.data_ptr(), partw.data_ptr(), B, H, H, Ho, Ho, C, 3, stride, 1, 1, s()), reps)
            add("dwconv dW", name, f"C={C} {H}->{Ho}", us, nb)
    if "attn" in only:
        for name, H, d, h in ATTN:
            rows_ = B * H * H
            qkv, o, do = rnd(rows_, 3 * d), torch.empty(rows_, d, device="cuda", dtype=DT), rnd(rows_, d)
            nseq, S, c = B * 4, H * H // 4, d // h
            lse, dsum = torch.empty(nseq * h * S, device="cuda"), torch.empty(nseq * h * S, device="cuda")
            dqkv = torch.empty_like(qkv)
            args = (nseq, S, h, c, 2, 2, H // 2, H, H, float(c) ** -0.5, 0)
            us = timeit(lambda: _lib.call("cvh_attn_fwd", 1, qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), None, *args, s()), reps)
            flops = 4 * nseq * h * S * S * c
            add("attn fwd", name, f"nseq={nseq} S={S} c={c} ({flops / us / 1e6:.1f} TF/s)", us, (qkv.numel() + o.numel()) * ES)
            us = timeit(lambda: _lib.call("cvh_attn_bwd", 1, qkv.data_ptr(), o.data_ptr(), do.data_ptr(), dqkv.data_ptr(), lse.data_ptr(),
                                           dsum.data_ptr(), None, *args, s()), reps)
            add("attn bwd", name, f"nseq={nseq} S={S} c={c} ({2.5 * flops / us / 1e6:.1f} TF/s)", us, (2 * qkv.numel() + 2 * o.numel()) * ES)
    if "bn" in only:
        for name, C, H in [("l1 exp out", 64, 128), ("l2.1 exp out", 256, 64), ("l3 fusion out", 96, 32)]:
            rows_ = B * H * H
            x, do, y = rnd(rows_, C), rnd(rows_, C), torch.empty(rows_, C, device="cuda", dtype=DT)
            st = torch.rand(8, C, device="cuda") + 0.5
            R = _lib.query("cvh_colreduce_rows", rows_, C)
            part = torch.empty(R * 2 * C, device="cuda")
            us = timeit(lambda: _lib.call("cvh_bn_apply", 1, x.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1, None, y.data_ptr(), rows_, C, s()), reps)
            add("bn_apply", name, f"rows={rows_} C={C}", us, 2 * x.numel() * ES)
            us = timeit(lambda: _lib.call("cvh_bn_bwd_reduce", 1, x.data_ptr(), do.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
                                           st[3].data_ptr(), 1, rows_, C, part.data_ptr(), s()), reps)
            add("bn_bwd_reduce", name, f"rows={rows_} C={C}", us, 2 * x.numel() * ES)
            us = timeit(lambda: _lib.call("cvh_bn_bwd_apply", 1, x.data_ptr(), do.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), 1, st[2].data_ptr(),
                                           st[3].data_ptr(), st[4].data_ptr(), y.data_ptr(), rows_, C, s()), reps)
            add("bn_bwd_apply", name, f"rows={rows_} C={C}", us, 3 * x.numel() * ES)
            co = torch.empty(5, C, device="cuda")
            us = timeit(lambda: _lib.call("cvh_bn_bwd_finalize", part.data_ptr(), R, C, float(rows_), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
                                           1, 0, co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), co[4].data_ptr(), s()), reps)
            add("bn_bwd_finalize", name, f"C={C}", us, R * 2 * C * 4)
            us = timeit(lambda: _lib.call("cvh_bn_finalize", part.data_ptr(), R, C, float(rows_), st[0].data_ptr(), st[1].data_ptr(), st[5].data_ptr(),
                                           st[6].data_ptr(), 0.1, 1e-5, co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), s()), reps)
            add("bn_finalize", name, f"C={C}", us, R * 2 * C * 4)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="gemm,tn,dw,attn,bn")
    ap.add_argument("--tune", action="append", default=[], help="KEY=v1,v2 (cvh_set_tuning key, values to A/B)")
    a = ap.parse_args()
    only = set(a.only.split(","))
    variants = [("default", [])]
    for t in a.tune:
        k, vs = t.split("=")
        variants = [(f"{k}={v}", [(int(k), int(v))]) for v in vs.split(",")]
    results = {}
    for rnd_i in range(2 if len(variants) > 1 else 1):  # interleave variants twice, keep the min
        for vname, sets in variants:
            for k, v in sets:
                _lib.call("cvh_set_tuning", k, v)
            for r in bench_all(a.batch, a.reps, only):
                key = (r[0], r[1], r[2])
                results.setdefault(key, {})
                results[key][vname] = min(results[key].get(vname, 1e30), r[3])
                results[key]["bytes"] = r[4]
    names = [v[0] for v in variants]
    print(f"{'kernel':15s} {'layer':12s} {'shape':46s} " + " ".join(f"{n + ' us':>14s} {'GB/s':>7s}" for n in names))
    tot = {n: 0.0 for n in names}
    for (kind, name, shape), d in results.items():
        line = f"{kind:15s} {name:12s} {shape:46s} "
        for n in names:
            line += f"{d[n]:14.1f} {d['bytes'] / d[n] / 1e3:7.0f} "
            tot[n] += d[n]
        print(line)
    print("total us: " + "  ".join(f"{n}: {tot[n]:.0f}" for n in names))


if __name__ == "__main__":
    main()
