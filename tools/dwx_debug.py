"""Developer aid for csrc/dwx.hip: runs the forward / backward kernels on one small case per stride and prints WHERE they deviate from the
torch formulas (by tile position, parity class, channel-in-block, weight tap) instead of a bare max error.   python tools/dwx_debug.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-cvnets_amd"))
import test_dwx_gpu as T  # noqa: E402
from cvnets_amd import _lib  # noqa: E402

DEV = "cuda:0"


def summarize(name, got, ref, tol=1e-2):
    d = (got - ref).abs() / ref.abs().max()
    bad = d > tol
    print(f"{name}: max rel err {float(d.max()):.3e}, bad {int(bad.sum())}/{bad.numel()}")
    return bad


def run(B, H, W, Cin, hid, stride):
    print(f"=== B{B} {H}x{W} Cin{Cin} hid{hid} s{stride}")
    g, x, w1, wd, scale, shift = T._inputs(B, H, W, Cin, hid, 5)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y1, a1, ref = T._ref_forward(x, w1, wd, scale, shift, stride)
    R = _lib.query("cvh_dwx_rows", B, Ho, Wo, hid, stride)
    y2 = torch.full((B, Ho, Wo, hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.zeros(R, 2, hid, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("cvh_dwx_fwd", 1, x.data_ptr(), w1.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, wd.data_ptr(), y2.data_ptr(), part.data_ptr(),
              B, H, W, Ho, Wo, Cin, hid, stride, st)
    torch.cuda.synchronize()
    got = torch.nan_to_num(y2.float(), nan=1e9)
    bad = summarize("fwd y2", got, ref)
    if bad.any():
        print("  bad by ho:", bad.sum((0, 2, 3)).tolist())
        print("  bad by wo:", bad.sum((0, 1, 3)).tolist())
        print("  bad by ch%16:", bad.reshape(-1, hid // 8 // 2 if hid % 16 else hid // 16, 16).sum((0, 1)).tolist() if hid % 16 == 0 else "-")
        print("  bad by ch//16:", bad.reshape(-1, hid).sum(0).reshape(-1, 16).sum(1).tolist() if hid % 16 == 0 else "-")
    # backward
    mean = torch.randn(hid, device=DEV, generator=g) * 0.2
    invstd = torch.rand(hid, device=DEV, generator=g) + 0.7
    in_stats = torch.stack([mean, invstd, scale, shift]).contiguous()
    g2 = torch.randn(B, Ho, Wo, hid, device=DEV, generator=g).bfloat16()
    import torch.nn.functional as F
    y1r = y1.clone().requires_grad_(True)
    a = F.silu(y1r * scale + shift)
    ar = a.detach().bfloat16().float() - a.detach() + a
    wt = wd.float().t().reshape(hid, 1, 3, 3).clone().requires_grad_(True)
    out = F.conv2d(ar.permute(0, 3, 1, 2), wt, stride=stride, padding=1, groups=hid).permute(0, 2, 3, 1)
    out.backward(g2.float())
    g1_ref = y1r.grad / scale
    dw_ref = wt.grad.reshape(hid, 9)
    g1 = torch.full((B, H, W, hid), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.zeros(R, 2, hid, device=DEV)
    dwp = torch.zeros(R, hid * 9, device=DEV)
    _lib.call("cvh_dwx_bwd", 1, x.data_ptr(), w1.data_ptr(), in_stats.data_ptr(), 1, g2.data_ptr(), None, None, None, None, wd.data_ptr(),
              g1.data_ptr(), part.data_ptr(), dwp.data_ptr(), B, H, W, Ho, Wo, Cin, hid, stride, st)
    torch.cuda.synchronize()
    got = torch.nan_to_num(g1.float(), nan=1e9)
    bad = summarize("bwd g1", got, g1_ref)
    if bad.any():
        print("  bad by hi%8:", bad.sum((0, 2, 3)).reshape(-1)[: (H // 8) * 8].reshape(-1, 8).sum(0).tolist())
        print("  bad by wi%16:", bad.sum((0, 1, 3)).reshape(-1)[: (W // 16) * 16].reshape(-1, 16).sum(0).tolist())
        if hid % 16 == 0:
            print("  bad by ch%16:", bad.reshape(-1, hid).sum(0).reshape(-1, 16).sum(0).tolist())
    dw = dwp.sum(0).reshape(hid, 9)
    d = (dw - dw_ref).abs() / dw_ref.abs().max()
    print(f"bwd dW: max rel err {float(d.max()):.3e}; per tap max: {[round(float(v), 4) for v in d.max(0).values]}")
    if hid % 16 == 0:
        print("   per ch%16 max:", [round(float(v), 4) for v in d.max(1).values.reshape(-1, 16).max(0).values])
    s = part.sum(0)
    gf = g1.float().reshape(-1, hid)
    xh = ((y1 - mean) * invstd).reshape(-1, hid)
    print(f"bwd stats: s1 err {float((s[0] - gf.sum(0)).abs().max() / gf.abs().sum(0).max()):.2e}  s2 err "
          f"{float((s[1] - (gf * xh).sum(0)).abs().max() / (gf * xh).abs().sum(0).max()):.2e}")


if __name__ == "__main__":
    run(1, 16, 32, 64, 64, 1)
    run(1, 16, 32, 32, 128, 2)
    run(2, 21, 27, 64, 256, 1)
    run(2, 19, 23, 64, 256, 2)
