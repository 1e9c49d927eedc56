"""Developer aid: one fused InvertedResidual step with the y1-recomputing kernels (CVH_IR_X=1) and without, comparing the tensors that
leave every kernel of the backward pass (recorded by wrapping _lib.call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-cvnets_amd"))
from cvnets_amd import _lib, fused, layers, ops  # noqa: E402
from cvnets_amd.modules import InvertedResidual  # noqa: E402

DEV = "cuda:0"
Cin, Cout, stride, hw = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 32, 1, 40))]
opts = layers.default_opts()
x = torch.randn(6, Cin, hw, hw + 8, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
hid = 4 * Cin
B, H, W = 6, hw, hw + 8
orig_call = _lib.call
rec = {}


def wrap(mode):
    def call(name, *args):
        rc = orig_call(name, *args)
        torch.cuda.synchronize()
        if name in ("cvh_dwx_bwd", "cvh_dwconv_bn_bwd"):
            # g_in pointer position: dwx_bwd arg 11, dwconv_bn_bwd arg 7;  stats_part 12 / 8
            gi = args[11] if name == "cvh_dwx_bwd" else args[7]
            sp = args[12] if name == "cvh_dwx_bwd" else args[8]
            rec[(mode, "g1_ptr")] = gi
            rec[(mode, "part_ptr")] = sp
        rec.setdefault((mode, "calls"), []).append(name)
        return rc
    return call


res = {}
go = None
for mode in ("1", "fwd", "0"):
    torch.manual_seed(5)
    m = InvertedResidual(opts, Cin, Cout, stride=stride, expand_ratio=4).to(DEV).train()
    xin = x.clone().requires_grad_(True)
    fused._IR_X = mode
    _lib.call = wrap(mode)
    fused._lib.call = _lib.call
    ops.set_compute_dtype(torch.bfloat16)
    keep = []
    orig_empty = ops.nhwc_empty

    def nhwc_empty(*a, **k):
        t = orig_empty(*a, **k)
        keep.append(t)
        return t
    ops.nhwc_empty = nhwc_empty
    fused.ops.nhwc_empty = nhwc_empty
    out = m(xin)
    if go is None:
        go = torch.randn_like(out.float()).to(out.dtype)
    out.backward(go)
    ops.finish_backward()
    torch.cuda.synchronize()
    ops.nhwc_empty = orig_empty
    ops.set_compute_dtype(None)
    g1 = [t for t in keep if t.data_ptr() == rec[(mode, "g1_ptr")]]
    res[mode] = {"out": out.detach().float(), "dx": xin.grad.float(), "g1": g1[0].float().clone() if g1 else None,
                 "params": {n: p.grad.float().clone() for n, p in m.named_parameters()}}
    print(mode, "calls in step:", rec[(mode, "calls")])
_lib.call = orig_call
fused._lib.call = orig_call
for mode in ("1", "fwd"):
    print(f"--- mode {mode} vs 0")
    for k in ("out", "dx", "g1"):
        a, b = res[mode][k], res["0"][k]
        if a is None or b is None:
            print(k, "missing")
            continue
        print(f"  {k}: rel max err {float((a - b).abs().max() / b.abs().max()):.3e}")
    for n in res["0"]["params"]:
        a, b = res[mode]["params"][n], res["0"]["params"][n]
        print(f"  {n}: rel max err {float((a - b).abs().max() / (b.abs().max() + 1e-12)):.3e}")
