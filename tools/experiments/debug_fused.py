"""step-by-step run of the fused InvertedResidual launches with a device sync after each (developer aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
import torch
import cvnets_amd
from cvnets_amd import _lib, fused, ops

orig_call = _lib.call
def traced(name, *args):
    print("launch", name, flush=True)
    rc = orig_call(name, *args)
    torch.cuda.synchronize()
    print("   ok", name, flush=True)
    return rc
_lib.call = traced
fused._lib.call = traced
ops._lib.call = traced

from cvnets_amd.layers import default_opts
from cvnets_amd.modules import InvertedResidual
dtype = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.bfloat16
cvnets_amd.set_compute_dtype(dtype)
m = InvertedResidual(default_opts(), 32, 32, stride=1, expand_ratio=4).cuda().train()
x = ops.to_nhwc(torch.randn(2, 32, 16, 16, device="cuda"), dtype).requires_grad_(True)
y = m(x)
print("fwd done", float(y.float().abs().mean()))
y.float().square().mean().backward()
torch.cuda.synchronize()
print("bwd done", float(x.grad.float().abs().mean()))
