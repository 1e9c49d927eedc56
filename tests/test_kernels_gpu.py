"""GPU parity tests, one HIP operator at a time, forward AND backward, against a plain PyTorch fp32 reference of the
same op evaluated on the same (already dtype-rounded) inputs.  Tolerances: see tests/util.py (fp32 2e-4, bf16 2e-2,
relative to the reference's max magnitude).  All calls go through the C ABI (ctypes) of libcvnets_hip.so."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import check, l2_err, nhwc, rel_err

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from cvnets_amd import ops as _ops
    return _ops


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(_dev())


def _act_ref(x, act):
    return {0: lambda t: t, 1: F.silu, 2: F.gelu}[act](x)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_roundtrip(ops, dtype):
    x = _rand(3, 3, 10, 12)
    y = ops.to_nhwc(x, dtype)
    assert y.shape == (3, 8, 10, 12) and ops.is_nhwc(y)
    ref = torch.zeros(3, 8, 10, 12, device=_dev())
    ref[:, :3] = x
    check("nchw->nhwc", y, ref.to(dtype), dtype)
    back = ops.nhwc_to_nchw_f32(y, 3)
    check("nhwc->nchw", back, x.to(dtype).float(), dtype)


# ------------------------------------------------------------------------------------------------
# the MFMA layout check of the CDNA guide: identity-like A, ASYMMETRIC B (catches row/col swaps)
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric(ops, dtype):
    M, K, N = 96, 64, 40
    x = torch.zeros(M, K, device=_dev())
    for i in range(M):
        x[i, i % K] = 1.0 + (i // K)
    w = (torch.arange(N * K, device=_dev(), dtype=torch.float32).reshape(N, K) % 61) / 8.0 - 3.0  # exactly representable in bf16
    y = ops.linear(x.to(dtype), w)
    ref = x.to(dtype).float() @ w.to(dtype).float().t()
    check("asymmetric gemm", y, ref, dtype)


LINEAR_SHAPES = [(128, 16, 64), (1000, 64, 32), (257, 144, 432), (4096, 288, 144), (64, 640, 1000), (513, 240, 720), (300, 96, 96),
                 (77, 32, 128), (20000, 32, 128), (130, 384, 192),
                 # transformer-sized linears: bf16 takes the 128x128 direct-to-LDS kernel (csrc/gemm_big.hip) for fwd AND dX
                 (2500, 768, 768), (2048, 256, 1024), (4100, 3072, 768), (3000, 512, 2304),
                 # N % 256 == 0, K % 64 == 0 (K > 64) and >= 1024 tiles of 256 x 256 (csrc/gemm_big.hip gemm_big_eligible): gemm_nt256_kernel,
                 # rows not a multiple of the tile — 129 x 12, 344 x 3, 157 x 8, 547 x 2 tiles
                 (33000, 768, 3072), (88000, 3072, 768), (40000, 512, 2048), (140000 + 5, 128, 512)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N", LINEAR_SHAPES)
def test_linear_fwd_bwd(ops, dtype, M, K, N):
    x = _rand(M, K, seed=1).to(dtype).requires_grad_(True)
    w = _rand(N, K, seed=2, scale=1 / math.sqrt(K)).requires_grad_(True)
    b = _rand(N, seed=3, scale=0.1).requires_grad_(True)
    res = _rand(M, N, seed=4).to(dtype).requires_grad_(True)
    for act in (0, 1, 2):
        y = ops.linear(x, w, b, act=act, residual=res)
        wr = w.detach().to(dtype).float().requires_grad_(True)
        xr = x.detach().float().requires_grad_(True)
        br = b.detach().clone().requires_grad_(True)
        rr = res.detach().float().requires_grad_(True)
        pre = xr @ wr.t() + br
        if dtype == torch.bfloat16:
            pre = pre  # kernel applies act on the fp32 accumulator; pre-act is only ROUNDED for the saved copy
        ref = _act_ref(pre, act) + rr
        check(f"linear fwd act{act}", y, ref, dtype)
        go = _rand(M, N, seed=5).to(dtype)
        gx, gw, gb, gr = torch.autograd.grad(y, [x, w, b, res], go)
        rx, rw, rb, rres = torch.autograd.grad(ref, [xr, wr, br, rr], go.float())
        check(f"linear dx act{act}", gx, rx, dtype, scale=2)
        check(f"linear dw act{act}", gw, rw, dtype, scale=2)
        check(f"linear db act{act}", gb, rb, dtype, scale=2)
        check(f"linear dres act{act}", gr, rres, dtype)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, dil
    (2, 8, 16, 16, 16, 3, 2, 1),     # stem-like (input padded 3->8 below handled separately)
    (2, 16, 12, 12, 64, 1, 1, 1),
    (3, 96, 8, 8, 96, 3, 1, 1),
    (2, 32, 9, 11, 48, 3, 1, 1),     # odd spatial, N=48
    (2, 160, 8, 8, 640, 1, 1, 1),
    (2, 64, 10, 10, 64, 3, 1, 2),    # dilation 2
    (1, 144, 32, 32, 96, 1, 1, 1),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,dil", CONV_CASES)
@pytest.mark.parametrize("use_bn", [False, True])
def test_conv_bn_act(ops, dtype, B, Cin, H, W, Cout, k, stride, dil, use_bn):
    pad = (k - 1) // 2 * dil
    x0 = _rand(B, Cin, H, W, seed=1)
    x = nhwc(x0, dtype).requires_grad_(stride == 1)
    w = _rand(Cout, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k)).requires_grad_(True)
    g = (1 + 0.1 * _rand(Cout, seed=3)).requires_grad_(True)
    be = (0.1 * _rand(Cout, seed=4)).requires_grad_(True)
    rm = 0.1 * _rand(Cout, seed=5)
    rv = 1 + 0.1 * _rand(Cout, seed=6).abs()
    rm0, rv0 = rm.clone(), rv.clone()
    act = 1
    if use_bn:
        y = ops.conv_bn_act(x, w, None, g, be, rm, rv, stride=stride, pad=pad, dil=dil, act=act, use_bn=True, training=True)
    else:
        y = ops.conv_bn_act(x, w, None, stride=stride, pad=pad, dil=dil, act=act)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    c = F.conv2d(xr, wr, None, stride=stride, padding=pad, dilation=dil)
    gr, ber = g.detach().clone().requires_grad_(True), be.detach().clone().requires_grad_(True)
    if use_bn:
        if dtype == torch.bfloat16:
            c = c + (c.to(dtype).float() - c).detach()  # BN sees the conv output as stored (bf16), gradient straight-through
        c = F.batch_norm(c, rm0, rv0, gr, ber, training=True, momentum=0.1, eps=1e-5)
    ref = F.silu(c)
    assert ops.is_nhwc(y)
    check("conv fwd", y, ref, dtype)
    if use_bn:
        check("running_mean", rm, rm0, torch.float32, scale=10 if dtype == torch.bfloat16 else 1)
        check("running_var", rv, rv0, torch.float32, scale=10 if dtype == torch.bfloat16 else 1)
    go = nhwc(_rand(*y.shape, seed=7), dtype)
    ins = [w] + ([g, be] if use_bn else []) + ([x] if stride == 1 else [])
    rins = [wr] + ([gr, ber] if use_bn else []) + ([xr] if stride == 1 else [])
    grads = torch.autograd.grad(y, ins, go)
    rgrads = torch.autograd.grad(ref, rins, go.float())
    names = ["dw"] + (["dgamma", "dbeta"] if use_bn else []) + (["dx"] if stride == 1 else [])
    for n, a, b_ in zip(names, grads, rgrads):
        check(f"conv {n}", a, b_, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_stem_padded_input(ops, dtype):
    """3-channel image -> NHWC8 -> 3x3 s2 conv with a [16,3,3,3] weight (Cin padded to 8 inside the pack)."""
    x0 = _rand(2, 3, 32, 32, seed=1)
    w = _rand(16, 3, 3, 3, seed=2, scale=0.2).requires_grad_(True)
    x = ops.to_nhwc(x0, dtype)
    y = ops.conv_bn_act(x, w, None, stride=2, pad=1, dil=1, act=0)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    ref = F.conv2d(x0.to(dtype).float(), wr, None, stride=2, padding=1)
    check("stem fwd", y, ref, dtype)
    go = nhwc(_rand(*y.shape, seed=3), dtype)
    (gw,) = torch.autograd.grad(y, [w], go)
    (rw,) = torch.autograd.grad(ref, [wr], go.float())
    assert gw.shape == w.shape
    check("stem dw", gw, rw, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_concat_residual(ops, dtype):
    """fusion conv over cat(res, fm) with two source pointers + residual add in the BN-apply pass."""
    B, C, H, W = 2, 32, 8, 8
    a = nhwc(_rand(B, C, H, W, seed=1), dtype).requires_grad_(True)
    b = nhwc(_rand(B, C, H, W, seed=2), dtype).requires_grad_(True)
    r = nhwc(_rand(B, C, H, W, seed=8), dtype).requires_grad_(True)
    w = _rand(C, 2 * C, 3, 3, seed=3, scale=1 / math.sqrt(18 * C)).requires_grad_(True)
    g = (1 + 0.1 * _rand(C, seed=4)).requires_grad_(True)
    be = (0.1 * _rand(C, seed=5)).requires_grad_(True)
    rm, rv = torch.zeros(C, device=_dev()), torch.ones(C, device=_dev())
    y = ops.conv_bn_act(a, w, None, g, be, rm, rv, stride=1, pad=1, act=0, use_bn=True, training=True, residual=r, x2=b)
    ar, br, rr = (t.detach().float().requires_grad_(True) for t in (a, b, r))
    wr = w.detach().to(dtype).float().requires_grad_(True)
    gr, ber = g.detach().clone().requires_grad_(True), be.detach().clone().requires_grad_(True)
    c = F.conv2d(torch.cat((ar, br), 1), wr, None, padding=1)
    if dtype == torch.bfloat16:
        c = c + (c.to(dtype).float() - c).detach()
    ref = F.batch_norm(c, None, None, gr, ber, training=True) + rr
    check("concat conv fwd", y, ref, dtype)
    go = nhwc(_rand(*y.shape, seed=7), dtype)
    grads = torch.autograd.grad(y, [a, b, r, w, g, be], go)
    rgrads = torch.autograd.grad(ref, [ar, br, rr, wr, gr, ber], go.float())
    for n, x1, x2 in zip(["da", "db", "dres", "dw", "dgamma", "dbeta"], grads, rgrads):
        check(f"concat conv {n}", x1, x2, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_eval_mode(ops, dtype):
    B, C, H, W = 2, 16, 6, 6
    x = nhwc(_rand(B, C, H, W, seed=1), dtype).requires_grad_(True)
    w = _rand(24, C, 1, 1, seed=2, scale=0.25).requires_grad_(True)
    g, be = 1 + 0.1 * _rand(24, seed=3), 0.1 * _rand(24, seed=4)
    rm, rv = 0.1 * _rand(24, seed=5), 1 + 0.1 * _rand(24, seed=6).abs()
    y = ops.conv_bn_act(x, w, None, g, be, rm, rv, act=1, use_bn=True, training=False)
    xr = x.detach().float().requires_grad_(True)
    c = F.conv2d(xr, w.detach().to(dtype).float())
    ref = F.silu(F.batch_norm(c, rm, rv, g, be, training=False))
    check("bn eval fwd", y, ref, dtype)
    go = nhwc(_rand(*y.shape, seed=7), dtype)
    (gx,) = torch.autograd.grad(y, [x], go)
    (rx,) = torch.autograd.grad(ref, [xr], go.float())
    check("bn eval dx", gx, rx, dtype, scale=3)


DW_CASES = [(2, 64, 16, 16, 1, 1), (2, 128, 16, 16, 2, 1), (3, 32, 9, 7, 1, 1), (1, 512, 8, 8, 2, 1), (2, 144, 10, 10, 1, 2), (2, 48, 15, 15, 2, 1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,C,H,W,stride,dil", DW_CASES)
def test_dwconv_bn_act(ops, dtype, B, C, H, W, stride, dil):
    x = nhwc(_rand(B, C, H, W, seed=1), dtype).requires_grad_(True)
    w = _rand(C, 1, 3, 3, seed=2, scale=1 / 3).requires_grad_(True)
    g = (1 + 0.1 * _rand(C, seed=3)).requires_grad_(True)
    be = (0.1 * _rand(C, seed=4)).requires_grad_(True)
    rm, rv = torch.zeros(C, device=_dev()), torch.ones(C, device=_dev())
    y = ops.dwconv_bn_act(x, w, g, be, rm, rv, stride=stride, pad=dil, dil=dil, act=1, use_bn=True, training=True)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    gr, ber = g.detach().clone().requires_grad_(True), be.detach().clone().requires_grad_(True)
    c = F.conv2d(xr, wr, None, stride=stride, padding=dil, dilation=dil, groups=C)
    if dtype == torch.bfloat16:
        c = c + (c.to(dtype).float() - c).detach()
    ref = F.silu(F.batch_norm(c, None, None, gr, ber, training=True))
    check("dw fwd", y, ref, dtype)
    go = nhwc(_rand(*y.shape, seed=7), dtype)
    grads = torch.autograd.grad(y, [x, w, g, be], go)
    rgrads = torch.autograd.grad(ref, [xr, wr, gr, ber], go.float())
    for n, a, b_ in zip(["dx", "dw", "dgamma", "dbeta"], grads, rgrads):
        check(f"dw {n}", a, b_, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(64, 144), (1000, 192), (77, 240), (33, 768), (512, 64), (10, 80), (5, 1024)])
def test_layernorm(ops, dtype, rows, C):
    x = _rand(rows, C, seed=1).to(dtype).requires_grad_(True)
    g = (1 + 0.1 * _rand(C, seed=2)).requires_grad_(True)
    b = (0.1 * _rand(C, seed=3)).requires_grad_(True)
    y = ops.layer_norm(x, g, b, 1e-5)
    xr = x.detach().float().requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    check("ln fwd", y, ref, dtype)
    go = _rand(rows, C, seed=4).to(dtype)
    grads = torch.autograd.grad(y, [x, g, b], go)
    rgrads = torch.autograd.grad(ref, [xr, gr, br], go.float())
    for n, a, b_ in zip(["dx", "dgamma", "dbeta"], grads, rgrads):
        check(f"ln {n}", a, b_, dtype, scale=3)


def _attn_ref(qkv, B, S, h, causal, kpm=None):
    d = qkv.shape[-1] // 3
    c = d // h
    t = qkv.view(B, S, 3, h, c).transpose(1, 3)
    q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
    a = (q * c ** -0.5) @ k.transpose(-1, -2)
    if causal:
        a = a + torch.full((S, S), float("-inf"), device=a.device).triu(1)
    if kpm is not None:
        a = a.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    a = torch.softmax(a, -1)
    return (a @ v).transpose(1, 2).reshape(B * S, d)


ATTN_CASES = [(3, 16, 4, 60), (2, 64, 4, 48), (2, 256, 4, 36), (2, 197, 12, 64), (3, 77, 8, 64), (4, 25, 4, 20), (2, 100, 4, 16), (2, 40, 4, 30),
              (1, 400, 4, 36), (2, 33, 2, 24)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,h,c", ATTN_CASES)
@pytest.mark.parametrize("causal", [False, True])
def test_attention_contiguous(ops, dtype, B, S, h, c, causal):
    d = h * c
    qkv = _rand(B * S, 3 * d, seed=1).to(dtype).requires_grad_(True)
    o = ops.attention(qkv, h, (B, S, 1, 1, S, 1, S), causal=causal)
    qr = qkv.detach().float().requires_grad_(True)
    ref = _attn_ref(qr, B, S, h, causal)
    check("attn fwd", o, ref, dtype)
    go = _rand(B * S, d, seed=2).to(dtype)
    (g,) = torch.autograd.grad(o, [qkv], go)
    (r,) = torch.autograd.grad(ref, [qr], go.float())
    check("attn dqkv", g, r, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_key_padding(ops, dtype):
    B, S, h, c = 3, 77, 8, 64
    d = h * c
    qkv = _rand(B * S, 3 * d, seed=1).to(dtype).requires_grad_(True)
    kpm = torch.zeros(B, S, device=_dev())
    kpm[:, -9:] = 1
    o = ops.attention(qkv, h, (B, S, 1, 1, S, 1, S), causal=True, key_padding_mask=kpm)
    qr = qkv.detach().float().requires_grad_(True)
    ref = _attn_ref(qr, B, S, h, True, kpm)
    check("attn kpm fwd", o, ref, dtype)
    go = _rand(B * S, d, seed=2).to(dtype)
    (g,) = torch.autograd.grad(o, [qkv], go)
    (r,) = torch.autograd.grad(ref, [qr], go.float())
    check("attn kpm dqkv", g, r, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,h,c", [(2, 32, 32, 4, 36), (3, 8, 8, 4, 60), (2, 16, 12, 4, 48), (2, 4, 4, 4, 16)])
def test_attention_unfold_map(ops, dtype, B, H, W, h, c):
    """attention over MobileViT's unfolded patches addressed in place == reference unfold -> MHA core -> fold."""
    import sys
    from oracle.mobilevit_oracle import folding, unfolding
    d = h * c
    fm = _rand(B, 3 * d, H, W, seed=1)                      # the qkv feature map, NCHW
    x = nhwc(fm, dtype)
    t = ops.tokens_of(x).detach().requires_grad_(True)      # [B*H*W, 3d] in NHWC pixel order
    n_h, n_w = H // 2, W // 2
    o = ops.attention(t, h, (B * 4, n_h * n_w, 2, 2, n_w, H, W))
    # reference: unfold q,k,v maps exactly like cvnets/modules/mobilevit_block.py:186-231
    fr = x.detach().float().contiguous().requires_grad_(True)
    patches, info = unfolding(fr, 2, 2)                      # [4B, N, 3d]
    refp = _attn_ref(patches.reshape(-1, 3 * d), B * 4, n_h * n_w, h, False).view(B * 4, n_h * n_w, d)
    ref_map = folding(refp, info, 2, 2)                      # [B, d, H, W]
    ref = ref_map.permute(0, 2, 3, 1).reshape(B * H * W, d)
    check("unfold attn fwd", o, ref, dtype)
    go = _rand(B * H * W, d, seed=2).to(dtype)
    (g,) = torch.autograd.grad(o, [t], go)
    (r,) = torch.autograd.grad(ref, [fr], go.float())
    check("unfold attn dqkv", g, r.permute(0, 2, 3, 1).reshape(B * H * W, 3 * d), dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_and_dropout(ops, dtype):
    x = nhwc(_rand(4, 64, 8, 8, seed=1), dtype).requires_grad_(True)
    y = ops.GlobalAvgPool.apply(x)
    xr = x.detach().float().requires_grad_(True)
    ref = xr.mean(dim=[-2, -1])
    check("pool fwd", y, ref, dtype)
    go = _rand(4, 64, seed=2).to(dtype)
    (g,) = torch.autograd.grad(y, [x], go)
    (r,) = torch.autograd.grad(ref, [xr], go.float())
    check("pool bwd", g, r, dtype)
    # dropout: mask statistics + exact mask reuse in backward + epilogue/standalone agreement
    t = torch.ones(512, 256, device=_dev(), dtype=dtype, requires_grad=True)
    p = 0.1
    d1 = ops.dropout(t, p, True)
    keep = (d1 != 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01, keep
    vals = d1[d1 != 0].float()
    assert torch.allclose(vals, torch.full_like(vals, 1 / (1 - p)), rtol=1e-2)
    (gd,) = torch.autograd.grad(d1, [t], torch.ones_like(d1))
    assert torch.equal(gd != 0, d1 != 0)
    d2 = ops.dropout(t, p, True)
    assert not torch.equal(d1 != 0, d2 != 0)  # different call site id -> different mask


def test_big_gemm_matches_generic_kernel(ops):
    """the large-tile kernel and the generic implicit-GEMM kernel must agree on the same bf16 problem (identical fp32 accumulate order
    per 64-wide K step is NOT guaranteed, so: tight tolerance, not bit equality), including the dropout mask (same element indexing)."""
    from cvnets_amd import _lib

    M, K, N = 44100, 768, 1536  # >= 1024 tiles of 256 x 256 (knob 1 -> gemm_nt256_kernel), M >= 8192 (knob 2 -> 256 x 128); M % 256 != 0: row clamp
    x = _rand(M, K, seed=11).to(torch.bfloat16)
    w = _rand(N, K, seed=12, scale=1 / math.sqrt(K))
    b = _rand(N, seed=13, scale=0.1)
    res = _rand(M, N, seed=14).to(torch.bfloat16)
    outs = []
    for knob in (1, 0, 2, 3):  # 1: 256 x 256 four-stage kernel (default for these shapes), 0: generic, 2: 256 x 128, 3: 128 x 128
        _lib.call("cvh_set_tuning", 5, knob)
        try:
            y = ops.LinearAct.apply(x, w, b, res, None, (2, 0.1, 77, False, 0))
            outs.append(y.float())
        finally:
            _lib.call("cvh_set_tuning", 5, 1)
    for o in outs[1:]:
        assert torch.equal(outs[0] == res.float(), o == res.float())  # identical dropout masks
        assert l2_err(outs[0], o) < 2e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_dropout_epilogue(ops, dtype):
    M, K, N = 256, 64, 128
    x = _rand(M, K, seed=1).to(dtype).requires_grad_(True)
    w = _rand(N, K, seed=2, scale=0.125).requires_grad_(True)
    res = torch.zeros(M, N, device=_dev(), dtype=dtype)
    y0 = ops.linear(x, w, None)
    y = ops.linear(x, w, None, drop_p=0.25, residual=res)
    delta = (y.float() - res.float())
    mask = delta.abs() > 1e-6
    frac = mask.float().mean().item()
    assert abs(frac - 0.75) < 0.02, frac
    check("dropout epilogue kept values", delta[mask], (y0.float() / 0.75)[mask], dtype, scale=2)
    (gx,) = torch.autograd.grad(y, [x], torch.ones_like(y))
    gref = ((mask.float() / 0.75).to(dtype).float() @ w.detach().to(dtype).float())
    check("dropout epilogue dx", gx, gref, dtype, scale=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W,Ho,Wo", [(1, 1, 2, 2), (2, 2, 1, 1), (5, 5, 6, 6), (6, 6, 5, 5), (2, 3, 2, 4), (7, 9, 8, 10), (10, 10, 5, 7), (4, 4, 9, 13)])
def test_resize_bilinear(ops, dtype, H, W, Ho, Wo):
    x = nhwc(_rand(2, 16, H, W, seed=1), dtype).requires_grad_(True)
    y = ops.resize_bilinear(x, Ho, Wo)
    xr = x.detach().float().requires_grad_(True)
    ref = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False)
    check("resize fwd", y, ref, dtype)
    go = nhwc(_rand(2, 16, Ho, Wo, seed=2), dtype)
    (g,) = torch.autograd.grad(y, [x], go)
    (r,) = torch.autograd.grad(ref, [xr], go.float())
    check("resize bwd", g, r, dtype, scale=2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(3, 64, 1, 1), (2, 144, 2, 2), (5, 24, 1, 1), (1, 64, 2, 2)])
def test_layer_norm_reference_quirk(ops, dtype, cfg):
    """S == C token tensors: the reference's LayerNorm normalises over the TOKEN axis with (std + eps) and token-indexed affine
    (cvnets/layers/normalization/layer_norm.py:53-66); cvh_ln_seq_* must reproduce exactly that, on contiguous sequences and on
    sequences gathered from an NHWC map (MobileViT unfolding)."""
    from cvnets_amd.layers import LayerNorm

    B, C, ph, pw = cfg
    S = C
    torch.manual_seed(0)
    ln = LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C) * 0.5 + 1.0)
        ln.bias.copy_(torch.randn(C) * 0.1)
    if ph == 1:
        H, W, n_w = 1, S, S
    else:  # S patches of ph x pw pixels: feature map (n_h*ph) x (n_w*pw) with n_h*n_w == S
        n_w = int(round(S ** 0.5)) if int(round(S ** 0.5)) ** 2 == S else S // 8
        H, W = (S // n_w) * ph, n_w * pw
    rows = B * H * W
    x = _rand(rows, C, seed=3).to(dtype).requires_grad_(True)
    seqmap = (B * ph * pw, S, ph, pw, n_w, H, W)
    y = ops.layer_norm_tokens(x, ln, seqmap)
    # reference formula on the explicitly unfolded [B', S, C] tensor
    xr = x.detach().float().requires_grad_(True)
    fm = xr.view(B, H // ph, ph, W // pw, pw, C).permute(0, 2, 4, 1, 3, 5).reshape(B * ph * pw, S, C)  # [b*P + i*pw + j, nh*n_w + nw, c]
    wr, br = ln.weight.detach().clone().requires_grad_(True), ln.bias.detach().clone().requires_grad_(True)
    sd, mu = torch.std_mean(fm, dim=1, keepdim=True, unbiased=False)
    ref = torch.addcmul(br.reshape(1, C, 1), (fm - mu) / (sd + ln.eps), wr.reshape(1, C, 1))
    ref_rows = ref.view(B, ph, pw, H // ph, W // pw, C).permute(0, 3, 1, 4, 2, 5).reshape(rows, C)
    check("ln quirk fwd", y, ref_rows, dtype)
    go = _rand(rows, C, seed=4).to(dtype)
    gx, gw, gb = torch.autograd.grad(y, [x, ln.weight, ln.bias], go)
    rx, rw, rb = torch.autograd.grad(ref_rows, [xr, wr, br], go.float())
    check("ln quirk dx", gx, rx, dtype, scale=3)
    check("ln quirk dgamma", gw, rw, dtype, scale=3)
    check("ln quirk dbeta", gb, rb, dtype, scale=3)
    ln.reference_quirk = False  # opt-out: the documented channel-last LayerNorm
    y2 = ops.layer_norm_tokens(x, ln, seqmap)
    check("ln no-quirk", y2, F.layer_norm(x.detach().float(), (C,), ln.weight.detach(), ln.bias.detach(), ln.eps), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,H", [(300, 64, 128), (2500, 256, 1024)])
@pytest.mark.parametrize("act", [1, 2])
def test_ffn_pair_fused_activation_backward(ops, dtype, M, K, H, act):
    """fc1 -> act -> fc2 with the activation backward folded into fc2's dX GEMM epilogue (LinearAct expose_pre / in_pre) must give the
    same outputs and gradients as the unfused pair (second shape: the large-tile kernel in bf16)."""
    x = _rand(M, K, seed=21).to(dtype).requires_grad_(True)
    w1 = _rand(H, K, seed=22, scale=1 / math.sqrt(K)).requires_grad_(True)
    b1 = _rand(H, seed=23, scale=0.1).requires_grad_(True)
    w2 = _rand(K, H, seed=24, scale=1 / math.sqrt(H)).requires_grad_(True)
    b2 = _rand(K, seed=25, scale=0.1).requires_grad_(True)
    go = _rand(M, K, seed=26).to(dtype)
    h, pre = ops.linear(x, w1, b1, act=act, expose_pre=True)
    y = ops.linear(h, w2, b2, residual=x, in_pre=pre, in_act=act)
    g_f = torch.autograd.grad(y, [x, w1, b1, w2, b2], go)
    h0 = ops.linear(x, w1, b1, act=act)
    y0 = ops.linear(h0, w2, b2, residual=x)
    g_u = torch.autograd.grad(y0, [x, w1, b1, w2, b2], go)
    assert torch.equal(y, y0)
    for name, a, b in zip(("dx", "dw1", "db1", "dw2", "db2"), g_f, g_u):
        check(f"ffn fused {name}", a, b, dtype, scale=1 if dtype == torch.float32 else 2)
