"""General ADDITIVE attention masks in the fused attention kernels (cvnets/layers/multi_head_attention.py:197-208: `attn = attn + attn_mask`
for any [N, S, T] mask, then the key-padding mask as -inf; the reference's own tests/modules/test_transformer.py and
tests/test_multi_head_attn.py pin the semantics).  The forward and both backward kernels add the mask to the scaled scores; the causal
triangle stays the generated fast path.  Checked against the oracle (oracle/mobilevit_oracle.py, which is pinned on the reference): outputs,
the input gradient and every parameter gradient, fp32 (2e-4 / 2e-3) and bf16 (2e-2 / 5e-2); sequence lengths that are not tile multiples,
masks with -inf entries, a shared [S, S] mask through the sequence-first variant, masks together with a key-padding mask, with
cross-attention (mask [N, S, T], S != T) and through a whole TransformerEncoder."""
import pytest
import torch
import torch.nn.functional as F

from oracle import mobilevit_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mask(b, s, t, seed, inf_frac=0.15):
    g = torch.Generator().manual_seed(seed)
    m = torch.randn(b, s, t, generator=g) * 2.0
    hide = torch.rand(b, s, t, generator=g) < inf_frac
    hide[..., 0] = False  # every query keeps one visible key: a fully masked row is NaN in the reference as well
    return m.masked_fill(hide, float("-inf"))


def _grads(y, go, params):
    return torch.autograd.grad(y, params, go)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,C,H,kp", [(3, 50, 64, 4, False), (2, 197, 96, 3, True), (4, 16, 32, 2, False), (2, 77, 128, 8, True)])
def test_additive_mask_self_attention(B, S, C, H, kp, dtype):
    import cvnets_amd
    torch.manual_seed(B + S)
    m = cvnets_amd.MultiHeadAttention(C, H).to(DEV).train()
    x = torch.randn(B, S, C, device=DEV, requires_grad=True)
    am = _mask(B, S, S, 7 + S)
    kpm = None
    if kp:
        kpm = torch.zeros(B, S)
        kpm[0, S - 9:] = float("-inf")  # the float form the reference's own test uses
    params = [x] + list(m.parameters())
    cvnets_amd.set_compute_dtype(dtype)
    try:
        y = m(x, attn_mask=am.to(DEV), key_padding_mask=None if kpm is None else kpm.to(DEV))
        go = torch.randn(y.shape, device=DEV)
        gs = _grads(y.float(), go, params)
    finally:
        cvnets_amd.set_compute_dtype(None)
    sd = {"mha." + k: v.detach().cpu().float() for k, v in m.state_dict().items()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = orc.multi_head_attention(pr, "mha", xr, H, attn_mask=am, key_padding_mask=kpm)
    rg = torch.autograd.grad(ref, [xr] + [pr["mha." + k] for k, _ in m.named_parameters()], go.cpu())
    to, tg = (2e-4, 2e-3) if dtype == torch.float32 else (2e-2, 5e-2)
    assert torch.isfinite(y.float()).all()
    assert float((y.detach().float().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < to
    for g, r in zip(gs, rg):
        assert float((g.float().cpu() - r).abs().max() / (r.abs().max() + 1e-12)) < tg


def test_shared_2d_mask_sequence_first():
    """forward_pytorch (multi_head_attention.py:241-273): a non-causal [S, S] float mask and a boolean one through F.multi_head_attention_forward"""
    import cvnets_amd
    torch.manual_seed(11)
    S, B, C, H = 23, 3, 64, 4
    m = cvnets_amd.MultiHeadAttention(C, H).to(DEV).train()
    w = m.qkv_proj.weight.detach()
    for am in (_mask(1, S, S, 5)[0].to(DEV), (torch.rand(S, S, device=DEV) < 0.3).triu(1)):
        x = torch.randn(S, B, C, device=DEV, requires_grad=True)
        cvnets_amd.set_compute_dtype(torch.float32)
        try:
            y = m(x, attn_mask=am, use_pytorch_mha=True)
            go = torch.randn_like(y)
            (gx,) = torch.autograd.grad(y, [x], go)
        finally:
            cvnets_amd.set_compute_dtype(None)
        xr = x.detach().clone().requires_grad_(True)
        ref, _ = F.multi_head_attention_forward(
            query=xr, key=xr, value=xr, embed_dim_to_check=C, num_heads=H, in_proj_weight=torch.empty([0]), in_proj_bias=m.qkv_proj.bias.detach(),
            bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0.0, out_proj_weight=m.out_proj.weight.detach(),
            out_proj_bias=m.out_proj.bias.detach(), training=True, key_padding_mask=None, need_weights=False, attn_mask=am,
            use_separate_proj_weight=True, q_proj_weight=w[:C], k_proj_weight=w[C:2 * C], v_proj_weight=w[2 * C:])
        (rx,) = torch.autograd.grad(ref, [xr], go)
        assert float((y.float() - ref).abs().max() / ref.abs().max()) < 2e-4
        assert float((gx.float() - rx).abs().max() / rx.abs().max()) < 2e-3


@pytest.mark.parametrize("S,T", [(20, 36), (50, 17), (33, 33)])
def test_additive_mask_cross_attention(S, T):
    import cvnets_amd
    torch.manual_seed(S * T)
    B, C, H = 3, 64, 4
    m = cvnets_amd.MultiHeadAttention(C, H).to(DEV).train()
    xq = torch.randn(B, S, C, device=DEV, requires_grad=True)
    xk = torch.randn(B, T, C, device=DEV, requires_grad=True)
    am = _mask(B, S, T, 3 + S)
    kpm = torch.zeros(B, T, dtype=torch.bool)
    kpm[1, T - 5:] = True
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        y = m(xq, xk, key_padding_mask=kpm.to(DEV), attn_mask=am.to(DEV))
        go = torch.randn_like(y)
        gq, gk = torch.autograd.grad(y, [xq, xk], go)
    finally:
        cvnets_amd.set_compute_dtype(None)
    sd = {"mha." + k: v.detach().cpu().float() for k, v in m.state_dict().items()}
    xqr, xkr = xq.detach().cpu().clone().requires_grad_(True), xk.detach().cpu().clone().requires_grad_(True)
    ref = orc.multi_head_attention(sd, "mha", xqr, H, attn_mask=am, key_padding_mask=kpm, x_kv=xkr)
    rq, rk = torch.autograd.grad(ref, [xqr, xkr], go.cpu())
    assert float((y.float().cpu() - ref).abs().max() / ref.abs().max()) < 2e-4
    assert float((gq.float().cpu() - rq).abs().max() / rq.abs().max()) < 2e-3
    assert float((gk.float().cpu() - rk).abs().max() / rk.abs().max()) < 2e-3


def test_transformer_encoder_with_additive_mask():
    """TransformerEncoder.forward(x, attn_mask=..., key_padding_mask=...) (cvnets/modules/transformer.py:129-156) against the oracle's encoder"""
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    torch.manual_seed(21)
    B, S, C, FFN, H = 2, 45, 64, 128, 4
    enc = cvnets_amd.modules.TransformerEncoder(default_opts(), embed_dim=C, ffn_latent_dim=FFN, num_heads=H, attn_dropout=0.0, dropout=0.0,
                                               ffn_dropout=0.0).to(DEV).train()
    x = torch.randn(B, S, C, device=DEV, requires_grad=True)
    am = _mask(B, S, S, 9)
    kpm = torch.zeros(B, S)
    kpm[1, 40:] = float("-inf")
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        y = enc(x, attn_mask=am.to(DEV), key_padding_mask=kpm.to(DEV))
        go = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, [x], go)
    finally:
        cvnets_amd.set_compute_dtype(None)
    sd = {"e." + k: v.detach().cpu().float() for k, v in enc.state_dict().items()}
    xr = x.detach().cpu().clone().requires_grad_(True)
    ref = orc.transformer_encoder(sd, "e", xr, H, act="swish", attn_mask=am, key_padding_mask=kpm)
    (rx,) = torch.autograd.grad(ref, [xr], go.cpu())
    assert float((y.float().cpu() - ref).abs().max() / ref.abs().max()) < 2e-4
    assert float((gx.float().cpu() - rx).abs().max() / rx.abs().max()) < 2e-3


def test_causal_mask_is_still_generated_in_kernel():
    """the causal triangle (text tower, cvnets/text_encoders/transformer.py:343-352) must not fall onto the bias path"""
    from cvnets_amd.layers import _split_mask
    s = 12
    tri = torch.full((s, s), float("-inf")).triu(1)
    assert _split_mask(tri[None].expand(3, s, s), 3, s, s) == (True, None)
    other = tri.clone()
    other[0, 5] = 0.0
    causal, bias = _split_mask(torch.stack([tri, other]), 2, s, s)
    assert causal is False and bias.shape == (2, s, s)  # one differing sample: the whole batch takes the general path
