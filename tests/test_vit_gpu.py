"""ViT (patch-embedding conv stem + class token + learnable positional embedding + 12 x pre-norm
MHSA/MLP + LayerNorm + classifier) through the HIP path: parity against the reference's golden fixtures
(tests/golden/vit_tiny_*.npz, oracle/make_golden.py) and the live CPU oracle; plus the token-plumbing kernels
(csrc/tokens.hip) and the non-overlapping strided-conv dX scatter (cvh_conv_dx_patch) one by one.

Tolerances: fp32 mode logits rel-L2 <= 1e-4, per-tensor gradients <= 2e-3; bf16 mode within BF16_SLACK x the reference's own
bf16-autocast deviation recorded in the fixture (see tests/test_model_gpu.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu
BF16_SLACK = 1.5
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("vit_tiny_64_b2", "tiny", 2, 64), ("vit_tiny_224_b2", "tiny", 2, 224)]


def _build(mode, dtype):
    import cvnets_amd
    from oracle.weights import seeded_state_dict

    model = cvnets_amd.build_vit(mode, **{"model.classification.vit.dropout": 0.0})
    model.emb_dropout.p = 0.0
    shapes = json.load(open(os.path.join(GOLD, f"vit_{mode}_keys.json")))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes
    sd = seeded_state_dict(shapes, seed=0)
    sd["cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": (1, 1, shapes["cls_token"][2])}, seed=0)["cls_token_values"]
    model.load_state_dict(sd, strict=True)
    cvnets_amd.set_compute_dtype(dtype)
    return model.to("cuda:0"), sd


def _step(model, x, y):
    model.train()
    model.zero_grad(set_to_none=True)
    logits = model(x)
    loss = F.cross_entropy(logits.float(), y, label_smoothing=0.1)
    loss.backward()
    return logits.detach().float().cpu(), float(loss.detach()), {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}


@pytest.mark.parametrize("name,mode,batch,res", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vit_train_step_vs_reference_golden(name, mode, batch, res, dtype):
    from oracle.weights import seeded_input, seeded_labels

    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model, sd = _build(mode, dtype)
    x = seeded_input((batch, 3, res, res), seed=1).cuda()
    y = seeded_labels(batch, 1000, seed=1).cuda()
    model.eval()
    with torch.no_grad():
        le = model(x).float().cpu()
    logits, loss, grads = _step(model, x, y)
    fp32 = dtype == torch.float32
    ref_bf16 = json.loads(str(gold["ref_bf16_autocast_err"]))
    e_eval = l2_err(le, torch.from_numpy(gold["logits_eval"]))
    e_train = l2_err(logits, torch.from_numpy(gold["logits_train"]))
    print(f"[{name} {dtype}] logits rel-L2 eval {e_eval:.2e} train {e_train:.2e} loss {loss:.5f} vs {float(gold['loss']):.5f}")
    assert e_eval < (1e-4 if fp32 else 3e-2), e_eval
    assert e_train < (1e-4 if fp32 else max(2e-2, BF16_SLACK * ref_bf16["logits_train"])), (e_train, ref_bf16)
    assert abs(loss - float(gold["loss"])) < (1e-4 if fp32 else max(2e-2, BF16_SLACK * ref_bf16["loss"]))
    names = [str(n) for n in gold["grad_names"]]
    assert names == [k for k, _ in model.named_parameters()]
    gn = torch.tensor([grads[k].norm().item() for k in names], dtype=torch.float64)
    gref = torch.from_numpy(gold["grad_norm"])
    worst = float(((gn - gref).abs() / (gref + 1e-3 * gref.max())).max())
    print(f"[{name} {dtype}] worst per-tensor grad-norm deviation {worst:.2e}")
    assert worst < (2e-3 if fp32 else max(1e-2, BF16_SLACK * ref_bf16["grad_norm_worst"])), (worst, ref_bf16)
    for key in gold.files:
        if key.startswith("grad::"):
            e = l2_err(grads[key[6:]], torch.from_numpy(gold[key]))
            print(f"   {key} rel-L2 {e:.2e}")
            assert e < (2e-3 if fp32 else max(3e-2, BF16_SLACK * ref_bf16["grad_full_worst"])), (key, e, ref_bf16)
        if key.startswith("bn::"):
            e = l2_err(model.state_dict()[key[4:]].float().cpu(), torch.from_numpy(gold[key]))
            assert e < (1e-4 if fp32 else 3e-2), (key, e)


def test_vit_vs_live_oracle_all_gradients():
    """fresh inputs, rectangular image (pos-embedding table resized 196 -> 3x5 = 15), batch 3: every gradient tensor vs the oracle."""
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels

    model, sd = _build("tiny", torch.float32)
    x = seeded_input((3, 3, 48, 80), seed=11)
    y = seeded_labels(3, 1000, seed=11)
    logits, loss, grads = _step(model, x.cuda(), y.cuda())
    o_logits, o_loss, o_grads, o_running = orc.generic_train_step(orc.vit_forward, sd, x, y, mode="tiny")
    assert l2_err(logits, o_logits) < 1e-4
    assert abs(loss - float(o_loss)) < 1e-4
    gmax = max(float(v.norm()) for v in o_grads.values())
    for k, g in o_grads.items():
        assert l2_err(grads[k], g) < 2e-3 or g.norm() < 1e-5 * gmax, (k, l2_err(grads[k], g))
    for k, v in o_running.items():
        assert l2_err(model.state_dict()[k].float().cpu(), v) < 1e-4, k


# ------------------------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("has_cls", [True, False])
def test_vit_embed_fwd_bwd(dtype, has_cls):
    from cvnets_amd import ops

    B, N, E = 3, 49, 192
    g = torch.Generator().manual_seed(0)
    patch = torch.randn(B * N, E, generator=g).to(dtype)
    pos = torch.randn(N, E, generator=g)
    cls = torch.randn(E, generator=g) if has_cls else None
    S = N + int(has_cls)
    dout = torch.randn(B * S, E, generator=g).to(dtype)
    pr, posr = patch.float().clone().requires_grad_(), pos.clone().requires_grad_()
    clsr = cls.clone().requires_grad_() if has_cls else None
    t = pr.view(B, N, E) + posr
    if has_cls:
        t = torch.cat((clsr.view(1, 1, E).expand(B, -1, -1), t), dim=1)
    t.reshape(B * S, E).backward(dout.float())
    pg, posg = patch.detach().cuda().requires_grad_(), pos.detach().cuda().requires_grad_()
    clsg = cls.detach().cuda().requires_grad_() if has_cls else None
    out = ops.VitEmbed.apply(pg, posg, clsg, B)
    out.backward(dout.cuda())
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert l2_err(out.float().cpu(), t.reshape(B * S, E).detach()) < tol
    assert l2_err(pg.grad.float().cpu(), pr.grad) < tol
    assert l2_err(posg.grad.cpu(), posr.grad) < tol
    if has_cls:
        assert l2_err(clsg.grad.cpu(), clsr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rows_gather(dtype):
    from cvnets_amd import ops

    B, S, E = 5, 17, 64
    x = torch.randn(B * S, E).to(dtype)
    for idx in (0, 16):
        xg = x.cuda().requires_grad_()
        y = ops.RowsGather.apply(xg, B, S, idx)
        assert torch.equal(y.cpu(), x.view(B, S, E)[:, idx])
        dy = torch.randn(B, E).to(dtype)
        y.backward(dy.cuda())
        ref = torch.zeros(B, S, E, dtype=dtype)
        ref[:, idx] = dy
        assert torch.equal(xg.grad.cpu(), ref.view(B * S, E))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_lookup(dtype):
    """text-tower embedding (cvh_embed_lookup_fwd/bwd): gather + positional embedding; scatter-add with repeated tokens and padding."""
    from cvnets_amd import _lib
    from cvnets_amd.ops import _dt, _p, _stream

    B, S, E, V = 4, 9, 64, 50
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, V, (B, S), generator=g)
    tok[:, -2:] = 0  # padding index, repeated
    table = torch.randn(V, E, generator=g)
    pos = torch.randn(S, E, generator=g)
    out = torch.empty(B * S, E, dtype=dtype, device="cuda")
    tk, tb, ps = tok.cuda(), table.cuda(), pos.cuda()
    _lib.call("cvh_embed_lookup_fwd", _dt(out), _p(tk), _p(tb), _p(ps), _p(out), B * S, S, E, _stream())
    ref = table[tok] + pos
    assert l2_err(out.float().cpu(), ref.view(B * S, E)) < (1e-6 if dtype == torch.float32 else 8e-3)
    dout = torch.randn(B * S, E, generator=g).to(dtype)
    dtab = torch.zeros(V, E, device="cuda")
    dg = dout.cuda()  # a named reference: `_p(dout.cuda())` hands the kernel the address of a tensor that is freed before the launch
    _lib.call("cvh_embed_lookup_bwd", _dt(out), _p(tk), _p(dg), _p(dtab), B * S, E, 0, _stream())
    dref = torch.zeros(V, E).index_add_(0, tok.view(-1), dout.float())
    dref[0] = 0  # padding_idx rows receive no gradient (nn.Embedding(padding_idx=0))
    assert l2_err(dtab.cpu(), dref) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 8, 32, 16, 16, 4), (3, 48, 48, 12, 20, 2), (2, 48, 192, 6, 10, 2), (1, 64, 40, 8, 8, 2)])
def test_patch_conv_dx_and_dw(dtype, cfg):
    """k == stride, pad 0 convolutions (ViT stem): forward, dX through the scatter GEMM (cvh_conv_dx_patch), dW, dbias."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.layers import ConvLayer2d, default_opts

    B, Cin, Cout, H, W, k = cfg
    cvnets_amd.set_compute_dtype(dtype)
    opts = default_opts()
    layer = ConvLayer2d(opts, Cin, Cout, k, stride=k, padding=(0, 0), bias=True, use_norm=False, use_act=False).cuda()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H // k, W // k, generator=g)
    w = layer.block.conv.weight.detach().cpu()
    b = layer.block.conv.bias.detach().cpu()
    q = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    xr, wr, br = q(x).requires_grad_(), q(w).requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, stride=k)
    yr.backward(q(dy))
    xg = ops.to_nhwc(x.cuda()).detach().requires_grad_()
    y = layer(xg)
    y.backward(dy.cuda().to(y.dtype).contiguous(memory_format=torch.channels_last))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert l2_err(y.float().cpu(), yr.detach()) < tol
    assert l2_err(xg.grad.float().cpu(), xr.grad) < tol
    assert l2_err(layer.block.conv.weight.grad.float().cpu(), wr.grad) < tol
    assert l2_err(layer.block.conv.bias.grad.float().cpu(), br.grad) < tol


def test_masked_attention_like_reference_test():
    """mirror of the reference's tests/modules/test_transformer.py::test_masked_attention (float -inf key padding mask, 8 heads of
    dimension 1, 66 tokens): unmasked queries of a masked sequence agree with each other, masked positions differ, the unmasked
    sequence is uniform."""
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    cvnets_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    B, N, C = 2, 64 + 2, 8
    t = cvnets_amd.TransformerEncoder(default_opts(), embed_dim=C, ffn_latent_dim=4 * C).cuda().eval()
    x = torch.ones([B, N, C])
    x[:, :] = torch.randn([C])
    key_padding_mask = torch.zeros([B, N])
    key_padding_mask[0, 63:] = float("-inf")
    x[0, 63:] = 0
    y = t(x.cuda(), key_padding_mask=key_padding_mask.cuda()).float().cpu()
    assert torch.all((y[0, 0] - y[0, :63]).abs() < 1e-5)
    assert torch.all(y[0, 0] != y[0, 63:])
    assert torch.all((y[1, 0] - y[1, :]).abs() < 1e-5)


@pytest.mark.parametrize("input_seq_len", [34, 128, 192])
@pytest.mark.parametrize("sequence_first", [True, False])
@pytest.mark.parametrize("padding_idx", [None, 0])
def test_pos_embedding_like_reference_test(input_seq_len, sequence_first, padding_idx):
    """mirror of the reference's tests/test_pos_embeddings.py (learnable variant): output shape contract, plus values vs F.interpolate"""
    from cvnets_amd.layers import PositionalEmbedding

    pe = PositionalEmbedding(opts=None, num_embeddings=128, embedding_dim=512, padding_idx=padding_idx, is_learnable=True,
                             sequence_first=sequence_first).cuda()
    out = pe(input_seq_len)
    assert out.shape[0 if sequence_first else 1] == input_seq_len and out.shape[1 if sequence_first else 0] == 1
    w = pe.pos_embed.pos_embed.detach().cpu()
    ref = F.interpolate(w, size=(input_seq_len, 512), mode="bilinear") if input_seq_len != 128 else w
    assert l2_err(out.float().cpu().reshape(input_seq_len, 512), ref.reshape(input_seq_len, 512)) < 1e-6
    if padding_idx is not None:
        assert float(pe.pos_embed.pos_embed[0, 0, padding_idx].abs().max()) == 0.0


@pytest.mark.parametrize("output_dim", [32, 48])
@pytest.mark.parametrize("batch_size", [1, 2])
@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("use_attn_mask", [True, False])
def test_multihead_self_attn_like_reference_test(output_dim, batch_size, bias, use_attn_mask):
    """mirror of the reference's tests/test_multi_head_attn.py::test_multihead_self_attn (seq 5, embed 8, 2 heads, output_dim != embed_dim,
    with / without bias, with / without the causal additive mask): the HIP layer vs the oracle's forward_default restatement, which
    oracle/make_golden.py pins against the reference (whose own test pins forward_default against torch's MHA)."""
    import cvnets_amd
    from oracle import mobilevit_oracle as orc

    cvnets_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    seq_len, embed_dim = 5, 8
    mha = cvnets_amd.MultiHeadAttention(embed_dim=embed_dim, num_heads=2, attn_dropout=0.0, bias=bias, output_dim=output_dim).cuda().eval()
    x = torch.randn(batch_size, seq_len, embed_dim)
    mask = None
    if use_attn_mask:
        mask = torch.full((seq_len, seq_len), float("-inf")).triu_(1).unsqueeze(0).expand(batch_size, -1, -1)
    got = mha(x_q=x.cuda(), attn_mask=mask.cuda() if mask is not None else None).float().cpu()
    sd = {"mha." + k: v.detach().cpu() for k, v in mha.state_dict().items()}
    ref = orc.multi_head_attention(sd, "mha", x, 2, attn_mask=mask)
    torch.testing.assert_close(actual=got, expected=ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("mode", ["small", "base"])
def test_vit_other_sizes_vs_live_oracle(mode):
    """ViT small (384 / 6 heads) and base (768 / 12 heads; bf16 linears of this size take the large-tile kernels) at 64x64, batch 2."""
    import cvnets_amd
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict

    model = cvnets_amd.build_vit(mode, **{"model.classification.vit.dropout": 0.0})
    model.emb_dropout.p = 0.0
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    sd["cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": shapes["cls_token"]}, seed=0)["cls_token_values"]
    model.load_state_dict(sd)
    cvnets_amd.set_compute_dtype(torch.float32)
    model = model.cuda()
    x, y = seeded_input((2, 3, 64, 64), seed=6), seeded_labels(2, 1000, seed=6)
    logits, loss, grads = _step(model, x.cuda(), y.cuda())
    o_logits, o_loss, o_grads, _ = orc.generic_train_step(orc.vit_forward, sd, x, y, mode=mode)
    assert l2_err(logits, o_logits) < 1e-4 and abs(loss - float(o_loss)) < 1e-4
    gmax = max(float(v.norm()) for v in o_grads.values())
    for k, g in o_grads.items():
        assert l2_err(grads[k], g) < 2e-3 or g.norm() < 1e-5 * gmax, (k, l2_err(grads[k], g))
