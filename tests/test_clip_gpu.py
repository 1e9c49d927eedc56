"""CLIP (ViT image tower + projection head, causal text transformer + EOT projection, contrastive loss)
through the HIP path against the reference's golden fixture (tests/golden/clip_tiny_64_b8.npz, oracle/make_golden.py) and the live
CPU oracle; plus the head / loss kernels one by one against PyTorch fp32.

Tolerances: fp32 mode embeddings rel-L2 <= 1e-4, loss 1e-4, per-tensor gradients <= 2e-3; bf16 mode within BF16_SLACK x the
reference's own bf16-autocast deviation recorded in the fixture."""
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


def _build(dtype):
    import cvnets_amd
    from oracle.weights import seeded_state_dict

    gold = np.load(os.path.join(GOLD, "clip_tiny_64_b8.npz"))
    c = json.loads(str(gold["config"]))
    model = cvnets_amd.build_clip(**{"model.classification.vit.mode": c["vit_mode"], "model.text.transformer.model_dim": c["text_dim"],
                                     "model.text.transformer.n_transformer_layers": c["text_layers"],
                                     "model.text.transformer.n_heads_per_layer": c["text_heads"], "dataset.text_vocab_size": c["vocab"],
                                     "dataset.text_context_length": c["ctx"], "model.multi_modal_image_text.clip.projection_dim": c["proj"]})
    model.image_encoder.emb_dropout.p = 0.0
    shapes = json.load(open(os.path.join(GOLD, "clip_tiny_keys.json")))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes
    sd = seeded_state_dict({k: tuple(v) for k, v in shapes.items()}, seed=0)
    sd["logit_scale"] = torch.tensor(float(np.log(1.0 / 0.07)))
    sd["image_encoder.cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": tuple(shapes["image_encoder.cls_token"])}, seed=0)["cls_token_values"]
    model.load_state_dict(sd, strict=True)
    cvnets_amd.set_compute_dtype(dtype)
    return model.to("cuda:0"), sd, gold, c


def _step(model, x, tok):
    import cvnets_amd
    from cvnets_amd.layers import default_opts

    loss_fn = cvnets_amd.ContrastiveLossClip(default_opts()).train()
    model.train()
    model.zero_grad(set_to_none=True)
    out = model({"image": x, "text": tok})
    img, txt = out["image"].detach().float().cpu(), out["text"].detach().float().cpu()
    losses = loss_fn(None, out)
    losses["total_loss"].backward()
    return img, txt, float(losses["total_loss"].detach()), {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_clip_train_step_vs_reference_golden(dtype):
    from oracle.weights import seeded_input

    model, sd, gold, c = _build(dtype)
    x = seeded_input((c["batch"], 3, c["res"], c["res"]), seed=1).cuda()
    tok = torch.from_numpy(gold["tokens"]).cuda()
    img, txt, loss, grads = _step(model, x, tok)
    fp32 = dtype == torch.float32
    ref = json.loads(str(gold["ref_bf16_autocast_err"]))
    e_img, e_txt = l2_err(img, torch.from_numpy(gold["image"])), l2_err(txt, torch.from_numpy(gold["text"]))
    print(f"[clip {dtype}] image {e_img:.2e} text {e_txt:.2e} loss {loss:.5f} vs {float(gold['loss']):.5f}")
    assert e_img < (1e-4 if fp32 else max(1e-2, BF16_SLACK * ref["image"])), e_img
    assert e_txt < (1e-4 if fp32 else max(1e-2, BF16_SLACK * ref["text"])), e_txt
    assert abs(loss - float(gold["loss"])) < (1e-4 if fp32 else max(2e-2, BF16_SLACK * ref["loss"]))
    names = [str(n) for n in gold["grad_names"]]
    assert names == [k for k, _ in model.named_parameters()]
    gn = torch.tensor([grads[k].norm().item() for k in names], dtype=torch.float64)
    gref = torch.from_numpy(gold["grad_norm"])
    worst = float(((gn - gref).abs() / (gref + 1e-3 * gref.max())).max())
    print(f"[clip {dtype}] worst per-tensor grad-norm deviation {worst:.2e}")
    assert worst < (2e-3 if fp32 else max(2e-2, BF16_SLACK * ref["grad_norm_worst"])), (worst, ref)
    for key in gold.files:
        if key.startswith("grad::"):
            e = l2_err(grads[key[6:]], torch.from_numpy(gold[key]))
            print(f"   {key} rel-L2 {e:.2e}")
            assert e < (2e-3 if fp32 else max(4e-2, BF16_SLACK * ref["grad_full_worst"])), (key, e, ref)


def test_clip_vs_live_oracle_all_gradients():
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_caption_tokens as clip_tokens
    from oracle.weights import seeded_input

    model, sd, gold, c = _build(torch.float32)
    x = seeded_input((16, 3, 48, 80), seed=31)
    tok = clip_tokens(16, 12, c["vocab"], seed=31)  # shorter context than the table: positional embedding is interpolated 16 -> 12
    img, txt, loss, grads = _step(model, x.cuda(), tok.cuda())
    o_img, o_txt, o_loss, o_grads, _ = orc.clip_train_step(sd, x, tok, vit_mode=c["vit_mode"], text_layers=c["text_layers"], text_heads=c["text_heads"])
    assert l2_err(img, o_img) < 1e-4 and l2_err(txt, o_txt) < 1e-4
    assert abs(loss - float(o_loss)) < 1e-4
    gmax = max(float(v.norm()) for v in o_grads.values())
    for k, g in o_grads.items():
        assert l2_err(grads[k], g) < 2e-3 or g.norm() < 1e-5 * gmax, (k, l2_err(grads[k], g))


# ------------------------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 64), (37, 512), (256, 768), (3, 8)])
def test_l2_normalize(dtype, shape):
    from cvnets_amd import ops

    g = torch.Generator().manual_seed(7)
    x = torch.randn(*shape, generator=g).to(dtype)
    x[0] = 0  # zero row: norm clamped by eps
    dy = torch.randn(*shape, generator=g).to(dtype)
    xr = x.float().clone().requires_grad_()
    yr = F.normalize(xr, dim=-1)
    yr.backward(dy.float())
    xg = x.cuda().requires_grad_()
    y = ops.l2_normalize(xg)
    y.backward(dy.cuda())
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert l2_err(y.float().cpu(), yr.detach()) < tol
    assert l2_err(xg.grad.float().cpu()[1:], xr.grad[1:]) < (1e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(8, 8, 0), (16, 64, 32), (256, 2048, 1792), (5, 1000, 3)])
def test_scaled_cross_entropy(dtype, cfg):
    from cvnets_amd import ops

    N, M, off = cfg
    g = torch.Generator().manual_seed(8)
    logits = (torch.rand(N, M, generator=g) * 2 - 1).to(dtype)
    scale = torch.tensor(14.2857)
    lr, sr = logits.float().clone().requires_grad_(), scale.clone().requires_grad_()
    ref = F.cross_entropy(sr * lr, torch.arange(N) + off) * 0.5
    ref.backward()
    lg, sg = logits.cuda().requires_grad_(), scale.cuda().requires_grad_()
    loss = ops.scaled_cross_entropy(lg, sg, off) * 0.5
    loss.backward()
    f32 = dtype == torch.float32
    assert abs(float(loss.detach()) - float(ref.detach())) < (1e-5 if f32 else 1e-4) * max(1.0, abs(float(ref.detach())))
    assert l2_err(lg.grad.float().cpu(), lr.grad) < (1e-5 if f32 else 8e-3)
    assert abs(float(sg.grad) - float(sr.grad)) < (1e-5 if f32 else 1e-4) * max(1.0, abs(float(sr.grad)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rows_gather_idx(dtype):
    from cvnets_amd import ops

    B, S, E = 6, 11, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B * S, E, generator=g).to(dtype)
    rows = torch.arange(B) * S + torch.randint(0, S, (B,), generator=g)
    xg = x.cuda().requires_grad_()
    y = ops.RowsGatherIdx.apply(xg, rows.cuda())
    assert torch.equal(y.cpu(), x[rows])
    dy = torch.randn(B, E, generator=g).to(dtype)
    y.backward(dy.cuda())
    ref = torch.zeros_like(x)
    ref[rows] = dy
    assert torch.equal(xg.grad.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_text_embedding_layer(dtype):
    """Embedding mirror: lookup + positional add in one kernel, gradients for the table (padding row excluded) and the positions."""
    import cvnets_amd
    from cvnets_amd.layers import Embedding, default_opts

    cvnets_amd.set_compute_dtype(dtype)
    V, E, B, S = 50, 64, 4, 9
    emb = Embedding(default_opts(), V, E, padding_idx=0).cuda()
    g = torch.Generator().manual_seed(10)
    tok = torch.randint(0, V, (B, S), generator=g)
    pos = torch.randn(S, E, generator=g)
    dy = torch.randn(B, S, E, generator=g).to(dtype)
    w = emb.weight.detach().cpu().clone().requires_grad_()
    pr = pos.clone().requires_grad_()
    yr = F.embedding(tok, w, 0) + pr
    yr.backward(dy.float())
    pg = pos.cuda().requires_grad_()
    y = emb(tok.cuda(), pos=pg)
    y.backward(dy.cuda())
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert l2_err(y.float().cpu(), yr.detach()) < tol
    assert l2_err(emb.weight.grad.cpu(), w.grad) < 1e-5
    assert l2_err(pg.grad.cpu(), pr.grad) < 1e-5
