"""bf16 parity at the REAL width of BASELINE's configs 3 and 4 (ViT-B/16 at 224 x 224, CLIP ViT-B/16 + 12 x 512 text tower): the shapes on which
the bf16 large-tile kernels of the benchmark run — gemm_nt256 / gemm_nt128 (forward, dX), gemm_tn256 (dW with the bias fold), S = 197 attention
(head width 64), the 77-token causal text attention — compared as a whole train step against the oracle
(oracle/mobilevit_oracle.py: vit_forward / clip_train_step, cvnets/models/classification/vit.py:480-582, cvnets/models/multi_modal_img_text/clip.py,
cvnets/text_encoders/transformer.py:354-426) evaluated in float32.  The fixtures under tests/golden/ pin the oracle on the reference for the
`tiny` configuration; here the oracle itself runs at `base` width.  It is plain torch code and is executed ON THE GPU in float32 (ATen / rocBLAS
with TF32 off, MIOpen off — tests/conftest.py): a batch large enough to reach the large tiles (>= 1024 output tiles of 256 x 256 for ViT-B) would
take the CPU minutes.  It stays the checker: nothing of the product runs through it.

Bounds are ABSOLUTE, 2 x the deviation measured on MI355X when the test was written (round 6), as in tests/test_bf16_parity_gpu.py:
bf16 storage of every activation against an fp32 evaluation, not a kernel defect (oracle/bf16_points.py)."""
import pytest
import torch
import torch.nn.functional as F

from util import l2_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to(sd, dev):
    return {k: v.to(dev) for k, v in sd.items()}


def test_vit_base_224_bf16_train_step_vs_fp32_oracle():
    import cvnets_amd
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict

    B = 128  # 128 x 197 = 25216 token rows: 99 x 12 tiles of 256 x 256 for fc1 — the kernels bench_models.py times at 512 images
    model = cvnets_amd.build_vit("base", **{"model.classification.vit.dropout": 0.0})
    model.emb_dropout.p = 0.0
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    sd["cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": shapes["cls_token"]}, seed=0)["cls_token_values"]
    model.load_state_dict(sd)
    x, y = seeded_input((B, 3, 224, 224), seed=9).to(DEV), seeded_labels(B, 1000, seed=9).to(DEV)
    torch.backends.cuda.matmul.allow_tf32 = False
    o_logits, o_loss, o_grads, _ = orc.generic_train_step(orc.vit_forward, _to(sd, DEV), x, y, mode="base")
    o_logits, o_grads = o_logits.float().cpu(), {k: v.float().cpu() for k, v in o_grads.items()}
    cvnets_amd.set_compute_dtype(torch.bfloat16)
    try:
        model = model.to(DEV).train()
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = F.cross_entropy(logits.float(), y, label_smoothing=0.1)
        loss.backward()
        cvnets_amd.ops.finish_backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
    finally:
        cvnets_amd.set_compute_dtype(None)
    e_logits = l2_err(logits.float().cpu(), o_logits)
    e_loss = abs(float(loss) - float(o_loss)) / abs(float(o_loss))
    gmax = max(float(v.norm()) for v in o_grads.values())
    errs = sorted(((l2_err(grads[k], g), k) for k, g in o_grads.items() if g.norm() > 1e-4 * gmax), reverse=True)
    g_all = l2_err(torch.cat([grads[k].flatten() for k in o_grads]), torch.cat([g.flatten() for g in o_grads.values()]))
    print(f"[vit base 224 b{B} bf16] logits rel-L2 {e_logits:.2e}  loss rel {e_loss:.2e}  all gradients rel-L2 {g_all:.2e}  worst tensors {errs[:4]}")
    assert e_logits < VIT_BOUNDS["logits"] and e_loss < VIT_BOUNDS["loss"] and g_all < VIT_BOUNDS["grad_all"], (e_logits, e_loss, g_all)
    assert errs[0][0] < VIT_BOUNDS["grad_worst"], errs[:4]


def test_clip_base_bf16_train_step_vs_fp32_oracle():
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_caption_tokens, seeded_input, seeded_state_dict

    B, ctx, vocab = 64, 77, 49408
    model = cvnets_amd.build_clip(**{"model.classification.vit.mode": "base", "model.text.transformer.model_dim": 512,
                                     "model.text.transformer.n_transformer_layers": 12, "model.text.transformer.n_heads_per_layer": 8,
                                     "dataset.text_vocab_size": vocab, "dataset.text_context_length": ctx,
                                     "model.multi_modal_image_text.clip.projection_dim": 512})
    model.image_encoder.emb_dropout.p = 0.0
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    sd["logit_scale"] = torch.tensor(float(torch.log(torch.tensor(1.0 / 0.07))))
    sd["image_encoder.cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": shapes["image_encoder.cls_token"]}, seed=0)["cls_token_values"]
    model.load_state_dict(sd, strict=True)
    x = seeded_input((B, 3, 224, 224), seed=13).to(DEV)
    tok = seeded_caption_tokens(B, ctx, vocab, seed=13).to(DEV)
    torch.backends.cuda.matmul.allow_tf32 = False
    o_img, o_txt, o_loss, o_grads, _ = orc.clip_train_step(_to(sd, DEV), x, tok, vit_mode="base", text_layers=12, text_heads=8)
    o_img, o_txt, o_grads = o_img.float().cpu(), o_txt.float().cpu(), {k: v.float().cpu() for k, v in o_grads.items()}
    cvnets_amd.set_compute_dtype(torch.bfloat16)
    try:
        model = model.to(DEV).train()
        loss_fn = cvnets_amd.ContrastiveLossClip(default_opts()).train()
        model.zero_grad(set_to_none=True)
        out = model({"image": x, "text": tok})
        img, txt = out["image"].detach().float().cpu(), out["text"].detach().float().cpu()
        loss = loss_fn(None, out)["total_loss"]
        loss.backward()
        cvnets_amd.ops.finish_backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
    finally:
        cvnets_amd.set_compute_dtype(None)
    e_img, e_txt = l2_err(img, o_img), l2_err(txt, o_txt)
    e_loss = abs(float(loss) - float(o_loss)) / abs(float(o_loss))
    gmax = max(float(v.norm()) for v in o_grads.values())
    errs = sorted(((l2_err(grads[k], g), k) for k, g in o_grads.items() if g.norm() > 1e-4 * gmax), reverse=True)
    g_all = l2_err(torch.cat([grads[k].flatten() for k in o_grads]), torch.cat([g.flatten() for g in o_grads.values()]))
    print(f"[clip base b{B} bf16] image {e_img:.2e} text {e_txt:.2e} loss rel {e_loss:.2e} all gradients rel-L2 {g_all:.2e} worst tensors {errs[:4]}")
    assert e_img < CLIP_BOUNDS["image"] and e_txt < CLIP_BOUNDS["text"] and e_loss < CLIP_BOUNDS["loss"] and g_all < CLIP_BOUNDS["grad_all"]
    assert errs[0][0] < CLIP_BOUNDS["grad_worst"], errs[:4]


# 2 x the deviations measured on MI355X in round 6 (printed by a run with -s):
#   ViT-B/16 224 x 224, 128 images: logits 9.0e-3, loss 1.5e-4, all gradients 1.1e-2, worst tensor (cls_token) 1.6e-2
#   CLIP ViT-B/16 + 12 x 512, 64 pairs: image 8.8e-3, text 1.0e-2, loss 8.3e-5, all gradients 2.8e-2, worst tensor (a LayerNorm weight) 3.9e-2
VIT_BOUNDS = {"logits": 1.8e-2, "loss": 1e-3, "grad_all": 2.2e-2, "grad_worst": 3.3e-2}
CLIP_BOUNDS = {"image": 1.8e-2, "text": 2.1e-2, "loss": 1e-3, "grad_all": 5.6e-2, "grad_worst": 7.8e-2}
