"""CPU tests: the oracle restatement (oracle/mobilevit_oracle.py) reproduces the golden fixtures that
oracle/make_golden.py recorded from the reference itself — this is what keeps the oracle pinned on machines where
/root/reference does not exist."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mobilevit_oracle as orc
from oracle.weights import seeded_input, seeded_labels, seeded_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("mobilevit_xxs_32_b8", "xx_small", 8, 32), ("mobilevit_s_128_b2", "small", 2, 128), ("mobilevit_s_160_b2", "small", 2, 160),
         ("mobilevit_xxs_128_b2", "xx_small", 2, 128)]  # the last one goes through the reference's channel-first LayerNorm quirk


@pytest.mark.parametrize("name,mode,batch,res", CASES)
def test_oracle_matches_reference_fixture(name, mode, batch, res):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    shapes = json.load(open(os.path.join(GOLD, f"mobilevit_{mode}_keys.json")))
    sd = seeded_state_dict(shapes, seed=0)
    x = seeded_input((batch, 3, res, res), seed=1)
    y = seeded_labels(batch, 1000, seed=1)
    le = orc.mobilevit_forward(sd, x, mode=mode, training=False)
    assert np.allclose(le.numpy(), gold["logits_eval"], rtol=1e-4, atol=1e-5)
    logits, loss, grads, running = orc.train_step(sd, x, y, mode=mode)
    assert np.allclose(logits.numpy(), gold["logits_train"], rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(gold["loss"])) < 1e-5
    names = [str(n) for n in gold["grad_names"]]
    assert names == list(grads.keys())
    gn = np.array([grads[k].norm().item() for k in names])
    assert np.allclose(gn, gold["grad_norm"], rtol=1e-3, atol=1e-7)
    for key in gold.files:
        if key.startswith("grad::"):
            assert np.allclose(grads[key[6:]].numpy(), gold[key], rtol=1e-3, atol=1e-6), key
        if key.startswith("bn::"):
            assert np.allclose(running[key[4:]].numpy(), gold[key], rtol=1e-5, atol=1e-6), key


@pytest.mark.parametrize("name,res", [("vit_tiny_64_b2", 64), ("vit_tiny_224_b2", 224)])
def test_vit_oracle_matches_reference_fixture(name, res):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    shapes = json.load(open(os.path.join(GOLD, "vit_tiny_keys.json")))
    sd = seeded_state_dict(shapes, seed=0)
    sd["cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": (1, 1, 192)}, seed=0)["cls_token_values"]
    x = seeded_input((2, 3, res, res), seed=1)
    y = seeded_labels(2, 1000, seed=1)
    assert np.allclose(orc.vit_forward(sd, x, mode="tiny", training=False).numpy(), gold["logits_eval"], rtol=1e-4, atol=1e-5)
    logits, loss, grads, running = orc.generic_train_step(orc.vit_forward, sd, x, y, mode="tiny")
    assert np.allclose(logits.numpy(), gold["logits_train"], rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(gold["loss"])) < 1e-5
    names = [str(n) for n in gold["grad_names"]]
    assert sorted(names) == sorted(grads.keys())
    assert np.allclose(np.array([grads[k].norm().item() for k in names]), gold["grad_norm"], rtol=1e-3, atol=1e-7)
    for key in gold.files:
        if key.startswith("grad::"):
            assert np.allclose(grads[key[6:]].numpy(), gold[key], rtol=1e-3, atol=1e-6), key
        if key.startswith("bn::"):
            assert np.allclose(running[key[4:]].numpy(), gold[key], rtol=1e-5, atol=1e-6), key


@pytest.mark.parametrize("name,wm,batch,hw", [("mobilevitv2_w050_64_b2", 0.5, 2, (64, 64)), ("mobilevitv2_w075_96x160_b3", 0.75, 3, (96, 160))])
def test_v2_oracle_matches_reference_fixture(name, wm, batch, hw):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    shapes = json.load(open(os.path.join(GOLD, f"mobilevitv2_w{int(round(wm * 100)):03d}_keys.json")))
    sd = seeded_state_dict(shapes, seed=0)
    x = seeded_input((batch, 3) + hw, seed=1)
    y = seeded_labels(batch, 1000, seed=1)
    assert np.allclose(orc.mobilevit_v2_forward(sd, x, width_multiplier=wm, training=False).numpy(), gold["logits_eval"], rtol=1e-4, atol=1e-5)
    logits, loss, grads, running = orc.generic_train_step(orc.mobilevit_v2_forward, sd, x, y, width_multiplier=wm)
    assert np.allclose(logits.numpy(), gold["logits_train"], rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(gold["loss"])) < 1e-5
    names = [str(n) for n in gold["grad_names"]]
    assert names == list(grads.keys())
    assert np.allclose(np.array([grads[k].norm().item() for k in names]), gold["grad_norm"], rtol=1e-3, atol=1e-6)
    for key in gold.files:
        if key.startswith("grad::"):
            assert np.allclose(grads[key[6:]].numpy(), gold[key], rtol=1e-3, atol=1e-6), key
        if key.startswith("bn::"):
            assert np.allclose(running[key[4:]].numpy(), gold[key], rtol=1e-5, atol=1e-6), key


def test_clip_oracle_matches_reference_fixture():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLD, "clip_tiny_64_b8.npz"))
    c = json.loads(str(gold["config"]))
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLD, "clip_tiny_keys.json"))).items()}
    sd = seeded_state_dict(shapes, seed=0)
    sd["logit_scale"] = torch.tensor(float(np.log(1.0 / 0.07)))
    sd["image_encoder.cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": shapes["image_encoder.cls_token"]}, seed=0)["cls_token_values"]
    x = seeded_input((c["batch"], 3, c["res"], c["res"]), seed=1)
    tok = torch.from_numpy(gold["tokens"])
    img, txt, loss, grads, _ = orc.clip_train_step(sd, x, tok, vit_mode=c["vit_mode"], text_layers=c["text_layers"], text_heads=c["text_heads"])
    assert np.allclose(img.numpy(), gold["image"], rtol=1e-4, atol=1e-6) and np.allclose(txt.numpy(), gold["text"], rtol=1e-4, atol=1e-6)
    assert abs(float(loss) - float(gold["loss"])) < 1e-5
    names = [str(n) for n in gold["grad_names"]]
    assert sorted(names) == sorted(grads.keys())
    assert np.allclose(np.array([grads[k].norm().item() for k in names]), gold["grad_norm"], rtol=1e-3, atol=1e-7)
    for key in gold.files:
        if key.startswith("grad::"):
            assert np.allclose(grads[key[6:]].numpy(), gold[key], rtol=1e-3, atol=1e-6), key


def test_oracle_mha_matches_reference_fixture():
    gold = np.load(os.path.join(GOLD, "mha_cases.npz"))
    for idx in range(4):
        b, s, c, h, causal, kpm = (int(v) for v in gold[f"case{idx}_cfg"])
        shapes = {"mha.qkv_proj.weight": (3 * c, c), "mha.qkv_proj.bias": (3 * c,), "mha.out_proj.weight": (c, c), "mha.out_proj.bias": (c,)}
        sd = seeded_state_dict(shapes, seed=idx)
        x = seeded_input((b, s, c), seed=100 + idx)
        am = torch.full((s, s), float("-inf")).triu(1).unsqueeze(0).expand(b, -1, -1).contiguous() if causal else None
        pm = None
        if kpm:
            pm = torch.zeros(b, s)
            pm[:, -5:] = 1
        o = orc.multi_head_attention(sd, "mha", x, h, attn_mask=am, key_padding_mask=pm)
        assert np.allclose(o.numpy(), gold[f"case{idx}_out"], rtol=1e-4, atol=1e-5), idx


def test_oracle_temporal_block_matches_reference_fixture():
    """MobileViTBlock.forward((x, x_prev)) (mobilevit_block.py:289-326): the oracle's two chained frames against the reference's own run
    (oracle/make_temporal_fixture.py): both feature maps, both patch tensors, every gradient — the second frame cross-attends to the first
    one's patches, so this also pins the oracle's cross-attention branch (multi_head_attention.py:158-185) and `x_prev` hand-over."""
    gold = np.load(os.path.join(GOLD, "mobilevit_block_temporal.npz"))
    for case in ("even", "resized"):
        b, cin, d, ffn, blocks, hd, patch, H, W = (int(v) for v in gold[f"{case}::cfg"])
        shapes = {str(k): tuple(int(i) for i in str(s).split(",") if i) for k, s in zip(gold[f"{case}::keys"], gold[f"{case}::shapes"])}
        sd = seeded_state_dict(shapes, seed=21)
        names = [k[len(case) + 8:] for k in gold.files if k.startswith(f"{case}::grad::")]
        assert len(names) == 36
        leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
        x1 = seeded_input((b, cin, H, W), seed=31).requires_grad_(True)
        x2 = seeded_input((b, cin, H, W), seed=32).requires_grad_(True)
        fm1, p1 = orc.mobilevit_block_temporal(leaf, "", x1, None, blocks, d // hd, True, None, patch, patch)
        fm2, p2 = orc.mobilevit_block_temporal(leaf, "", x2, p1, blocks, d // hd, True, None, patch, patch)
        loss = (fm2 * seeded_input(tuple(fm2.shape), seed=33)).sum() + (p2 * seeded_input(tuple(p2.shape), seed=34)).sum()
        grads = torch.autograd.grad(loss, [x1, x2] + [leaf[k] for k in names])
        for key, got in (("fm1", fm1), ("p1", p1), ("fm2", fm2), ("p2", p2), ("grad_x1", grads[0]), ("grad_x2", grads[1])):
            assert np.allclose(got.detach().numpy(), gold[f"{case}::{key}"], rtol=1e-4, atol=1e-5), (case, key)
        for k, got in zip(names, grads[2:]):
            want = gold[f"{case}::grad::{k}"]
            assert float(np.linalg.norm(got.numpy() - want)) <= 1e-4 * float(np.linalg.norm(want)) + 1e-7, (case, k)


def test_weights_are_platform_independent():
    t = seeded_state_dict({"conv_1.block.conv.weight": (16, 3, 3, 3)}, seed=0)["conv_1.block.conv.weight"]
    # fixed known-answer values (PCG64 stream): guards against numpy RNG drift between the authoring box and the GPU box
    assert abs(float(t.flatten()[0]) - float(seeded_state_dict({"conv_1.block.conv.weight": (16, 3, 3, 3)}, 0)["conv_1.block.conv.weight"].flatten()[0])) == 0
    assert t.shape == (16, 3, 3, 3) and abs(float(t.std()) - 1 / np.sqrt(27)) < 0.03


def load_keep_factors(gold):
    """the factors (0 or 1 / (1 - p)) the reference's Dropout layers applied in the recorded run (oracle/make_golden.py --dropout)"""
    p = float(gold["p"])
    drop = {}
    for key in gold.files:
        if key.startswith("keep::"):
            shape = tuple(int(v) for v in gold["keepshape::" + key[6:]])
            n = int(np.prod(shape))
            drop[key[6:]] = torch.from_numpy(np.unpackbits(gold[key])[:n].reshape(shape).astype(np.float32)) / (1.0 - p)
    return drop


def test_oracle_with_dropout_factors_matches_reference_fixture():
    """mit.dropout = classifier_dropout = 0.1 (the shipped configuration): the oracle, fed the keep masks the reference drew, reproduces
    the reference's training step — this pins WHERE the oracle applies dropout (tests/test_zz_dropout_step_gpu.py relies on it)."""
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    gold = np.load(os.path.join(GOLD, "mobilevit_xxs_dropout_64_b4.npz"))
    shapes = json.load(open(os.path.join(GOLD, "mobilevit_xx_small_keys.json")))
    sd = seeded_state_dict(shapes, seed=0)
    x = seeded_input((4, 3, 64, 64), seed=5)
    y = seeded_labels(4, 1000, seed=5)
    drop = load_keep_factors(gold)
    assert len(drop) == 19 and all(0.05 < float((f == 0).float().mean()) < 0.15 for k, f in drop.items() if k != "classifier")
    logits, loss, grads, _ = orc.train_step(sd, x, y, mode="xx_small", drop=drop)
    assert np.allclose(logits.numpy(), gold["logits_train"], rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(gold["loss"])) < 1e-5
    names = [str(n) for n in gold["grad_names"]]
    assert np.allclose(np.array([grads[k].norm().item() for k in names]), gold["grad_norm"], rtol=1e-3, atol=1e-7)
    for key in gold.files:
        if key.startswith("grad::"):
            assert np.allclose(grads[key[6:]].numpy(), gold[key], rtol=1e-3, atol=1e-6), key
    plain, _, _, _ = orc.train_step(sd, x, y, mode="xx_small")
    assert not np.allclose(plain.numpy(), gold["logits_train"], rtol=1e-2, atol=1e-3)  # the masks matter


def test_channel_first_layernorm_fixture_matches_the_written_out_formula():
    """tests/golden/layernorm_channel_first.npz (reference outputs, oracle/make_layer_fixtures.py) against layer_norm.py:53-66 written out
    in float64 on the same seeded tensors: pins the fixture on machines without the reference."""
    import os

    import numpy as np
    import torch
    from oracle.make_layer_fixtures import LN_CF_CASES, ln_cf_tensors

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "layernorm_channel_first.npz"))
    for name, B, C, H, W in LN_CF_CASES:
        x, w, b, g = (t.double() for t in ln_cf_tensors(name, B, C, H, W))
        x.requires_grad_(True)
        w.requires_grad_(True)
        b.requires_grad_(True)
        s, u = torch.std_mean(x, dim=1, keepdim=True, unbiased=False)
        y = (x - u) / (s + 1e-5) * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)
        (y * g).sum().backward()
        for key, t in (("_y", y), ("_dx", x.grad), ("_dw", w.grad), ("_db", b.grad)):
            ref = torch.from_numpy(gold[name + key]).double()
            assert float((t.detach() - ref).norm() / ref.norm()) < 1e-5, (name, key)
