#!/usr/bin/env python
"""ORACLE tooling — generates tests/golden/* by RUNNING THE REFERENCE ITSELF.

Runs only in the authoring container (needs /root/reference; the GPU box has no copy).
  python oracle/make_golden.py

For each case it
  1. builds the reference model through ``cvnets.get_model(opts)`` (opts = cvnets.modeling_arguments
     defaults + the flattened reference YAML) under the torchvision shim,
  2. loads the deterministic weights of oracle/weights.py, runs a train-mode fwd + label-smoothed
     CE + bwd and an eval-mode fwd on the seeded input, all fp32 on CPU, dropout p = 0,
  3. asserts that the oracle restatement (oracle/mobilevit_oracle.py) reproduces the reference
     (logits, loss, every gradient, BN running stats) to fp32 round-off — this is what pins the
     oracle — and
  4. writes the REFERENCE's outputs to tests/golden/<case>.npz (+ the key/shape manifest).
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from oracle import mobilevit_oracle as orc  # noqa: E402
from oracle.weights import seeded_caption_tokens as clip_tokens  # noqa: E402
from oracle.weights import seeded_input, seeded_labels, seeded_state_dict  # noqa: E402

CASES = [
    # name, mode, batch, H=W    (BASELINE.json configs[0] and [1] shapes; plus a mid-size case)
    ("mobilevit_xxs_32_b8", "xx_small", 8, 32),
    ("mobilevit_s_128_b2", "small", 2, 128),
    ("mobilevit_s_256_b2", "small", 2, 256),
    ("mobilevit_s_160_b2", "small", 2, 160),   # VBS resolution (mobilevit.yaml sampler 160..320)
    # layer_3 sees 64 patches x 64 channels: the reference LayerNorm takes its channel-first branch (layer_norm.py:53-66)
    ("mobilevit_xxs_128_b2", "xx_small", 2, 128),
]
FULL_GRADS = [
    "conv_1.block.conv.weight", "conv_1.block.norm.weight", "conv_1.block.norm.bias",
    "layer_1.0.block.conv_3x3.block.conv.weight",
    "layer_3.1.global_rep.0.pre_norm_mha.1.qkv_proj.bias",
    "layer_3.1.global_rep.0.pre_norm_mha.0.weight",
    "layer_5.1.global_rep.2.pre_norm_ffn.4.bias",
    "layer_5.1.fusion.block.norm.weight",
    "classifier.fc.bias",
]


def grad_sample_fields(ref_grads):
    """large-batch fixtures: every gradient at oracle.weights.sample_indices positions + per-tensor L1 norm and signed sum (for the
    absolute bf16 bounds and the signed-bias check of tests/test_bf16_parity_gpu.py) — a few hundred KB instead of the full gradients"""
    from oracle.weights import sample_indices
    names = list(ref_grads.keys())
    out = {"grad_l1": np.array([ref_grads[k].double().abs().sum().item() for k in names], dtype=np.float64),
           "grad_sum": np.array([ref_grads[k].double().sum().item() for k in names], dtype=np.float64),
           "grad_numel": np.array([ref_grads[k].numel() for k in names], dtype=np.int64)}
    for k in names:
        out["gsamp::" + k] = ref_grads[k].reshape(-1)[torch.from_numpy(sample_indices(k, ref_grads[k].numel()))].numpy()
    return out


def build_reference_model(mode: str, dropout: float = 0.0, classifier_dropout: float = 0.0):
    os.chdir(REF)
    import cvnets
    from options.utils import flatten_yaml_as_dict

    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/classification/imagenet/mobilevit.yaml")))
    for k, v in cfg.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    setattr(opts, "dataset.category", "classification")
    setattr(opts, "dev.device", "cpu")
    setattr(opts, "model.classification.mit.mode", mode)
    # parity configuration: dropout off (torch Philox streams are not reproducible across impls)
    setattr(opts, "model.classification.mit.dropout", dropout)
    setattr(opts, "model.classification.mit.attn_dropout", 0.0)
    setattr(opts, "model.classification.mit.ffn_dropout", 0.0)
    setattr(opts, "model.classification.classifier_dropout", classifier_dropout)
    return cvnets.get_model(opts)


def run_dropout_case(outdir, name="mobilevit_xxs_dropout_64_b4", mode="xx_small", batch=4, res=64, p=0.1):
    """The shipped training configuration has mit.dropout = 0.1 and classifier_dropout = 0.1 (config/classification/imagenet/mobilevit.yaml).
    The reference draws its masks from torch's generator; forward hooks on its Dropout layers record the factor (0 or 1 / (1 - p)) each
    one applied, the oracle is run with exactly those factors (oracle.mobilevit_oracle.train_step(drop=...)) and must reproduce the
    reference's logits, loss and gradients to fp32 round-off: this pins WHERE the oracle applies dropout.  The fixture keeps the keep bits."""
    torch.manual_seed(0)
    model = build_reference_model(mode, dropout=p, classifier_dropout=p)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    x = seeded_input((batch, 3, res, res), seed=5)
    y = seeded_labels(batch, 1000, seed=5)
    model.train()
    drop, hooks = {}, []
    for nm, m in model.named_modules():
        if type(m).__name__ == "Dropout" and m.p > 0:
            if nm == "classifier.dropout":
                key = "classifier"
            elif nm.endswith(".pre_norm_mha.2"):
                key = nm[: -len(".pre_norm_mha.2")] + ".mha"
            elif nm.endswith(".pre_norm_ffn.5"):
                key = nm[: -len(".pre_norm_ffn.5")] + ".ffn"
            else:
                raise RuntimeError("unexpected Dropout layer " + nm)
            hooks.append(m.register_forward_hook(lambda m_, i, o, key=key: drop.__setitem__(key, (o.detach() != 0).float() / (1.0 - m_.p))))
    torch.manual_seed(1234)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1)
    model.zero_grad()
    loss.backward()
    for h in hooks:
        h.remove()
    ref_grads = {k: q.grad.detach().clone() for k, q in model.named_parameters()}
    assert len(drop) == 2 * (2 + 4 + 3) + 1, sorted(drop)
    o_logits, o_loss, o_grads, _ = orc.train_step(sd, x, y, mode=mode, drop=drop)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    checks = {"logits": rel(o_logits, logits.detach()), "loss": abs(float(o_loss) - float(loss)),
              "grad_worst_rel": max(rel(o_grads[k], ref_grads[k]) for k in ref_grads)}
    # without the factors the same comparison must be far off (the hooks did record something that matters)
    n_logits, _, _, _ = orc.train_step(sd, x, y, mode=mode)
    checks["logits_without_masks"] = rel(n_logits, logits.detach())
    print(name, {k: f"{v:.2e}" for k, v in checks.items()})
    assert checks["logits"] < 1e-5 and checks["loss"] < 1e-5 and checks["grad_worst_rel"] < 2e-4 and checks["logits_without_masks"] > 1e-2, checks
    names = list(ref_grads.keys())
    out = {"logits_train": logits.detach().numpy(), "loss": np.float32(loss.item()), "grad_names": np.array(names),
           "grad_norm": np.array([ref_grads[k].norm().item() for k in names], dtype=np.float64), "p": np.float32(p),
           "oracle_vs_reference": np.array(json.dumps(checks))}
    for k in FULL_GRADS:
        out["grad::" + k] = ref_grads[k].numpy()
    for k, f in drop.items():
        out["keep::" + k] = np.packbits((f != 0).numpy().reshape(-1))
        out["keepshape::" + k] = np.array(f.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)


def run_case(name, mode, batch, res, outdir):
    torch.manual_seed(0)
    model = build_reference_model(mode)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    x = seeded_input((batch, 3, res, res), seed=1)
    y = seeded_labels(batch, 1000, seed=1)

    # --- reference: eval forward, then train step ---
    model.eval()
    with torch.no_grad():
        logits_eval = model(x).clone()
    model.train()
    taps = {}
    hooks = []
    for nm in ("conv_1", "layer_1", "layer_2", "layer_3", "layer_4", "layer_5"):
        hooks.append(getattr(model, nm).register_forward_hook(
            lambda m, i, o, nm=nm: taps.__setitem__(nm, o.detach().clone())))
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1)
    model.zero_grad()
    loss.backward()
    for h in hooks:
        h.remove()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    ref_sd_after = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # --- the reference's OWN bf16 path (torch.autocast on CPU, same policy as engine/utils.py autocast_fn on CUDA) vs its
    #     fp32 path: this is the error level inherent to bf16 storage/compute for this case; the bf16 tolerances of
    #     tests/test_model_gpu.py are expressed relative to it ---
    model.load_state_dict(sd, strict=True)
    model.train()
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lb = model(x)
        loss_b = torch.nn.functional.cross_entropy(lb.float(), y, label_smoothing=0.1)
    loss_b.backward()
    gb = {k: p.grad.detach().clone().float() for k, p in model.named_parameters()}

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    bf16_ref = {
        "logits_train": rel(lb.detach().float(), logits.detach()),
        "loss": abs(float(loss_b) - float(loss)),
        "grad_global": (sum(float((gb[k].double() - ref_grads[k].double()).pow(2).sum()) for k in gb) /
                        sum(float(ref_grads[k].double().pow(2).sum()) for k in gb)) ** 0.5,
        "grad_norm_worst": max(abs(gb[k].norm().item() - ref_grads[k].norm().item()) /
                               (ref_grads[k].norm().item() + 1e-3 * max(v.norm().item() for v in ref_grads.values())) for k in gb),
        "grad_full_worst": max(rel(gb[k], ref_grads[k]) for k in FULL_GRADS),
    }
    print(name, "reference bf16-autocast vs fp32:", {k: f"{v:.2e}" for k, v in bf16_ref.items()})
    model.load_state_dict(sd, strict=True)

    # --- oracle restatement on the same weights: must agree to fp32 round-off ---
    o_eval = orc.mobilevit_forward(sd, x, mode=mode, training=False)
    o_logits, o_loss, o_grads, o_running = orc.train_step(sd, x, y, mode=mode)

    checks = {"logits_eval": rel(o_eval, logits_eval), "logits_train": rel(o_logits, logits.detach()),
              "loss": abs(float(o_loss) - float(loss))}
    worst_g = max(rel(o_grads[k], ref_grads[k]) for k in ref_grads)
    worst_bn = max(rel(v, ref_sd_after[k]) for k, v in o_running.items())
    checks["grad_worst_rel"] = worst_g
    checks["bn_running_worst_rel"] = worst_bn
    print(name, {k: f"{v:.2e}" for k, v in checks.items()})
    assert checks["logits_eval"] < 1e-5 and checks["logits_train"] < 1e-5, checks
    assert checks["loss"] < 1e-5 and worst_g < 2e-4 and worst_bn < 1e-5, checks
    assert set(o_grads) == set(ref_grads)

    names = list(ref_grads.keys())
    out = {
        "logits_train": logits.detach().numpy(),
        "logits_eval": logits_eval.numpy(),
        "loss": np.float32(loss.item()),
        "grad_names": np.array(names),
        "grad_norm": np.array([ref_grads[k].norm().item() for k in names], dtype=np.float64),
        "grad_sum": np.array([ref_grads[k].double().sum().item() for k in names], dtype=np.float64),
        "oracle_vs_reference": np.array(json.dumps(checks)),
        "ref_bf16_autocast_err": np.array(json.dumps(bf16_ref)),
    }
    for k in FULL_GRADS:
        out["grad::" + k] = ref_grads[k].numpy()
    for k in ("conv_1.block.norm.running_mean", "conv_1.block.norm.running_var",
              "layer_3.1.fusion.block.norm.running_mean", "layer_3.1.fusion.block.norm.running_var",
              "conv_1x1_exp.block.norm.running_var"):
        out["bn::" + k] = ref_sd_after[k].numpy()
    if batch >= 16:
        out.update(grad_sample_fields(ref_grads))
    for k, t in taps.items():
        out["tap_stats::" + k] = np.array([t.mean().item(), t.std().item(), t.abs().max().item()], dtype=np.float64)
        out["tap_slice::" + k] = t[0, : min(8, t.shape[1]), : min(4, t.shape[2]), : min(4, t.shape[3])].numpy()
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    with open(os.path.join(outdir, f"mobilevit_{mode}_keys.json"), "w") as f:
        json.dump({k: list(s) for k, s in shapes.items()}, f, indent=0)


VIT_CASES = [("vit_tiny_64_b2", "tiny", 2, 64), ("vit_tiny_224_b2", "tiny", 2, 224)]
VIT_FULL_GRADS = ["cls_token", "pos_embed.pos_embed.pos_embed", "patch_emb.0.block.conv.weight", "patch_emb.1.block.conv.weight",
                  "patch_emb.2.block.conv.bias", "transformer.0.pre_norm_mha.1.qkv_proj.bias", "transformer.11.pre_norm_ffn.4.bias",
                  "post_transformer_norm.weight", "classifier.bias"]


def build_reference_vit(mode: str):
    os.chdir(REF)
    import cvnets
    from options.utils import flatten_yaml_as_dict

    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/classification/imagenet/vit.yaml")))
    for k, v in cfg.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    setattr(opts, "dataset.category", "classification")
    setattr(opts, "dev.device", "cpu")
    setattr(opts, "model.classification.vit.mode", mode)
    setattr(opts, "model.classification.vit.dropout", 0.0)          # parity configuration: dropout off
    setattr(opts, "model.classification.gradient_checkpointing", False)
    model = cvnets.get_model(opts)
    model.emb_dropout.p = 0.0                                        # config/vit.py hard-codes 0.1 for "tiny"
    return model


def run_vit_case(name, mode, batch, res, outdir):
    torch.manual_seed(0)
    model = build_reference_vit(mode)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    sd["cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": (1, 1, shapes["cls_token"][2])}, seed=0)["cls_token_values"]
    model.load_state_dict(sd, strict=True)
    x = seeded_input((batch, 3, res, res), seed=1)
    y = seeded_labels(batch, 1000, seed=1)
    model.eval()
    with torch.no_grad():
        logits_eval = model(x).clone()
    model.train()
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1)
    model.zero_grad()
    loss.backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    ref_sd_after = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lb = model(x)
        loss_b = torch.nn.functional.cross_entropy(lb.float(), y, label_smoothing=0.1)
    loss_b.backward()
    gb = {k: p.grad.detach().clone().float() for k, p in model.named_parameters()}

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    gmax = max(v.norm().item() for v in ref_grads.values())
    bf16_ref = {"logits_train": rel(lb.detach().float(), logits.detach()), "loss": abs(float(loss_b) - float(loss)),
                "grad_global": (sum(float((gb[k].double() - ref_grads[k].double()).pow(2).sum()) for k in gb) /
                                sum(float(ref_grads[k].double().pow(2).sum()) for k in gb)) ** 0.5,
                "grad_norm_worst": max(abs(gb[k].norm().item() - ref_grads[k].norm().item()) / (ref_grads[k].norm().item() + 1e-3 * gmax) for k in gb),
                "grad_full_worst": max(rel(gb[k], ref_grads[k]) for k in VIT_FULL_GRADS)}
    print(name, "reference bf16-autocast vs fp32:", {k: f"{v:.2e}" for k, v in bf16_ref.items()})
    o_eval = orc.vit_forward(sd, x, mode=mode, training=False)
    o_logits, o_loss, o_grads, o_running = orc.generic_train_step(orc.vit_forward, sd, x, y, mode=mode)
    checks = {"logits_eval": rel(o_eval, logits_eval), "logits_train": rel(o_logits, logits.detach()), "loss": abs(float(o_loss) - float(loss)),
              "grad_worst_rel": max(rel(o_grads[k], ref_grads[k]) for k in ref_grads if ref_grads[k].norm() > 1e-6 * gmax),
              "bn_running_worst_rel": max(rel(v, ref_sd_after[k]) for k, v in o_running.items())}
    print(name, {k: f"{v:.2e}" for k, v in checks.items()})
    assert checks["logits_eval"] < 1e-5 and checks["logits_train"] < 1e-5 and checks["loss"] < 1e-5, checks
    assert checks["grad_worst_rel"] < 2e-4 and checks["bn_running_worst_rel"] < 1e-5, checks
    names = list(ref_grads.keys())
    out = {"logits_train": logits.detach().numpy(), "logits_eval": logits_eval.numpy(), "loss": np.float32(loss.item()),
           "grad_names": np.array(names), "grad_norm": np.array([ref_grads[k].norm().item() for k in names], dtype=np.float64),
           "oracle_vs_reference": np.array(json.dumps(checks)), "ref_bf16_autocast_err": np.array(json.dumps(bf16_ref))}
    for k in VIT_FULL_GRADS:
        out["grad::" + k] = ref_grads[k].numpy()
    for k in ("patch_emb.0.block.norm.running_mean", "patch_emb.1.block.norm.running_var"):
        out["bn::" + k] = ref_sd_after[k].numpy()
    if batch >= 16:
        out.update(grad_sample_fields(ref_grads))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    with open(os.path.join(outdir, f"vit_{mode}_keys.json"), "w") as f:
        json.dump({k: list(s) for k, s in shapes.items()}, f, indent=0)


V2_CASES = [("mobilevitv2_w050_64_b2", 0.5, 2, 64), ("mobilevitv2_w100_224_b2", 1.0, 2, 224), ("mobilevitv2_w075_96x160_b3", 0.75, 3, (96, 160))]
V2_FULL_GRADS = ["conv_1.block.conv.weight", "layer_3.1.local_rep.0.block.conv.weight", "layer_3.1.global_rep.0.pre_norm_attn.0.weight",
                 "layer_3.1.global_rep.0.pre_norm_attn.1.qkv_proj.block.conv.weight", "layer_3.1.global_rep.0.pre_norm_attn.1.qkv_proj.block.conv.bias",
                 "layer_4.1.global_rep.3.pre_norm_ffn.1.block.conv.weight", "layer_5.1.global_rep.3.weight", "layer_5.1.conv_proj.block.norm.weight",
                 "classifier.1.bias"]


def build_reference_v2(wm: float):
    os.chdir(REF)
    import cvnets
    from options.utils import flatten_yaml_as_dict

    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/classification/imagenet/mobilevit_v2.yaml")))
    for k, v in cfg.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    setattr(opts, "dataset.category", "classification")
    setattr(opts, "dev.device", "cpu")
    setattr(opts, "model.classification.mitv2.width_multiplier", wm)
    return cvnets.get_model(opts)


def run_v2_case(name, wm, batch, res, outdir):
    torch.manual_seed(0)
    model = build_reference_v2(wm)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    hw = res if isinstance(res, tuple) else (res, res)
    x = seeded_input((batch, 3) + hw, seed=1)
    y = seeded_labels(batch, 1000, seed=1)
    model.eval()
    with torch.no_grad():
        logits_eval = model(x).clone()
    model.train()
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y, label_smoothing=0.1)
    model.zero_grad()
    loss.backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    ref_sd_after = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lb = model(x)
        loss_b = torch.nn.functional.cross_entropy(lb.float(), y, label_smoothing=0.1)
    loss_b.backward()
    gb = {k: p.grad.detach().clone().float() for k, p in model.named_parameters()}

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    gmax = max(v.norm().item() for v in ref_grads.values())
    bf16_ref = {"logits_train": rel(lb.detach().float(), logits.detach()), "loss": abs(float(loss_b.detach()) - float(loss.detach())),
                "grad_global": (sum(float((gb[k].double() - ref_grads[k].double()).pow(2).sum()) for k in gb) /
                                sum(float(ref_grads[k].double().pow(2).sum()) for k in gb)) ** 0.5,
                "grad_norm_worst": max(abs(gb[k].norm().item() - ref_grads[k].norm().item()) / (ref_grads[k].norm().item() + 1e-3 * gmax) for k in gb),
                "grad_full_worst": max(rel(gb[k], ref_grads[k]) for k in V2_FULL_GRADS)}
    print(name, "reference bf16-autocast vs fp32:", {k: f"{v:.2e}" for k, v in bf16_ref.items()})
    o_eval = orc.mobilevit_v2_forward(sd, x, width_multiplier=wm, training=False)
    o_logits, o_loss, o_grads, o_running = orc.generic_train_step(orc.mobilevit_v2_forward, sd, x, y, width_multiplier=wm)
    checks = {"logits_eval": rel(o_eval, logits_eval), "logits_train": rel(o_logits, logits.detach()), "loss": abs(float(o_loss) - float(loss.detach())),
              "grad_worst_rel": max(rel(o_grads[k], ref_grads[k]) for k in ref_grads if ref_grads[k].norm() > 1e-6 * gmax),
              "bn_running_worst_rel": max(rel(v, ref_sd_after[k]) for k, v in o_running.items())}
    print(name, {k: f"{v:.2e}" for k, v in checks.items()})
    assert checks["logits_eval"] < 1e-5 and checks["logits_train"] < 1e-5 and checks["loss"] < 1e-5, checks
    assert checks["grad_worst_rel"] < 2e-4 and checks["bn_running_worst_rel"] < 1e-5, checks
    names = list(ref_grads.keys())
    out = {"logits_train": logits.detach().numpy(), "logits_eval": logits_eval.numpy(), "loss": np.float32(loss.item()),
           "grad_names": np.array(names), "grad_norm": np.array([ref_grads[k].norm().item() for k in names], dtype=np.float64),
           "oracle_vs_reference": np.array(json.dumps(checks)), "ref_bf16_autocast_err": np.array(json.dumps(bf16_ref))}
    for k in V2_FULL_GRADS:
        out["grad::" + k] = ref_grads[k].numpy()
    for k in ("conv_1.block.norm.running_mean", "layer_4.1.conv_proj.block.norm.running_var"):
        out["bn::" + k] = ref_sd_after[k].numpy()
    if batch >= 16:
        out.update(grad_sample_fields(ref_grads))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    with open(os.path.join(outdir, f"mobilevitv2_w{int(round(wm * 100)):03d}_keys.json"), "w") as f:
        json.dump({k: list(s) for k, s in shapes.items()}, f, indent=0)


CLIP_CASE = dict(name="clip_tiny_64_b8", vit_mode="tiny", res=64, batch=8, text_dim=64, text_layers=2, text_heads=2, vocab=200, ctx=16, proj=64)
CLIP_FULL_GRADS = ["logit_scale", "image_encoder.classifier.proj", "image_encoder.cls_token", "image_encoder.patch_emb.1.block.conv.weight",
                   "text_encoder.projection_layer", "text_encoder.embedding_layer.weight", "text_encoder.positional_embedding.pos_embed.pos_embed",
                   "text_encoder.transformer.0.pre_norm_mha.1.qkv_proj.weight", "text_encoder.transformer.1.pre_norm_ffn.4.bias",
                   "text_encoder.final_layer_norm.weight"]


def build_reference_clip(c):
    os.chdir(REF)
    import cvnets
    from options.utils import flatten_yaml_as_dict

    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/multi_modal_img_text/clip_vit.yaml")))
    for k, v in cfg.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    for k, v in {"dataset.category": "multi_modal_image_text", "dev.device": "cpu", "dataset.text_vocab_size": c["vocab"],
                 "dataset.text_context_length": c["ctx"], "dataset.padding_index": 0, "ddp.use_distributed": False, "ddp.rank": 0,
                 "model.classification.vit.mode": c["vit_mode"], "model.classification.vit.dropout": 0.0,
                 "model.classification.gradient_checkpointing": False, "model.text.transformer.gradient_checkpoint": False,
                 "model.text.transformer.model_dim": c["text_dim"], "model.text.transformer.n_transformer_layers": c["text_layers"],
                 "model.text.transformer.n_heads_per_layer": c["text_heads"],
                 "model.multi_modal_image_text.clip.projection_dim": c["proj"]}.items():
        setattr(opts, k, v)
    model = cvnets.get_model(opts)
    model.image_encoder.emb_dropout.p = 0.0
    sys.path.insert(0, REF)
    from loss_fn.multi_modal_img_text.contrastive_loss_clip import ContrastiveLossClip
    return model, ContrastiveLossClip(opts), opts


def run_clip_case(outdir):
    c = CLIP_CASE
    torch.manual_seed(0)
    model, loss_fn, _ = build_reference_clip(c)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    sd["logit_scale"] = torch.tensor(float(np.log(1.0 / 0.07)))
    sd["image_encoder.cls_token"] = 0.02 * seeded_state_dict({"cls_token_values": shapes["image_encoder.cls_token"]}, seed=0)["cls_token_values"]
    model.load_state_dict(sd, strict=True)
    x = seeded_input((c["batch"], 3, c["res"], c["res"]), seed=1)
    tok = clip_tokens(c["batch"], c["ctx"], c["vocab"], seed=1)
    model.train()
    loss_fn.train()

    def run():
        model.zero_grad()
        out = model({"image": x, "text": tok})
        img, txt = out["image"].detach().clone(), out["text"].detach().clone()
        loss = loss_fn(None, out)["total_loss"]
        loss.backward()
        return img, txt, loss.detach().clone(), {k: p.grad.detach().clone().float() for k, p in model.named_parameters()}

    img, txt, loss, ref_grads = run()
    ref_sd_after = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        img_b, txt_b, loss_b, gb = run()

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    gmax = max(v.norm().item() for v in ref_grads.values())
    bf16_ref = {"image": rel(img_b.float(), img), "text": rel(txt_b.float(), txt), "loss": abs(float(loss_b) - float(loss)),
                "grad_global": (sum(float((gb[k].double() - ref_grads[k].double()).pow(2).sum()) for k in gb) /
                                sum(float(ref_grads[k].double().pow(2).sum()) for k in gb)) ** 0.5,
                "grad_norm_worst": max(abs(gb[k].norm().item() - ref_grads[k].norm().item()) / (ref_grads[k].norm().item() + 1e-3 * gmax) for k in gb),
                "grad_full_worst": max(rel(gb[k], ref_grads[k]) for k in CLIP_FULL_GRADS)}
    print(c["name"], "reference bf16-autocast vs fp32:", {k: f"{v:.2e}" for k, v in bf16_ref.items()})
    kw = dict(vit_mode=c["vit_mode"], text_layers=c["text_layers"], text_heads=c["text_heads"])
    o_img, o_txt, o_loss, o_grads, o_running = orc.clip_train_step(sd, x, tok, **kw)
    checks = {"image": rel(o_img, img), "text": rel(o_txt, txt), "loss": abs(float(o_loss) - float(loss)),
              "grad_worst_rel": max(rel(o_grads[k], ref_grads[k]) for k in ref_grads if ref_grads[k].norm() > 1e-6 * gmax),
              "bn_running_worst_rel": max(rel(v, ref_sd_after[k]) for k, v in o_running.items())}
    print(c["name"], {k: f"{v:.2e}" for k, v in checks.items()})
    assert max(checks["image"], checks["text"], checks["loss"]) < 1e-5 and checks["grad_worst_rel"] < 2e-4, checks
    names = list(ref_grads.keys())
    out = {"image": img.numpy(), "text": txt.numpy(), "loss": np.float32(loss.item()), "tokens": tok.numpy(),
           "grad_names": np.array(names), "grad_norm": np.array([ref_grads[k].norm().item() for k in names], dtype=np.float64),
           "oracle_vs_reference": np.array(json.dumps(checks)), "ref_bf16_autocast_err": np.array(json.dumps(bf16_ref)),
           "config": np.array(json.dumps(c))}
    for k in CLIP_FULL_GRADS:
        out["grad::" + k] = ref_grads[k].numpy()
    np.savez_compressed(os.path.join(outdir, c["name"] + ".npz"), **out)
    with open(os.path.join(outdir, "clip_tiny_keys.json"), "w") as f:
        json.dump({k: list(s) for k, s in shapes.items()}, f, indent=0)


def mha_cases(outdir):
    """Pins oracle.multi_head_attention / transformer_encoder against the reference layer incl.
    masks (the only numerical cross-check the reference's own tests hold for this path:
    tests/test_multi_head_attn.py:29-121, atol=rtol=1e-3, 3 implementations)."""
    os.chdir(REF)
    from cvnets.layers import MultiHeadAttention

    out = {}
    for idx, (b, s, c, h, causal, kpm) in enumerate([(2, 16, 32, 4, False, False), (3, 21, 48, 4, True, False),
                                                     (2, 77, 64, 8, True, True), (4, 64, 80, 4, False, False)]):
        torch.manual_seed(idx)
        layer = MultiHeadAttention(c, h, attn_dropout=0.0, bias=True).eval()
        shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
        sd = seeded_state_dict({"mha." + k: s_ for k, s_ in shapes.items()}, seed=idx)
        layer.load_state_dict({k[4:]: v for k, v in sd.items()})
        x = seeded_input((b, s, c), seed=100 + idx)
        am = None
        if causal:
            am = torch.full((s, s), float("-inf")).triu(1).unsqueeze(0).expand(b, -1, -1).contiguous()
        pm = None
        if kpm:
            pm = torch.zeros(b, s)
            pm[:, -5:] = 1
        with torch.no_grad():
            ref = layer(x, attn_mask=am, key_padding_mask=pm)
            ref_pt = layer(x.transpose(0, 1), attn_mask=None if am is None else am[0],
                           key_padding_mask=None if pm is None else pm.bool(), use_pytorch_mha=True).transpose(0, 1)
        o = orc.multi_head_attention(sd, "mha", x, h, attn_mask=am, key_padding_mask=pm)
        err = float((o - ref).abs().max())
        err_pt = float((ref_pt - ref).abs().max())
        print(f"mha case {idx}: oracle-vs-ref {err:.2e}  ref_default-vs-ref_pytorch {err_pt:.2e}")
        assert err < 1e-5 and err_pt < 1e-3
        out[f"case{idx}_out"] = ref.numpy()
        out[f"case{idx}_cfg"] = np.array([b, s, c, h, int(causal), int(kpm)])
    np.savez_compressed(os.path.join(outdir, "mha_cases.npz"), **out)



def run_endpoints_case(outdir, name="mobilevit_s_os8_96_b2", mode="small", output_stride=8, batch=2, res=96):
    """the backbone as the segmentation / detection heads use it — BaseImageEncoder.extract_end_points_all on a
    MobileViT built with output_stride (layer_4 / layer_5 trade their stride for dilation, base_image_encoder.py:36-47, 206-259;
    mobilevit.py:225-256).  The reference class is constructed directly with the kwarg the segmentation builder passes."""
    torch.manual_seed(0)
    build_reference_model(mode)  # sets cwd / sys.path and registers everything
    import cvnets
    from cvnets.models.classification.mobilevit import MobileViT as RefMobileViT
    from options.utils import flatten_yaml_as_dict
    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/classification/imagenet/mobilevit.yaml")))
    for k, v in cfg.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    setattr(opts, "dataset.category", "classification")
    setattr(opts, "dev.device", "cpu")
    setattr(opts, "model.classification.mit.mode", mode)
    for k in ("model.classification.mit.dropout", "model.classification.mit.attn_dropout", "model.classification.mit.ffn_dropout",
              "model.classification.classifier_dropout"):
        setattr(opts, k, 0.0)
    model = RefMobileViT(opts, output_stride=output_stride)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    x = seeded_input((batch, 3, res, res), seed=1)
    model.train()
    ep = model.extract_end_points_all(x, use_l5=True, use_l5_exp=True)
    loss = sum(v.square().mean() for k, v in ep.items() if k.startswith("out_"))
    model.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    names = list(grads.keys())
    out = {"loss": np.float32(loss.item()), "grad_names": np.array(names),
           "grad_norm": np.array([grads[k].norm().item() for k in names], dtype=np.float64),
           "strides": np.array([model.layer_4[0].stride, model.layer_4[0].dilation, model.layer_5[0].stride, model.layer_5[0].dilation])}
    for k, v in ep.items():
        out["shape::" + k] = np.array(v.shape)
        out["stats::" + k] = np.array([v.mean().item(), v.std().item(), v.abs().max().item()], dtype=np.float64)
        if k in ("out_l4", "out_l5"):
            out["full::" + k] = v.detach().numpy()
    for k in ("layer_4.0.block.conv_3x3.block.conv.weight", "layer_5.0.block.conv_3x3.block.conv.weight", "layer_5.1.local_rep.conv_3x3.block.conv.weight",
              "conv_1.block.conv.weight"):
        out["grad::" + k] = grads[k].numpy()
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, {k: tuple(v.shape) for k, v in ep.items()}, "loss", float(loss))

def build_reference_segmentation(cfg="deeplabv3_mobilevit", width=None):
    """cvnets.get_model(opts) on config/segmentation/pascal_voc/<cfg>.yaml (sync_batch_norm -> batch_norm, dropouts 0)"""
    os.chdir(REF)
    import cvnets
    from options.utils import flatten_yaml_as_dict
    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfgd = flatten_yaml_as_dict(yaml.safe_load(open(f"config/segmentation/pascal_voc/{cfg}.yaml")))
    for k, v in cfgd.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    setattr(opts, "dataset.category", "segmentation")
    setattr(opts, "dev.device", "cpu")
    setattr(opts, "model.classification.pretrained", None)
    setattr(opts, "model.normalization.name", "batch_norm")
    if width is not None:
        setattr(opts, "model.classification.mitv2.width_multiplier", width)
    setattr(opts, "model.segmentation.pspnet.psp_dropout", 0.0)
    for k in ("model.classification.mit.dropout", "model.classification.mit.attn_dropout", "model.classification.mit.ffn_dropout",
              "model.classification.classifier_dropout", "model.segmentation.classifier_dropout", "model.segmentation.deeplabv3.aspp_dropout",
              "model.segmentation.aux_dropout"):
        setattr(opts, k, 0.0)
    import logging
    logging.disable(logging.CRITICAL)
    return cvnets.get_model(opts)


def run_segmentation_case(outdir, name="deeplabv3_mobilevit_s_96_b2", batch=2, res=96, cfg="deeplabv3_mobilevit", width=None, head="deeplabv3",
                          rates=(12, 24, 36), output_stride=8, keys_name="deeplabv3_mobilevit_s_keys.json"):
    """DeepLabv3 head on the MobileViT-S encoder, built by the reference's own builder from
    config/segmentation/pascal_voc/deeplabv3_mobilevit.yaml (output stride 8, ASPP 512 channels, rates 12/24/36, ReLU head, auxiliary
    head; sync_batch_norm -> batch_norm for the single-process CPU run, dropouts 0).  The head restatement oracle/seg_oracle.py must
    reproduce the reference bit for bit on the reference's own end points before the reference's outputs are written."""
    torch.manual_seed(0)
    from oracle import seg_oracle
    model = build_reference_segmentation(cfg, width)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    x = seeded_input((batch, 3, res, res), seed=1)
    g = torch.Generator().manual_seed(5)
    target = torch.randint(0, 21, (batch, res, res), generator=g)
    target[:, :4, :] = 255  # ignored border, as the VOC masks have
    model.eval()
    with torch.no_grad():
        mask_eval = model(x)
    model.train()
    ep = model.encoder.extract_end_points_all(x, use_l5=True, use_l5_exp=False)
    # pin the head restatement on the reference's own end points (fresh state: running statistics as loaded)
    ep_d = {k: v.detach() for k, v in ep.items() if v is not None}
    if head == "deeplabv3":
        o_mask, o_aux, o_bn = seg_oracle.deeplabv3_head(sd, "seg_head", ep_d, rates=rates, output_stride=output_stride)
    else:
        o_mask, o_aux, o_bn = seg_oracle.pspnet_head(sd, "seg_head", ep_d, pool_sizes=(1, 2, 3, 6), output_stride=output_stride)
    model.load_state_dict(sd, strict=True)
    mask, aux = model(x)
    loss = seg_oracle.seg_loss(mask, aux, target)
    assert float((o_mask - mask).abs().max()) == 0.0 and float((o_aux - aux).abs().max()) == 0.0, "head restatement differs from the reference"
    for k, v in o_bn.items():
        assert float((v - model.state_dict()[k]).abs().max()) == 0.0, k
    model.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    names = list(grads.keys())
    out = {"mask_eval": mask_eval.numpy(), "mask_train": mask.detach().numpy(), "aux_train": aux.detach().numpy(), "loss": np.float32(loss.item()),
           "target": target.numpy().astype(np.int16), "grad_names": np.array(names),
           "grad_norm": np.array([grads[k].norm().item() for k in names], dtype=np.float64)}
    for k in names:
        if k.startswith("seg_head") and grads[k].numel() <= 150000:
            out["grad::" + k] = grads[k].numpy()
    for k in ("encoder.conv_1.block.conv.weight", "encoder.layer_4.0.block.conv_3x3.block.conv.weight", "encoder.layer_5.1.conv_proj.block.conv.weight"):
        if k in grads:
            out["grad::" + k] = grads[k].numpy()
    for k, v in model.state_dict().items():
        if k.startswith("seg_head") and ("running_mean" in k or "running_var" in k):
            out["bn::" + k] = v.numpy()
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    json.dump({k: list(v) for k, v in shapes.items()}, open(os.path.join(outdir, keys_name), "w"))
    print(name, tuple(mask.shape), tuple(aux.shape), "loss", float(loss), "head restatement == reference")


def build_reference_ssd():
    """cvnets.get_model(opts) on config/detection/ssd_coco/mobilevit.yaml (sync_batch_norm -> batch_norm, dropouts 0, 81 classes)"""
    os.chdir(REF)
    import cvnets
    from options.utils import flatten_yaml_as_dict
    parser = cvnets.modeling_arguments(argparse.ArgumentParser())
    opts = parser.parse_args([])
    cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/detection/ssd_coco/mobilevit.yaml")))
    for k, v in cfg.items():
        if hasattr(opts, k):
            setattr(opts, k, v)
    setattr(opts, "dataset.category", "detection")
    setattr(opts, "dev.device", "cpu")
    setattr(opts, "model.classification.pretrained", None)
    setattr(opts, "model.normalization.name", "batch_norm")
    setattr(opts, "model.detection.n_classes", 81)
    for k in ("model.classification.mit.dropout", "model.classification.mit.attn_dropout", "model.classification.mit.ffn_dropout",
              "model.classification.classifier_dropout"):
        setattr(opts, k, 0.0)
    import logging
    logging.disable(logging.CRITICAL)
    return cvnets.get_model(opts)


def run_detection_case(outdir, name="ssd_mobilevit_s_160_b2", batch=2, res=160):
    """the SSD head on the MobileViT-S encoder built by the reference's own builder (training forward: scores, boxes).
    oracle/det_oracle.py must reproduce the reference bit for bit on the reference's own end points before the outputs are written."""
    torch.manual_seed(0)
    from oracle import det_oracle
    model = build_reference_ssd()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    x = seeded_input((batch, 3, res, res), seed=1)
    model.eval()
    with torch.no_grad():
        ev = model.ssd_forward(model.get_backbone_features(x), device="cpu")
    model.train()
    ep = model.encoder.extract_end_points_all(x)
    o_scores, o_boxes, o_bn = det_oracle.ssd_forward(sd, {k: v.detach() for k, v in ep.items() if v is not None}, list(model.output_strides), 81)
    model.load_state_dict(sd, strict=True)
    out = model(x)
    scores, boxes = out["scores"], out["boxes"]
    assert float((o_scores - scores).abs().max()) == 0.0 and float((o_boxes - boxes).abs().max()) == 0.0, "SSD restatement differs from the reference"
    for k, v in o_bn.items():
        assert float((v - model.state_dict()[k]).abs().max()) == 0.0, k
    g = torch.Generator().manual_seed(9)
    t_s, t_b = torch.randn(scores.shape, generator=g), torch.randn(boxes.shape, generator=g)
    loss = torch.nn.functional.mse_loss(scores, t_s) + torch.nn.functional.mse_loss(boxes, t_b)
    model.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    names = list(grads.keys())
    anchors = ev[2]
    res_ = {"scores_eval": ev[0].numpy().astype(np.float16), "boxes_eval": ev[1].numpy(), "scores_train": scores.detach().numpy().astype(np.float32),
            "boxes_train": boxes.detach().numpy(), "anchors": anchors.numpy(), "loss": np.float32(loss.item()), "grad_names": np.array(names),
            "grad_norm": np.array([grads[k].norm().item() for k in names], dtype=np.float64),
            "output_strides": np.array(list(model.output_strides))}
    for k in names:
        if not k.startswith("encoder") and grads[k].numel() <= 70000:
            res_["grad::" + k] = grads[k].numpy()
    for k in ("encoder.conv_1.block.conv.weight", "encoder.layer_5.1.conv_proj.block.conv.weight"):
        res_["grad::" + k] = grads[k].numpy()
    for k, v in model.state_dict().items():
        if not k.startswith("encoder") and ("running_mean" in k or "running_var" in k):
            res_["bn::" + k] = v.numpy()
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **res_)
    json.dump({k: list(v) for k, v in shapes.items()}, open(os.path.join(outdir, "ssd_mobilevit_s_keys.json"), "w"))
    print(name, tuple(scores.shape), tuple(boxes.shape), tuple(anchors.shape), "loss", float(loss), "SSD restatement == reference")


LARGE_CASES = [("mobilevit_s_256_b16", "small", 16, 256)]        # the BASELINE configuration at a batch where train-mode BatchNorm noise is small
# the largest batch the 62 GB authoring container holds through the reference's unfused fp32 graph (~0.4 GB of saved activations per image):
# one train step at the benchmark resolution for tests/test_bench_scale_gpu.py (`--b64`)
BENCH_SCALE_CASES = [("mobilevit_s_256_b64", "small", 64, 256)]
LARGE_VIT_CASES = [("vit_tiny_224_b16", "tiny", 16, 224)]
LARGE_V2_CASES = [("mobilevitv2_w100_256_b16", 1.0, 16, 256)]

if __name__ == "__main__":
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    torch.set_num_threads(8)
    if "--dropout" in sys.argv:
        run_dropout_case(outdir)
        sys.exit(0)
    if "--detection" in sys.argv:
        run_detection_case(outdir)
        sys.exit(0)
    if "--segmentation" in sys.argv:
        run_segmentation_case(outdir)
        run_segmentation_case(outdir, name="deeplabv3_mobilevitv2_w050_96_b2", cfg="deeplabv3_mobilevitv2", width=0.5, head="deeplabv3", rates=(6, 12, 18),
                              output_stride=16, keys_name="deeplabv3_mobilevitv2_w050_keys.json")
        run_segmentation_case(outdir, name="pspnet_mobilevitv2_w050_96_b2", cfg="pspnet_mobilevitv2", width=0.5, head="pspnet", output_stride=16,
                              keys_name="pspnet_mobilevitv2_w050_keys.json")
        sys.exit(0)
    if "--endpoints" in sys.argv:
        run_endpoints_case(outdir)
        run_endpoints_case(outdir, name="mobilevit_xxs_os16_64_b2", mode="xx_small", output_stride=16, batch=2, res=64)
        sys.exit(0)
    if "--b64" in sys.argv:
        for c in BENCH_SCALE_CASES:
            run_case(*c, outdir)
        sys.exit(0)
    if "--large" in sys.argv:  # large-batch bf16 parity fixtures only (tests/test_bf16_parity_gpu.py)
        for c in LARGE_CASES:
            run_case(*c, outdir)
        for c in LARGE_VIT_CASES:
            run_vit_case(*c, outdir)
        for c in LARGE_V2_CASES:
            run_v2_case(*c, outdir)
        sys.exit(0)
    mha_cases(outdir)
    for c in CASES:
        run_case(*c, outdir)
    for c in VIT_CASES:
        run_vit_case(*c, outdir)
    for c in V2_CASES:
        run_v2_case(*c, outdir)
    run_clip_case(outdir)
    print("golden fixtures written to", outdir)
