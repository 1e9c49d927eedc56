"""CPU test of the drop-in boundary: a model built by the REFERENCE's own builder can be class-swapped onto
the HIP mirrors with an identical module tree / state_dict; the mirrors then refuse to run without a GPU (no silent CPU path).
Needs /root/reference (authoring container); skipped where the reference tree is absent (GPU box)."""
import argparse
import json
import os
import sys

import pytest
import torch

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this machine")


@pytest.fixture(scope="module")
def ref_model():
    sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import yaml
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
        yield cvnets.get_model(opts), opts
    finally:
        os.chdir(cwd)


def test_class_swap_keeps_tree_and_state_dict(ref_model):
    import cvnets_amd
    from cvnets_amd import dropin

    model, opts = ref_model
    before = {k: (tuple(v.shape), v.dtype) for k, v in model.state_dict().items()}
    names_before = [n for n, _ in model.named_modules()]
    counts, left = dropin.swap_to_hip(model, strict=True)
    assert left == []
    assert counts["MobileViTBlock"] == 3 and counts["InvertedResidual"] == 7 and counts["TransformerEncoder"] == 9
    assert counts["ConvLayer2d"] == 35 and counts["MultiHeadAttention"] == 9 and counts["MobileViT"] == 1
    assert {k: (tuple(v.shape), v.dtype) for k, v in model.state_dict().items()} == before
    assert [n for n, _ in model.named_modules()] == names_before
    gold = json.load(open(os.path.join(REPO, "tests", "golden", "mobilevit_small_keys.json")))
    assert {k: list(s) for k, (s, _) in before.items()} == gold
    # our own builder produces the same tree from the same opts
    ours = cvnets_amd.MobileViT(opts)
    assert [n for n, _ in ours.named_modules()] == names_before
    assert list(ours.state_dict().keys()) == list(before.keys())
    # and the swapped model has no CPU path
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 64, 64))


def test_reference_attribute_contract(ref_model):
    """attributes other reference code reaches into exist on the mirrors."""
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    m = cvnets_amd.MobileViT(default_opts())
    blk = m.layer_3[1]
    for attr in ("local_rep", "global_rep", "conv_proj", "fusion", "patch_h", "patch_w", "cnn_in_dim", "cnn_out_dim", "n_heads", "ffn_dim", "n_blocks"):
        assert hasattr(blk, attr), attr
    enc = blk.global_rep[0]
    assert enc.pre_norm_mha[1].qkv_proj.weight.shape == (3 * 144, 144)
    assert enc.pre_norm_ffn[1].weight.shape == (288, 144) and enc.pre_norm_ffn[4].weight.shape == (144, 288)
    conv = m.conv_1
    for attr in ("block", "in_channels", "out_channels", "stride", "groups", "kernel_size", "bias", "dilation"):
        assert hasattr(conv, attr), attr
    assert [n for n, _ in conv.block.named_children()] == ["conv", "norm", "act"]


def test_vit_class_swap(ref_model):
    """the reference's ViT builder -> class swap -> same tree / state_dict as cvnets_amd.build_vit."""
    import yaml
    import cvnets
    import cvnets_amd
    from cvnets_amd import dropin
    from options.utils import flatten_yaml_as_dict

    cwd = os.getcwd()
    os.chdir(REF)
    try:
        parser = cvnets.modeling_arguments(argparse.ArgumentParser())
        opts = parser.parse_args([])
        cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/classification/imagenet/vit.yaml")))
        for k, v in cfg.items():
            if hasattr(opts, k):
                setattr(opts, k, v)
        setattr(opts, "dataset.category", "classification")
        setattr(opts, "dev.device", "cpu")
        setattr(opts, "model.classification.gradient_checkpointing", False)
        model = cvnets.get_model(opts)
    finally:
        os.chdir(cwd)
    names_before = [n for n, _ in model.named_modules()]
    keys_before = list(model.state_dict().keys())
    counts, left = dropin.swap_to_hip(model, strict=True)
    assert left == []
    assert counts["VisionTransformer"] == 1 and counts["TransformerEncoder"] == 12 and counts["ConvLayer2d"] == 3
    ours = cvnets_amd.VisionTransformer(opts)
    assert [n for n, _ in ours.named_modules()] == names_before
    assert list(ours.state_dict().keys()) == keys_before == list(json.load(open(os.path.join(REPO, "tests", "golden", "vit_tiny_keys.json"))).keys())
    assert ours.emb_dropout.p == model.emb_dropout.p
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 64, 64))


def test_mobilevit_v2_class_swap(ref_model):
    """the reference's MobileViTv2 builder (mobilevit_v2.yaml, width 2.0) -> class swap -> same tree as cvnets_amd.MobileViTv2."""
    import yaml
    import cvnets
    import cvnets_amd
    from cvnets_amd import dropin
    from options.utils import flatten_yaml_as_dict

    cwd = os.getcwd()
    os.chdir(REF)
    try:
        parser = cvnets.modeling_arguments(argparse.ArgumentParser())
        opts = parser.parse_args([])
        cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/classification/imagenet/mobilevit_v2.yaml")))
        for k, v in cfg.items():
            if hasattr(opts, k):
                setattr(opts, k, v)
        setattr(opts, "dataset.category", "classification")
        setattr(opts, "dev.device", "cpu")
        model = cvnets.get_model(opts)
    finally:
        os.chdir(cwd)
    names_before = [n for n, _ in model.named_modules()]
    keys_before = list(model.state_dict().keys())
    counts, left = dropin.swap_to_hip(model, strict=True)
    assert left == []
    assert counts["MobileViTv2"] == 1 and counts["MobileViTBlockv2"] == 3 and counts["LinearAttnFFN"] == 9
    assert counts["LinearSelfAttention"] == 9 and counts["LayerNorm2D_NCHW"] == 21
    ours = cvnets_amd.MobileViTv2(opts)
    assert [n for n, _ in ours.named_modules()] == names_before
    assert list(ours.state_dict().keys()) == keys_before
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in model.state_dict().items()}
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 64, 64))


def test_clip_class_swap(ref_model):
    """the reference's CLIP builder (clip_vit.yaml, ViT-B/16 + 12-layer text transformer) -> class swap -> same tree as cvnets_amd.CLIP."""
    import yaml
    import cvnets
    import cvnets_amd
    from cvnets_amd import dropin
    from options.utils import flatten_yaml_as_dict

    cwd = os.getcwd()
    os.chdir(REF)
    try:
        parser = cvnets.modeling_arguments(argparse.ArgumentParser())
        opts = parser.parse_args([])
        cfg = flatten_yaml_as_dict(yaml.safe_load(open("config/multi_modal_img_text/clip_vit.yaml")))
        for k, v in cfg.items():
            if hasattr(opts, k):
                setattr(opts, k, v)
        for k, v in {"dataset.category": "multi_modal_image_text", "dev.device": "cpu", "dataset.text_vocab_size": 49408,
                     "dataset.text_context_length": 77, "dataset.padding_index": 0, "ddp.use_distributed": False, "ddp.rank": 0,
                     "model.classification.vit.mode": "tiny", "model.text.transformer.n_transformer_layers": 2}.items():
            setattr(opts, k, v)
        model = cvnets.get_model(opts)
    finally:
        os.chdir(cwd)
    names_before = [n for n, _ in model.named_modules()]
    keys_before = list(model.state_dict().keys())
    counts, left = dropin.swap_to_hip(model, strict=True)
    assert left == []
    assert counts["CLIP"] == 1 and counts["TextTransformer"] == 1 and counts["SimpleImageProjectionHead"] == 1 and counts["Embedding"] == 1
    ours = cvnets_amd.CLIP.build_model(opts)
    assert [n for n, _ in ours.named_modules()] == names_before
    assert list(ours.state_dict().keys()) == keys_before
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in model.state_dict().items()}
    with pytest.raises(RuntimeError):
        model({"image": torch.zeros(2, 3, 64, 64), "text": torch.ones(2, 77, dtype=torch.long)})


def test_vbs_shape_schedule_matches_reference():
    """cvnets_amd.schedule vs the reference's data/sampler/utils.py (loaded as a plain module: the `data` package itself needs
    torchvision) for the two shipped recipes, and the per-step draw vs random.seed(epoch) + random.choice."""
    import importlib.util
    import random

    from cvnets_amd import schedule

    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_sampler_utils", os.path.join(REF, "data", "sampler", "utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for args in (dict(crop_size_w=256, crop_size_h=256, batch_size_gpu0=128, max_scales=5, check_scale_div_factor=32, min_crop_size_w=160,
                      max_crop_size_w=320, min_crop_size_h=160, max_crop_size_h=320),      # mobilevit.yaml
                 dict(crop_size_w=384, crop_size_h=384, batch_size_gpu0=128, max_scales=5, check_scale_div_factor=32, min_crop_size_w=256,
                      max_crop_size_w=512, min_crop_size_h=256, max_crop_size_h=512),      # BASELINE config 5
                 dict(crop_size_w=224, crop_size_h=224, batch_size_gpu0=64, max_scales=7, check_scale_div_factor=16, min_crop_size_w=128,
                      max_crop_size_w=320, min_crop_size_h=128, max_crop_size_h=320)):
        assert schedule.image_batch_pairs(**args) == ref.image_batch_pairs(**args), args
    pairs = schedule.image_batch_pairs(384, 384, 128, 5, 32, 256, 512, 256, 512)
    assert pairs == [(256, 256, 288), (320, 320, 184), (384, 384, 128), (448, 448, 94), (512, 512, 72)]
    random.seed(3)
    expect = [random.choice(pairs) for _ in range(25)]
    assert schedule.vbs_sequence(pairs, 25, epoch=3) == expect


def test_segmentation_models_mirror_the_reference_key_sets():
    """cvnets_amd.build_segmentation constructs the module trees of the reference's SegEncoderDecoder + DeeplabV3 / PSPNet on MobileViT /
    MobileViTv2 encoders (key sets and shapes recorded from the reference builder by oracle/make_golden.py --segmentation); runs without
    a GPU (construction only)."""
    import json
    import os

    import cvnets_amd
    from cvnets_amd.layers import default_opts

    gold = os.path.join(os.path.dirname(__file__), "golden")
    common = {"model.segmentation.n_classes": 21, "model.segmentation.use_aux_head": True, "model.segmentation.use_level5_exp": False,
              "model.segmentation.deeplabv3.aspp_out_channels": 512}
    cases = [("mobilevit", "deeplabv3", {"model.classification.mit.mode": "small", "model.segmentation.output_stride": 8,
                                         "model.segmentation.deeplabv3.aspp_rates": (12, 24, 36)}, "deeplabv3_mobilevit_s_keys.json"),
             ("mobilevit_v2", "deeplabv3", {"model.classification.mitv2.width_multiplier": 0.5, "model.segmentation.output_stride": 16,
                                            "model.segmentation.deeplabv3.aspp_rates": (6, 12, 18)}, "deeplabv3_mobilevitv2_w050_keys.json"),
             ("mobilevit_v2", "pspnet", {"model.classification.mitv2.width_multiplier": 0.5, "model.segmentation.output_stride": 16},
              "pspnet_mobilevitv2_w050_keys.json")]
    for enc, head, over, keys in cases:
        shapes = json.load(open(os.path.join(gold, keys)))
        model = cvnets_amd.build_segmentation(default_opts(**{**common, **over}), enc, head)
        assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes, (enc, head)
        assert model.encoder.classifier is None and model.encoder.conv_1x1_exp is None
    assert type(model.seg_head.psp_layer.fusion[0].block.act).__name__ == "ReLU" and type(model.encoder.conv_1.block.act).__name__ == "Swish"


def test_ssd_model_mirrors_the_reference_key_set_and_anchors():
    """cvnets_amd.build_ssd constructs the module tree of the reference's SingleShotMaskDetector on MobileViT-S (394 state_dict keys recorded
    from the reference builder) and its anchor generator reproduces the reference's anchors for a 160 x 160 input exactly (host arithmetic)."""
    import json
    import os

    import numpy as np
    import torch

    import cvnets_amd
    from cvnets_amd.layers import default_opts

    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    shapes = json.load(open(os.path.join(gold_dir, "ssd_mobilevit_s_keys.json")))
    opts = default_opts(**{"model.classification.mit.mode": "small", "model.detection.n_classes": 81,
                           "anchor_generator.ssd.output_strides": [16, 32, 64, 128, 256, -1], "anchor_generator.ssd.aspect_ratios": [[2, 3]] * 5 + [[2]],
                           "anchor_generator.ssd.min_scale_ratio": 0.1, "anchor_generator.ssd.max_scale_ratio": 1.05,
                           "model.detection.ssd.proj_channels": [512, 256, 256, 128, 128, 64]})
    model = cvnets_amd.build_ssd(opts)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes
    gold = np.load(os.path.join(gold_dir, "ssd_mobilevit_s_160_b2.npz"))
    sizes = {16: 10, 32: 5, 64: 3, 128: 2, 256: 1, -1: 1}
    anchors = torch.cat([model.anchor_box_generator(sizes[o], sizes[o], o) for o in model.output_strides], 0).unsqueeze(0)
    assert torch.equal(anchors, torch.from_numpy(gold["anchors"]))
    assert model.anchor_box_generator.num_anchors_per_os() == [6, 6, 6, 6, 6, 4]


def test_ssd_on_mobilevitv2_mirrors_the_reference_key_set():
    """config/detection/ssd_coco/mobilevit_v2.yaml (width 0.75 here): build_ssd on the MobileViTv2 encoder has the reference builder's 358 keys / shapes"""
    import json
    import os

    import cvnets_amd
    from cvnets_amd.layers import default_opts

    shapes = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ssd_mobilevitv2_w075_keys.json")))
    opts = default_opts(**{"model.classification.mitv2.width_multiplier": 0.75, "model.detection.n_classes": 81,
                           "anchor_generator.ssd.output_strides": [16, 32, 64, 128, 256, -1], "anchor_generator.ssd.aspect_ratios": [[2, 3]] * 5 + [[2]],
                           "anchor_generator.ssd.min_scale_ratio": 0.1, "anchor_generator.ssd.max_scale_ratio": 1.05,
                           "model.detection.ssd.proj_channels": [512, 256, 256, 128, 128, 64]})
    model = cvnets_amd.build_ssd(opts, "mobilevit_v2")
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes
