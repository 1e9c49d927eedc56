"""MobileViT backbone + classifier assembled from the HIP-backed layers, with the reference's module tree and
state_dict keys (cvnets/models/classification/mobilevit.py:19-300, config/mobilevit.py:11-208,
base_image_encoder.py:261-301).  ``forward`` takes the reference's input ([B,3,H,W] float32 NCHW) and returns
logits [B, n_classes] in the compute dtype."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .layers import ConvLayer2d, Dropout, GlobalPool, LinearLayer, opt
from .modules import InvertedResidual, MobileViTBlock, MobileViTBlockv2, make_divisible


def get_configuration(opts) -> Dict:
    """cvnets/models/classification/config/mobilevit.py:11-208."""
    mode = opt(opts, "model.classification.mit.mode", "small")
    head_dim = opt(opts, "model.classification.mit.head_dim", None)
    num_heads = opt(opts, "model.classification.mit.number_heads", 4)
    if head_dim is not None and num_heads is not None:
        raise ValueError("--model.classification.mit.head-dim and --model.classification.mit.number-heads are mutually exclusive.")
    mode = mode.lower()
    table = {
        "xx_small": (2, (16, 24, 48, 64, 80), (64, 80, 96)),
        "x_small": (4, (32, 48, 64, 80, 96), (96, 120, 144)),
        "small": (4, (32, 64, 96, 128, 160), (144, 192, 240)),
    }
    if mode not in table:
        raise NotImplementedError(mode)
    exp, ch, td = table[mode]
    cfg = {
        "layer1": {"out_channels": ch[0], "expand_ratio": exp, "num_blocks": 1, "stride": 1, "block_type": "mv2"},
        "layer2": {"out_channels": ch[1], "expand_ratio": exp, "num_blocks": 3, "stride": 2, "block_type": "mv2"},
        "last_layer_exp_factor": 4,
    }
    for i, (name, nblk) in enumerate((("layer3", 2), ("layer4", 4), ("layer5", 3))):
        cfg[name] = {"out_channels": ch[2 + i], "transformer_channels": td[i], "ffn_dim": 2 * td[i], "transformer_blocks": nblk,
                     "patch_h": 2, "patch_w": 2, "stride": 2, "mv_expand_ratio": exp, "head_dim": head_dim, "num_heads": num_heads,
                     "block_type": "mobilevit"}
    return cfg


def _init_encoder_common(self, opts, kwargs) -> None:
    """BaseImageEncoder.__init__ (base_image_encoder.py:36-47): dilation bookkeeping for dense-prediction heads + checkpoint flag"""
    self.dilation = 1
    output_stride = kwargs.get("output_stride", None)
    self.dilate_l4 = self.dilate_l5 = False
    if output_stride == 8:
        self.dilate_l4 = self.dilate_l5 = True
    elif output_stride == 16:
        self.dilate_l5 = True
    self.output_stride = output_stride
    self.model_conf_dict = dict()
    self.neural_augmentor = None
    self.gradient_checkpointing = opt(opts, "model.classification.gradient_checkpointing", False)


def _forward_layer(self, layer, x):
    """base_image_encoder.py:196-204"""
    if self.training and getattr(self, "gradient_checkpointing", False):
        return ops.checkpoint(layer, x)
    return layer(x)


def _encoder_prologue(self, x):
    ops.pack_all(self)  # every conv / linear weight packed by one launch for this forward (train AND eval: the fused optimizer / EMA
    #                     kernels rewrite parameters through raw pointers, a cached pack could be stale)
    if self.training:
        ops.advance_dropout_seed(x.device)
        ops.bump_bn_counters(self)
    if x.dim() == 4 and x.shape[1] == 3 and not ops.is_nhwc(x):
        return x  # the raw NCHW image batch goes to conv_1 as it is: the stem kernels read the planes directly (layers._conv_forward)
    return ops.to_nhwc(x)


def _extract_end_points_all(self, x: Tensor, use_l5: Optional[bool] = True, use_l5_exp: Optional[bool] = False, *args, **kwargs) -> Dict[str, Tensor]:
    """base_image_encoder.py:206-254: the feature maps the segmentation / detection heads consume (out_l1 .. out_l5, out_l5_exp)"""
    if getattr(self, "neural_augmentor", None) is not None and self.training:
        raise NotImplementedError("neural augmentation is not on the HIP hot path")
    out_dict = {}
    x = _encoder_prologue(self, x)
    try:
        x = _forward_layer(self, self.conv_1, x)
        x = _forward_layer(self, self.layer_1, x)
        out_dict["out_l1"] = x
        x = _forward_layer(self, self.layer_2, x)
        out_dict["out_l2"] = x
        x = _forward_layer(self, self.layer_3, x)
        out_dict["out_l3"] = x
        x = _forward_layer(self, self.layer_4, x)
        out_dict["out_l4"] = x
        if use_l5:
            x = _forward_layer(self, self.layer_5, x)
            out_dict["out_l5"] = x
            if use_l5_exp:
                x = _forward_layer(self, self.conv_1x1_exp, x)
                out_dict["out_l5_exp"] = x
        return out_dict
    finally:
        ops.end_bn_counters()


def _extract_end_points_l4(self, x: Tensor, *args, **kwargs) -> Dict[str, Tensor]:
    """base_image_encoder.py:256-259"""
    return _extract_end_points_all(self, x, use_l5=False)


def _extract_features(self, x: Tensor, *args, **kwargs) -> Tensor:
    """base_image_encoder.py:261-275"""
    x = _encoder_prologue(self, x)
    try:
        for name in ("conv_1", "layer_1", "layer_2", "layer_3", "layer_4", "layer_5", "conv_1x1_exp"):
            x = _forward_layer(self, getattr(self, name), x)
        return x
    finally:
        ops.end_bn_counters()


class MobileViT(nn.Module):
    def __init__(self, opts, *args, **kwargs) -> None:
        super().__init__()
        num_classes = opt(opts, "model.classification.n_classes", 1000)
        classifier_dropout = opt(opts, "model.classification.classifier_dropout", 0.0)
        pool_type = opt(opts, "model.layer.global_pool", "mean")
        cfg = get_configuration(opts)
        _init_encoder_common(self, opts, kwargs)
        self.conv_1 = ConvLayer2d(opts=opts, in_channels=3, out_channels=16, kernel_size=3, stride=2, use_norm=True, use_act=True)
        self.model_conf_dict["conv1"] = {"in": 3, "out": 16}
        in_channels = 16
        for idx in range(1, 6):
            layer, out_channels = self._make_layer(opts=opts, input_channel=in_channels, cfg=cfg[f"layer{idx}"],
                                                   dilate=(idx == 4 and self.dilate_l4) or (idx == 5 and self.dilate_l5))
            setattr(self, f"layer_{idx}", layer)
            self.model_conf_dict[f"layer{idx}"] = {"in": in_channels, "out": out_channels}
            in_channels = out_channels
        exp_channels = min(cfg["last_layer_exp_factor"] * in_channels, 960)
        self.conv_1x1_exp = ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=exp_channels, kernel_size=1, stride=1,
                                        use_act=True, use_norm=True)
        self.model_conf_dict["exp_before_cls"] = {"in": in_channels, "out": exp_channels}
        self.classifier = nn.Sequential()
        self.classifier.add_module(name="global_pool", module=GlobalPool(pool_type=pool_type, keep_dim=False))
        if 0.0 < classifier_dropout < 1.0:
            self.classifier.add_module(name="dropout", module=Dropout(p=classifier_dropout, inplace=True))
        self.classifier.add_module(name="fc", module=LinearLayer(in_features=exp_channels, out_features=num_classes, bias=True))
        self.n_classes = num_classes

    def _make_layer(self, opts, input_channel, cfg: Dict, dilate: bool = False) -> Tuple[nn.Sequential, int]:
        if cfg.get("block_type", "mobilevit").lower() == "mobilevit":
            return self._make_mit_layer(opts, input_channel, cfg, dilate=dilate)
        return self._make_mobilenet_layer(opts, input_channel, cfg)

    @staticmethod
    def _make_mobilenet_layer(opts, input_channel: int, cfg: Dict) -> Tuple[nn.Sequential, int]:
        output_channels = cfg.get("out_channels")
        block = []
        for i in range(cfg.get("num_blocks", 2)):
            stride = cfg.get("stride", 1) if i == 0 else 1
            block.append(InvertedResidual(opts=opts, in_channels=input_channel, out_channels=output_channels, stride=stride,
                                          expand_ratio=cfg.get("expand_ratio", 4)))
            input_channel = output_channels
        return nn.Sequential(*block), input_channel

    def _make_mit_layer(self, opts, input_channel, cfg: Dict, dilate: bool = False) -> Tuple[nn.Sequential, int]:
        # mobilevit.py:225-256: segmentation backbones trade the stride of layer_4 / layer_5 for dilation (output_stride 16 / 8)
        prev_dilation = self.dilation
        block = []
        stride = cfg.get("stride", 1)
        if stride == 2:
            if dilate:
                self.dilation *= 2
                stride = 1
            block.append(InvertedResidual(opts=opts, in_channels=input_channel, out_channels=cfg.get("out_channels"), stride=stride,
                                          expand_ratio=cfg.get("mv_expand_ratio", 4), dilation=prev_dilation))
            input_channel = cfg.get("out_channels")
        head_dim = cfg.get("head_dim", 32)
        transformer_dim = cfg["transformer_channels"]
        if head_dim is None:
            num_heads = cfg.get("num_heads", 4) or 4
            head_dim = transformer_dim // num_heads
        if transformer_dim % head_dim != 0:
            raise ValueError(f"Transformer input dimension should be divisible by head dimension. Got {transformer_dim} and {head_dim}.")
        block.append(MobileViTBlock(
            opts=opts, in_channels=input_channel, transformer_dim=transformer_dim, ffn_dim=cfg.get("ffn_dim"),
            n_transformer_blocks=cfg.get("transformer_blocks", 1), patch_h=cfg.get("patch_h", 2), patch_w=cfg.get("patch_w", 2),
            dropout=opt(opts, "model.classification.mit.dropout", 0.1), ffn_dropout=opt(opts, "model.classification.mit.ffn_dropout", 0.0),
            attn_dropout=opt(opts, "model.classification.mit.attn_dropout", 0.1), head_dim=head_dim,
            no_fusion=opt(opts, "model.classification.mit.no_fuse_local_global_features", False),
            conv_ksize=opt(opts, "model.classification.mit.conv_kernel_size", 3)))
        return nn.Sequential(*block), input_channel

    # base_image_encoder.py:196-283
    extract_end_points_all = _extract_end_points_all
    extract_end_points_l4 = _extract_end_points_l4
    extract_features = _extract_features

    def forward_classifier(self, x: Tensor, *args, **kwargs) -> Tensor:
        x = self.extract_features(x)
        return self.classifier(x)

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        return self.forward_classifier(x, *args, **kwargs)


def build_mobilevit(mode: str = "small", opts=None, **overrides) -> MobileViT:
    from .layers import default_opts

    if opts is None:
        opts = default_opts(**{"model.classification.mit.mode": mode}, **overrides)
    return MobileViT(opts)


# =============================================================================================
# ViT  (cvnets/models/classification/vit.py:33-649, config/vit.py:12-99)
# =============================================================================================
def get_vit_configuration(opts) -> Dict:
    mode = (opt(opts, "model.classification.vit.mode", "tiny") or "tiny").lower()
    dropout = opt(opts, "model.classification.vit.dropout", 0.0)
    norm_layer = opt(opts, "model.classification.vit.norm_layer", "layer_norm")
    table = {"tiny": (192, 12, 3, 0.1), "small": (384, 12, 6, 0.0), "base": (768, 12, 12, 0.0), "large": (1024, 24, 16, 0.0),
             "huge": (1280, 32, 20, 0.0)}
    if mode not in table:
        raise NotImplementedError(f"Got unsupported ViT configuration: {mode}")
    e, n, h, pdrop = table[mode]
    return {"embed_dim": e, "n_transformer_layers": n, "n_attn_heads": h, "ffn_dim": 4 * e, "norm_layer": norm_layer,
            "pos_emb_drop_p": pdrop, "attn_dropout": 0.0, "ffn_dropout": 0.0, "dropout": dropout}


class VisionTransformer(nn.Module):
    """Conv-stem ViT of the reference: 3 patch-embedding convs (k4s4 / k2s2 / k2s2) -> + learnable positional embedding, class token
    (no positional embedding on it) -> N x TransformerEncoder -> LayerNorm(eps 1e-6) -> class-token classifier.
    Gradient checkpointing (vit.yaml:81) is a memory-saving re-execution of the same kernels and is not needed with 288 GB of HBM."""

    def __init__(self, opts, *args, **kwargs) -> None:
        super().__init__()
        from .layers import PositionalEmbedding, get_normalization_layer
        from .modules import TransformerEncoder

        num_classes = opt(opts, "model.classification.n_classes", 1000)
        if opt(opts, "model.classification.vit.use_pytorch_mha", False):
            raise NotImplementedError("use_pytorch_mha is not on the HIP hot path")
        cfg = get_vit_configuration(opts)
        embed_dim, norm_layer = cfg["embed_dim"], cfg["norm_layer"]
        num_embeddings = (224 // 16) ** 2
        stem_dim = max(32, embed_dim // 4)
        self.patch_emb = nn.Sequential(
            ConvLayer2d(opts=opts, in_channels=3, out_channels=stem_dim, kernel_size=4, stride=4, bias=False, use_norm=True, use_act=True),
            ConvLayer2d(opts=opts, in_channels=stem_dim, out_channels=stem_dim, kernel_size=2, stride=2, bias=False, use_norm=True, use_act=True),
            ConvLayer2d(opts=opts, in_channels=stem_dim, out_channels=embed_dim, kernel_size=2, stride=2, bias=True, use_norm=False, use_act=False),
        )
        n_layers = cfg["n_transformer_layers"]
        sd_max = opt(opts, "model.classification.vit.stochastic_dropout", 0.0)
        # vit.py:126-132: the drop rate grows linearly with depth (np.linspace, rounded to 3 decimals)
        sd_rates = [round(sd_max * i / max(n_layers - 1, 1), 3) for i in range(n_layers)]
        blocks = [TransformerEncoder(opts=opts, embed_dim=embed_dim, ffn_latent_dim=cfg["ffn_dim"], num_heads=cfg["n_attn_heads"],
                                     attn_dropout=cfg["attn_dropout"], dropout=cfg["dropout"], ffn_dropout=cfg["ffn_dropout"],
                                     transformer_norm_layer=norm_layer, stochastic_dropout=sd_rates[i])
                  for i in range(n_layers)]
        # base_image_encoder.py:36-47 / vit.py:150-165: activation checkpointing over `checkpoint_segments` chunks of the encoder stack
        self.gradient_checkpointing = opt(opts, "model.classification.gradient_checkpointing", False)
        self.checkpoint_segments = opt(opts, "model.classification.vit.checkpoint_segments", 4)
        self.post_transformer_norm = get_normalization_layer(opts=opts, num_features=embed_dim, norm_type=norm_layer)
        self.transformer = nn.Sequential(*blocks)
        self.classifier = LinearLayer(embed_dim, num_classes)
        if not opt(opts, "model.classification.vit.no_cls_token", False):
            self.cls_token = nn.Parameter(torch.zeros(size=(1, 1, embed_dim)))
            torch.nn.init.trunc_normal_(self.cls_token, std=0.02)
        else:
            self.cls_token = None
        self.pos_embed = PositionalEmbedding(opts=opts, num_embeddings=num_embeddings, embedding_dim=embed_dim, sequence_first=False,
                                             padding_idx=None,
                                             is_learnable=not opt(opts, "model.classification.vit.sinusoidal_pos_emb", False),
                                             interpolation_mode="bilinear")
        self.emb_dropout = Dropout(p=cfg["pos_emb_drop_p"])
        self.embed_dim = embed_dim
        self.n_classes = num_classes
        for m in self.modules():  # vit.py:204-208 update_layer_norm_eps
            if isinstance(m, nn.LayerNorm):
                m.eps = 1e-6

    def extract_patch_embeddings(self, x: Tensor):
        ops.pack_all(self)  # train AND eval (see MobileViT.extract_features)
        if self.training:
            ops.advance_dropout_seed(x.device)
        fm = self.patch_emb(ops.to_nhwc(x))
        B, E, n_h, n_w = fm.shape
        N = n_h * n_w
        pos = self.pos_embed.table(N)
        cls = self.cls_token.view(E) if self.cls_token is not None else None
        t = ops.VitEmbed.apply(ops.tokens_of(fm), pos, cls, B)
        t = ops.dropout(t, self.emb_dropout.p, self.training)
        return t, B, N + (1 if cls is not None else 0), (n_h, n_w)

    def extract_features(self, x: Tensor, *args, **kwargs):
        if kwargs.get("return_image_embeddings", False):
            raise NotImplementedError("return_image_embeddings (dense-prediction heads) is not on the HIP hot path")
        t, B, S, _ = self.extract_patch_embeddings(x)
        seqmap = (B, S, 1, 1, S, 1, S)
        layers = list(self.transformer)
        if self.training and getattr(self, "gradient_checkpointing", False):
            # vit.py:525-533 (checkpoint_sequential): the stack is cut into `checkpoint_segments` chunks; only chunk inputs are kept
            nseg = max(1, min(int(getattr(self, "checkpoint_segments", 4)), len(layers)))
            per = (len(layers) + nseg - 1) // nseg
            for i in range(0, len(layers), per):
                chunk = layers[i:i + per]

                def run(tt, chunk=chunk):
                    for layer in chunk:
                        tt = layer.forward_tokens(tt, seqmap)
                    return tt
                t = ops.checkpoint(run, t)
        else:
            for layer in layers:
                t = layer.forward_tokens(t, seqmap)
        n = self.post_transformer_norm
        t = ops.layer_norm_tokens(t, n, seqmap)
        if self.cls_token is None:
            raise NotImplementedError("mean-pooled ViT (no_cls_token) is not on the HIP hot path")
        return ops.RowsGather.apply(t, B, S, 0), None

    def forward_classifier(self, x: Tensor, *args, **kwargs):
        cls_embedding, image_embedding = self.extract_features(x, *args, **kwargs)
        return self.classifier(cls_embedding), image_embedding

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        prediction, _ = self.forward_classifier(x, *args, **kwargs)
        return prediction


def build_vit(mode: str = "tiny", opts=None, **overrides) -> VisionTransformer:
    from .layers import default_opts

    if opts is None:
        base = {"model.classification.vit.mode": mode, "model.activation.name": "gelu"}
        base.update(overrides)
        opts = default_opts(**base)
    return VisionTransformer(opts)


# =============================================================================================
# MobileViTv2  (cvnets/models/classification/mobilevit_v2.py:20-226, config/mobilevit_v2.py:11-77)
# =============================================================================================
def get_mitv2_configuration(opts) -> Dict:
    wm = opt(opts, "model.classification.mitv2.width_multiplier", 1.0)
    ffn_multiplier, mv2_exp_mult = 2, 2
    layer_0_dim = int(make_divisible(max(16, min(64, 32 * wm)), divisor=8, min_value=16))
    cfg = {
        "layer0": {"img_channels": 3, "out_channels": layer_0_dim},
        "layer1": {"out_channels": int(make_divisible(64 * wm, divisor=16)), "expand_ratio": mv2_exp_mult, "num_blocks": 1, "stride": 1,
                   "block_type": "mv2"},
        "layer2": {"out_channels": int(make_divisible(128 * wm, divisor=8)), "expand_ratio": mv2_exp_mult, "num_blocks": 2, "stride": 2,
                   "block_type": "mv2"},
        "last_layer_exp_factor": 4,
    }
    for name, out, attn, nblk in (("layer3", 256, 128, 2), ("layer4", 384, 192, 4), ("layer5", 512, 256, 3)):
        cfg[name] = {"out_channels": int(make_divisible(out * wm, divisor=8)), "attn_unit_dim": int(make_divisible(attn * wm, divisor=8)),
                     "ffn_multiplier": ffn_multiplier, "attn_blocks": nblk, "patch_h": 2, "patch_w": 2, "stride": 2,
                     "mv_expand_ratio": mv2_exp_mult, "block_type": "mobilevit"}
    return cfg


class MobileViTv2(nn.Module):
    def __init__(self, opts, *args, **kwargs) -> None:
        super().__init__()
        num_classes = opt(opts, "model.classification.n_classes", 1000)
        pool_type = opt(opts, "model.layer.global_pool", "mean")
        cfg = get_mitv2_configuration(opts)
        image_channels, out_channels = cfg["layer0"]["img_channels"], cfg["layer0"]["out_channels"]
        _init_encoder_common(self, opts, kwargs)
        self.conv_1 = ConvLayer2d(opts=opts, in_channels=image_channels, out_channels=out_channels, kernel_size=3, stride=2, use_norm=True,
                                  use_act=True)
        self.model_conf_dict["conv1"] = {"in": image_channels, "out": out_channels}
        in_channels = out_channels
        for idx in range(1, 6):
            layer, out_channels = self._make_layer(opts=opts, input_channel=in_channels, cfg=cfg[f"layer{idx}"],
                                                   dilate=(idx == 4 and self.dilate_l4) or (idx == 5 and self.dilate_l5))
            setattr(self, f"layer_{idx}", layer)
            self.model_conf_dict[f"layer{idx}"] = {"in": in_channels, "out": out_channels}
            in_channels = out_channels
        from .layers import Identity

        self.conv_1x1_exp = Identity()
        self.model_conf_dict["exp_before_cls"] = {"in": out_channels, "out": out_channels}
        self.classifier = nn.Sequential(GlobalPool(pool_type=pool_type, keep_dim=False),
                                        LinearLayer(in_features=out_channels, out_features=num_classes, bias=True))
        self.n_classes = num_classes

    def _make_layer(self, opts, input_channel, cfg: Dict, dilate: bool = False) -> Tuple[nn.Sequential, int]:
        if cfg.get("block_type", "mobilevit").lower() == "mobilevit":
            return self._make_mit_layer(opts, input_channel, cfg, dilate=dilate)
        return MobileViT._make_mobilenet_layer(opts, input_channel, cfg)

    def _make_mit_layer(self, opts, input_channel, cfg: Dict, dilate: bool = False) -> Tuple[nn.Sequential, int]:
        prev_dilation = self.dilation  # mobilevit_v2.py:165-190: stride traded for dilation (output_stride 16 / 8)
        block = []
        stride = cfg.get("stride", 1)
        if stride == 2:
            if dilate:
                self.dilation *= 2
                stride = 1
            block.append(InvertedResidual(opts=opts, in_channels=input_channel, out_channels=cfg.get("out_channels"), stride=stride,
                                          expand_ratio=cfg.get("mv_expand_ratio", 4), dilation=prev_dilation))
            input_channel = cfg.get("out_channels")
        block.append(MobileViTBlockv2(
            opts=opts, in_channels=input_channel, attn_unit_dim=cfg["attn_unit_dim"], ffn_multiplier=cfg.get("ffn_multiplier"),
            n_attn_blocks=cfg.get("attn_blocks", 1), patch_h=cfg.get("patch_h", 2), patch_w=cfg.get("patch_w", 2),
            dropout=opt(opts, "model.classification.mitv2.dropout", 0.0), ffn_dropout=opt(opts, "model.classification.mitv2.ffn_dropout", 0.0),
            attn_dropout=opt(opts, "model.classification.mitv2.attn_dropout", 0.0), conv_ksize=3,
            attn_norm_layer=opt(opts, "model.classification.mitv2.attn_norm_layer", "layer_norm_2d"), dilation=self.dilation))
        return nn.Sequential(*block), input_channel

    extract_end_points_all = _extract_end_points_all
    extract_end_points_l4 = _extract_end_points_l4
    extract_features = _extract_features

    def forward_classifier(self, x: Tensor, *args, **kwargs) -> Tensor:
        return self.classifier(self.extract_features(x))

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        return self.forward_classifier(x, *args, **kwargs)


def build_mobilevit_v2(width_multiplier: float = 1.0, opts=None, **overrides) -> MobileViTv2:
    from .layers import default_opts

    if opts is None:
        opts = default_opts(**{"model.classification.mitv2.width_multiplier": width_multiplier}, **overrides)
    return MobileViTv2(opts)
