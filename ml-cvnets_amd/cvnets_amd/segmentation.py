"""Semantic-segmentation consumers of the backbone: the DeepLabv3 head on a MobileViT / MobileViTv2 encoder.

Mirrors (same constructor arguments, attribute tree and state_dict keys, so a reference-built model can be class-swapped):
  cvnets/models/segmentation/enc_dec.py:21-153          SegEncoderDecoder
  cvnets/models/segmentation/heads/base_seg_head.py:24-112   BaseSegHead (aux head, up-sampling of the mask)
  cvnets/models/segmentation/heads/deeplabv3.py:19-126  DeeplabV3
  cvnets/modules/aspp_block.py:22-248                   ASPP, ASPPConv2d, ASPPPooling
  cvnets/models/segmentation/heads/pspnet.py:19-115     PSPNet;  cvnets/modules/pspnet_module.py:17-114  PSP

Every tensor op runs on the HIP kernels of the backbone path: dilated dense 3x3 / 1x1 convs + BatchNorm + ReLU (cvh_conv_gemm, cvh_bn_*),
global average pool, bilinear resize (both corner conventions), channel concat (cvh_cat_channels), Dropout2d (cvh_dropout2d).  The
classifier's 21 classes are not a multiple of the 8-channel NHWC granule: its weight / bias are zero-padded to 24 output channels
(autograd-visible padding of two tiny tensors), the mask is up-sampled with 24 channels and the first n_classes are returned.
PSPNet: adaptive average pools to 1/2/3/6 bins (cvh_adaptive_pool_*).  Not built: the separable-conv ASPP variant, SSD detection heads.
"""
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops
from .layers import AdaptiveAvgPool2d, ConvLayer2d, Dropout2d, UpSample, opt


def _conv_padded_classes(layer: ConvLayer2d, x: Tensor) -> Tensor:
    """1x1 conv + bias whose out_channels (n_classes) is not a multiple of 8: returns the map with pad8(n_classes) channels (extra ones = 0)"""
    conv = layer.block.conv
    n = conv.out_channels
    n8 = ops.pad8(n)
    if n8 == n:
        return layer(x)
    w = F.pad(conv.weight, (0, 0, 0, 0, 0, 0, 0, n8 - n))   # plumbing: [n, Cin, 1, 1] -> [n8, Cin, 1, 1]
    b = F.pad(conv.bias, (0, n8 - n)) if conv.bias is not None else None
    return ops.conv_bn_act(ops.to_nhwc(x), w, b, None, None, None, None, stride=1, pad=0, dil=1, act=ops.ACT_NONE, use_bn=False, training=layer.training)


class ASPPConv2d(ConvLayer2d):
    """cvnets/modules/aspp_block.py:138-169: 3x3 conv with dilation `rate` (padding = rate) -> BatchNorm -> activation"""

    def __init__(self, opts, in_channels: int, out_channels: int, dilation: int, *args, **kwargs) -> None:
        super().__init__(opts=opts, in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=1, use_norm=True, use_act=True,
                         dilation=dilation)

    def adjust_atrous_rate(self, rate: int) -> None:
        self.block.conv.dilation = (rate, rate)
        self.block.conv.padding = (rate, rate)


class ASPPPooling(nn.Module):
    """cvnets/modules/aspp_block.py:204-248: global average pool -> 1x1 conv-BN-act -> bilinear (align_corners=False) back to the input size"""

    def __init__(self, opts, in_channels: int, out_channels: int, *args, **kwargs) -> None:
        super().__init__()
        self.aspp_pool = nn.Sequential()
        self.aspp_pool.add_module(name="global_pool", module=AdaptiveAvgPool2d(output_size=1))
        self.aspp_pool.add_module(name="conv_1x1", module=ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=out_channels, kernel_size=1,
                                                                       stride=1, use_norm=True, use_act=True))
        self.in_channels = in_channels
        self.out_channels = out_channels

    def forward(self, x: Tensor) -> Tensor:
        H, W = x.shape[-2:]
        y = self.aspp_pool(x)
        return ops.resize_bilinear(ops.to_nhwc(y), H, W, False)

    def __repr__(self):
        return "{}(in_channels={}, out_channels={})".format(self.__class__.__name__, self.in_channels, self.out_channels)


class ASPP(nn.Module):
    """cvnets/modules/aspp_block.py:22-135"""

    def __init__(self, opts, in_channels: int, out_channels: int, atrous_rates: Tuple[int], is_sep_conv: Optional[bool] = False,
                 dropout: Optional[float] = 0.0, *args, **kwargs) -> None:
        super().__init__()
        if is_sep_conv:
            raise NotImplementedError("the separable-conv ASPP variant is not on the HIP path")
        assert len(atrous_rates) == 3
        modules = [ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, use_norm=True, use_act=True)]
        modules.extend([ASPPConv2d(opts=opts, in_channels=in_channels, out_channels=out_channels, dilation=rate) for rate in atrous_rates])
        modules.append(ASPPPooling(opts=opts, in_channels=in_channels, out_channels=out_channels))
        if not (0.0 <= dropout < 1.0):
            dropout = 0.0
        self.convs = nn.ModuleList(modules)
        self.project = ConvLayer2d(opts=opts, in_channels=5 * out_channels, out_channels=out_channels, kernel_size=1, stride=1, use_norm=True,
                                   use_act=True)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.atrous_rates = atrous_rates
        self.is_sep_conv_layer = is_sep_conv
        self.n_atrous_branches = len(atrous_rates)
        self.dropout_layer = Dropout2d(p=dropout)

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        out = ops.cat_channels([conv(x) for conv in self.convs])
        return self.dropout_layer(self.project(out))

    def __repr__(self):
        return "{}(in_channels={}, out_channels={}, atrous_rates={}, is_aspp_sep={}, dropout={})".format(
            self.__class__.__name__, self.in_channels, self.out_channels, self.atrous_rates, self.is_sep_conv_layer, self.dropout_layer.p)


class BaseSegHead(nn.Module):
    """cvnets/models/segmentation/heads/base_seg_head.py:24-112"""

    def __init__(self, opts, enc_conf: dict, use_l5_exp: Optional[bool] = False, *args, **kwargs):
        super().__init__()
        ch = lambda name: enc_conf[name]["out"]
        self.use_l5_exp = use_l5_exp
        self.enc_l5_exp_channels = ch("exp_before_cls")
        self.enc_l5_channels = ch("layer5")
        self.enc_l4_channels = ch("layer4")
        self.enc_l3_channels = ch("layer3")
        self.enc_l2_channels = ch("layer2")
        self.enc_l1_channels = ch("layer1")
        self.n_seg_classes = opt(opts, "model.segmentation.n_classes", 21)
        self.lr_multiplier = opt(opts, "model.segmentation.lr_multiplier", 1.0)
        self.classifier_dropout = opt(opts, "model.segmentation.classifier_dropout", 0.1)
        self.output_stride = opt(opts, "model.segmentation.output_stride", 16)
        self.aux_head = None
        if opt(opts, "model.segmentation.use_aux_head", False):
            drop_aux = opt(opts, "model.segmentation.aux_dropout", 0.1)
            inner_channels = max(int(self.enc_l4_channels // 4), 128)
            self.aux_head = nn.Sequential(
                ConvLayer2d(opts=opts, in_channels=self.enc_l4_channels, out_channels=inner_channels, kernel_size=3, stride=1, use_norm=True,
                            use_act=True, bias=False, groups=1),
                Dropout2d(drop_aux),
                ConvLayer2d(opts=opts, in_channels=inner_channels, out_channels=self.n_seg_classes, kernel_size=1, stride=1, use_norm=False,
                            use_act=False, bias=True, groups=1),
            )
        self.upsample_seg_out = None
        if self.output_stride != 1.0:
            self.upsample_seg_out = UpSample(scale_factor=self.output_stride, mode="bilinear", align_corners=True)

    def forward_aux_head(self, enc_out: Dict) -> Tensor:
        x = self.aux_head[1](self.aux_head[0](enc_out["out_l4"]))
        y = _conv_padded_classes(self.aux_head[2], x)
        return y[:, : self.n_seg_classes]

    def forward_seg_head(self, enc_out: Dict) -> Tensor:
        raise NotImplementedError

    def forward(self, enc_out: Dict, *args, **kwargs) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        out = self.forward_seg_head(enc_out=enc_out)  # pad8(n_classes) channels
        if self.upsample_seg_out is not None:
            mask_size = kwargs.get("orig_size", None)
            if mask_size is not None:
                self.upsample_seg_out.scale_factor = None
                self.upsample_seg_out.size = mask_size
            out = self.upsample_seg_out(out)
        out = out[:, : self.n_seg_classes]
        if self.aux_head is not None and self.training:
            return out, self.forward_aux_head(enc_out=enc_out)
        return out


class DeeplabV3(BaseSegHead):
    """cvnets/models/segmentation/heads/deeplabv3.py:19-126"""

    def __init__(self, opts, enc_conf: Dict, use_l5_exp: Optional[bool] = False, *args, **kwargs) -> None:
        atrous_rates = opt(opts, "model.segmentation.deeplabv3.aspp_rates", (6, 12, 18))
        out_channels = opt(opts, "model.segmentation.deeplabv3.aspp_out_channels", 256)
        is_sep_conv = opt(opts, "model.segmentation.deeplabv3.aspp_sep_conv", False)
        dropout = opt(opts, "model.segmentation.deeplabv3.aspp_dropout", 0.1)
        super().__init__(opts=opts, enc_conf=enc_conf, use_l5_exp=use_l5_exp)
        self.aspp = nn.Sequential()
        aspp_in_channels = self.enc_l5_channels if not self.use_l5_exp else self.enc_l5_exp_channels
        self.aspp.add_module(name="aspp_layer", module=ASPP(opts=opts, in_channels=aspp_in_channels, out_channels=out_channels,
                                                            atrous_rates=tuple(atrous_rates), is_sep_conv=is_sep_conv, dropout=dropout))
        self.classifier = ConvLayer2d(opts=opts, in_channels=out_channels, out_channels=self.n_seg_classes, kernel_size=1, stride=1,
                                      use_norm=False, use_act=False, bias=True)

    def forward_seg_head(self, enc_out: Dict) -> Tensor:
        x = enc_out["out_l5_exp"] if self.use_l5_exp else enc_out["out_l5"]
        return _conv_padded_classes(self.classifier, self.aspp(x))


class PSP(nn.Module):
    """cvnets/modules/pspnet_module.py:17-114: pyramid pooling — adaptive average pools to `pool_sizes` bins, 1x1 conv-BN-act each,
    bilinear (align_corners=True) back to the input size, concat with the input, 3x3 conv-BN-act fusion, Dropout2d"""

    def __init__(self, opts, in_channels: int, out_channels: int, pool_sizes: Optional[Tuple[int, ...]] = (1, 2, 3, 6),
                 dropout: Optional[float] = 0.0, *args, **kwargs) -> None:
        super().__init__()
        reduction_dim = in_channels // len(pool_sizes)
        reduction_dim = (reduction_dim // 16) * 16
        channels_after_concat = (reduction_dim * len(pool_sizes)) + in_channels
        self.psp_branches = nn.ModuleList([self._make_psp_layer(opts, o_size=ps, in_channels=in_channels, out_channels=reduction_dim)
                                           for ps in pool_sizes])
        self.fusion = nn.Sequential(
            ConvLayer2d(opts=opts, in_channels=channels_after_concat, out_channels=out_channels, kernel_size=3, stride=1, use_norm=True, use_act=True),
            Dropout2d(p=dropout))
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.pool_sizes = pool_sizes
        self.inner_channels = reduction_dim
        self.dropout = dropout

    @staticmethod
    def _make_psp_layer(opts, o_size: int, in_channels: int, out_channels: int) -> nn.Module:
        return nn.Sequential(AdaptiveAvgPool2d(output_size=(o_size, o_size)),
                             ConvLayer2d(opts, in_channels=in_channels, out_channels=out_channels, kernel_size=1, bias=False, use_norm=True, use_act=True))

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        H, W = x.shape[-2:]
        x = ops.to_nhwc(x)
        out = [x] + [ops.resize_bilinear(ops.to_nhwc(branch(x)), H, W, True) for branch in self.psp_branches]
        return self.fusion(ops.cat_channels(out))

    def __repr__(self):
        return "{}(in_channels={}, out_channels={}, pool_sizes={}, inner_channels={}, dropout_2d={})".format(
            self.__class__.__name__, self.in_channels, self.out_channels, self.pool_sizes, self.inner_channels, self.dropout)


class PSPNet(BaseSegHead):
    """cvnets/models/segmentation/heads/pspnet.py:19-115"""

    def __init__(self, opts, enc_conf: Dict, use_l5_exp: Optional[bool] = False, *args, **kwargs) -> None:
        psp_out_channels = opt(opts, "model.segmentation.pspnet.psp_out_channels", 512)
        psp_pool_sizes = opt(opts, "model.segmentation.pspnet.psp_pool_sizes", [1, 2, 3, 6])
        psp_dropout = opt(opts, "model.segmentation.pspnet.psp_dropout", 0.1)
        super().__init__(opts=opts, enc_conf=enc_conf, use_l5_exp=use_l5_exp)
        psp_in_channels = self.enc_l5_channels if not self.use_l5_exp else self.enc_l5_exp_channels
        self.psp_layer = PSP(opts=opts, in_channels=psp_in_channels, out_channels=psp_out_channels, pool_sizes=tuple(psp_pool_sizes), dropout=psp_dropout)
        self.classifier = ConvLayer2d(opts=opts, in_channels=psp_out_channels, out_channels=self.n_seg_classes, kernel_size=1, stride=1,
                                      use_norm=False, use_act=False, bias=True)

    def forward_seg_head(self, enc_out: Dict) -> Tensor:
        x = enc_out["out_l5_exp"] if self.use_l5_exp else enc_out["out_l5"]
        return _conv_padded_classes(self.classifier, self.psp_layer(x))


class SegEncoderDecoder(nn.Module):
    """cvnets/models/segmentation/enc_dec.py:21-153 (forward :91-108)"""

    def __init__(self, opts, encoder: nn.Module, seg_head: nn.Module, *args, **kwargs) -> None:
        super().__init__()
        self.encoder = encoder
        self.encoder.classifier = None
        use_l5_exp = opt(opts, "model.segmentation.use_level5_exp", False)
        if not use_l5_exp:
            self.encoder.conv_1x1_exp = None
        self.seg_head = seg_head
        self.use_l5_exp = use_l5_exp

    def forward(self, x: Tensor, *args, **kwargs):
        enc_end_points = self.encoder.extract_end_points_all(x, use_l5=True, use_l5_exp=self.use_l5_exp)
        return self.seg_head(enc_out=enc_end_points, *args, **kwargs)


def build_segmentation(opts, encoder: str = "mobilevit", head: str = "deeplabv3", head_activation: str = "relu") -> SegEncoderDecoder:
    """config/segmentation/{pascal_voc,ade20k}/{deeplabv3,pspnet}_mobilevit{,v2}.yaml: a MobileViT / MobileViTv2 encoder (its own
    activation, layers 4-5 dilated to model.segmentation.output_stride) + a DeepLabv3 / PSPNet head whose ConvLayer2d blocks use
    model.segmentation.activation.name (relu in the reference YAMLs)."""
    import copy

    from .models import MobileViT, MobileViTv2

    enc_cls = {"mobilevit": MobileViT, "mobilevit_v2": MobileViTv2}[encoder]
    enc = enc_cls(opts, output_stride=opt(opts, "model.segmentation.output_stride", None))
    head_opts = copy.copy(opts)
    setattr(head_opts, "model.activation.name", head_activation)
    head_cls = {"deeplabv3": DeeplabV3, "pspnet": PSPNet}[head]
    seg_head = head_cls(head_opts, enc_conf=enc.model_conf_dict, use_l5_exp=opt(opts, "model.segmentation.use_level5_exp", False))
    return SegEncoderDecoder(opts, encoder=enc, seg_head=seg_head)


def build_deeplabv3_mobilevit(opts, head_activation: str = "relu") -> SegEncoderDecoder:
    return build_segmentation(opts, "mobilevit", "deeplabv3", head_activation)
