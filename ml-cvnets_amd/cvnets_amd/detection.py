"""Object-detection consumer of the backbone: the SSD head on a MobileViT / MobileViTv2 encoder.

Mirrors (same constructor arguments, attribute tree and state_dict keys, so a reference-built model can be class-swapped):
  cvnets/models/detection/ssd.py:33-352          SingleShotMaskDetector (__init__, get_backbone_features, ssd_forward, forward)
  cvnets/modules/ssd_heads.py:17-132             SSDHead
  cvnets/layers/conv_layer.py:474-591            SeparableConv2d (depthwise 3x3 + BatchNorm -> pointwise 1x1)
  cvnets/anchor_generator/ssd_anchor_generator.py:19-195   SSDAnchorGenerator (host arithmetic, no kernels)

Every tensor op of the TRAINING forward (scores, boxes) runs on the HIP kernels of the backbone path: depthwise 3x3 (stride 1 / 2) +
BatchNorm, 1x1 convs (+BatchNorm, +ReLU), adaptive average pool.  Prediction widths n_anchors * (4 + n_classes) are not multiples of
the 8-channel NHWC granule: the pointwise weight / bias are zero-padded (autograd-visible padding of two small tensors) and the
padding is dropped when the map is reshaped to [B, anchors, 4 + n_classes].  Not on the HIP path: box decoding + NMS of the eval /
predict branch (torchvision ops on a few hundred boxes), the matcher / multibox loss (adjacent, kept), the FPN variant.
"""
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops
from .layers import AdaptiveAvgPool2d, ConvLayer2d, opt


def _pointwise_padded(layer: ConvLayer2d, x: Tensor) -> Tensor:
    """1x1 conv (+bias, no norm / act) whose out_channels is not a multiple of 8: returns the map with pad8(out_channels) channels"""
    conv = layer.block.conv
    n, n8 = conv.out_channels, ops.pad8(conv.out_channels)
    if n8 == n or "norm" in layer.block._modules or "act" in layer.block._modules:
        return layer(x)
    w = F.pad(conv.weight, (0, 0, 0, 0, 0, 0, 0, n8 - n))  # plumbing: [n, Cin, 1, 1] -> [n8, Cin, 1, 1]
    b = F.pad(conv.bias, (0, n8 - n)) if conv.bias is not None else None
    return ops.conv_bn_act(ops.to_nhwc(x), w, b, None, None, None, None, stride=1, pad=0, dil=1, act=ops.ACT_NONE, use_bn=False, training=layer.training)


class SeparableConv2d(nn.Module):
    """cvnets/layers/conv_layer.py:474-591: depthwise conv -> BatchNorm (no activation by default), then pointwise conv (-> norm -> act)"""

    def __init__(self, opts, in_channels: int, out_channels: int, kernel_size, stride=1, dilation=1, use_norm: bool = True, use_act: bool = True,
                 use_act_depthwise: bool = False, bias: bool = False, padding_mode: str = "zeros", act_name: Optional[str] = None, *args, **kwargs) -> None:
        super().__init__()
        if act_name is not None:
            raise NotImplementedError("per-layer activation override")
        self.dw_conv = ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=in_channels, kernel_size=kernel_size, stride=stride,
                                   dilation=dilation, groups=in_channels, bias=False, padding_mode=padding_mode, use_norm=True, use_act=use_act_depthwise)
        self.pw_conv = ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, dilation=1, groups=1,
                                   bias=bias, padding_mode=padding_mode, use_norm=use_norm, use_act=use_act)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.stride = stride
        self.kernel_size = kernel_size
        self.dilation = dilation

    def forward(self, x: Tensor) -> Tensor:
        return self.pw_conv(self.dw_conv(x))

    def __repr__(self):
        return "{}(in_channels={}, out_channels={}, kernel_size={}, stride={}, dilation={})".format(
            self.__class__.__name__, self.in_channels, self.out_channels, self.kernel_size, self.stride, self.dilation)


class SSDHead(nn.Module):
    """cvnets/modules/ssd_heads.py:17-132"""

    def __init__(self, opts, in_channels: int, n_anchors: int, n_classes: int, n_coordinates: Optional[int] = 4, proj_channels: Optional[int] = -1,
                 kernel_size: Optional[int] = 3, stride: Optional[int] = 1, *args, **kwargs) -> None:
        super().__init__()
        proj_layer = None
        self.proj_channels = None
        if proj_channels != -1 and proj_channels != in_channels and kernel_size > 1:
            proj_layer = ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=proj_channels, kernel_size=1, stride=1, groups=1, bias=False,
                                     use_norm=True, use_act=True)
            in_channels = proj_channels
            self.proj_channels = proj_channels
        self.proj_layer = proj_layer
        conv_fn = ConvLayer2d if kernel_size == 1 else SeparableConv2d
        if kernel_size > 1 and stride > 1:
            kernel_size = max(kernel_size, stride if stride % 2 != 0 else stride + 1)
        self.loc_cls_layer = conv_fn(opts=opts, in_channels=in_channels, out_channels=n_anchors * (n_coordinates + n_classes), kernel_size=kernel_size,
                                     stride=1, groups=1, bias=True, use_norm=False, use_act=False)
        self.n_coordinates = n_coordinates
        self.n_classes = n_classes
        self.n_anchors = n_anchors
        self.k_size = kernel_size
        self.stride = stride
        self.in_channel = in_channels

    def _sample_fm(self, x: Tensor) -> Tensor:
        start = max(0, self.stride // 2)
        return x[..., start::self.stride, start::self.stride]  # plumbing (strided view; materialised by the reshape below)

    def forward(self, x: Tensor, *args, **kwargs) -> Tuple[Tensor, Tensor]:
        B = x.shape[0]
        if self.proj_layer is not None:
            x = self.proj_layer(x)
        n = self.n_anchors * (self.n_coordinates + self.n_classes)
        if isinstance(self.loc_cls_layer, SeparableConv2d) or type(self.loc_cls_layer).__name__ == "SeparableConv2d":
            x = _pointwise_padded(self.loc_cls_layer.pw_conv, self.loc_cls_layer.dw_conv(x))
        else:
            x = _pointwise_padded(self.loc_cls_layer, x)
        if self.stride > 1:
            x = self._sample_fm(x)
        x = x.permute(0, 2, 3, 1)[..., :n]  # NHWC memory: a view; the padding channels are dropped here
        x = x.reshape(B, -1, self.n_coordinates + self.n_classes)  # plumbing (one small copy when padding was dropped)
        return torch.split(x, [self.n_coordinates, self.n_classes], dim=-1)

    def __repr__(self) -> str:
        s = "{}(in_channels={}, n_anchors={}, n_classes={}, n_coordinates={}, kernel_size={}, stride={}".format(
            self.__class__.__name__, self.in_channel, self.n_anchors, self.n_classes, self.n_coordinates, self.k_size, self.stride)
        if self.proj_layer is not None:
            s += ", proj=True, proj_channels={}".format(self.proj_channels)
        return s + ")"


class SSDAnchorGenerator(nn.Module):
    """cvnets/anchor_generator/ssd_anchor_generator.py:19-195 (+ base_anchor_generator.py: per-(h, w, stride) cache)"""

    def __init__(self, opts, *args, **kwargs) -> None:
        super().__init__()
        output_strides = opt(opts, "anchor_generator.ssd.output_strides", [32, 64, 128, 256, -1])
        aspect_ratios = opt(opts, "anchor_generator.ssd.aspect_ratios", [[2, 3]] * len(output_strides))
        min_ratio = opt(opts, "anchor_generator.ssd.min_scale_ratio", 0.1)
        max_ratio = opt(opts, "anchor_generator.ssd.max_scale_ratio", 1.05)
        no_clipping = opt(opts, "anchor_generator.ssd.no_clipping", False)
        step = opt(opts, "anchor_generator.ssd.step", [1])
        if isinstance(step, int):
            step = [step] * len(output_strides)
        else:
            step = list(step) + [1] * (len(output_strides) - len(step))
        aspect_ratios = [list(set(ar)) for ar in aspect_ratios]
        self.output_strides_aspect_ratio = dict(zip(output_strides, aspect_ratios))
        self.output_strides = output_strides
        self.anchors_dict = dict()
        self.num_output_strides = len(output_strides)
        self.num_aspect_ratios = len(aspect_ratios)
        scales = np.linspace(min_ratio, max_ratio, len(output_strides) + 1)
        self.sizes = {s: {"min": scales[i], "max": (scales[i] * scales[i + 1]) ** 0.5, "step": step[i]} for i, s in enumerate(output_strides)}
        self.clip = not no_clipping
        self.min_scale_ratio = min_ratio
        self.max_scale_ratio = max_ratio
        self.step = step

    def num_anchors_per_os(self) -> List:
        return [2 + 2 * len(ar) for ar in self.output_strides_aspect_ratio.values()]

    @torch.no_grad()
    def _generate_anchors(self, height: int, width: int, output_stride: int, device="cpu") -> Tensor:
        """[H' * W' * A, 4] = (cx, cy, w, h) in image fractions, cell-major then anchor-major: per cell the min-size square, the
        sqrt(min * next) square, then (w * sqrt(r), h / sqrt(r)) and its transpose for every aspect ratio r.  Built in float64 on the host
        (as the reference's Python-float loop does) and rounded to float32 once."""
        mn, mx = float(self.sizes[output_stride]["min"]), float(self.sizes[output_stride]["max"])
        step = max(1, self.sizes[output_stride]["step"])
        start = max(0, step // 2)
        shapes = [(mn, mn), (mx, mx)]
        for ratio in self.output_strides_aspect_ratio[output_stride]:
            r = ratio ** 0.5
            shapes += [(mn * r, mn / r), (mn / r, mn * r)]
        wh = np.asarray(shapes, dtype=np.float64)                                   # [A, 2]
        cy = (np.arange(start, height, step, dtype=np.float64) + 0.5) / height       # rows
        cx = (np.arange(start, width, step, dtype=np.float64) + 0.5) / width         # columns
        ctr = np.stack(np.meshgrid(cx, cy, indexing="xy"), axis=-1).reshape(-1, 1, 2)  # [H' * W', 1, (cx, cy)], row-major over (y, x)
        anchors = np.concatenate([np.broadcast_to(ctr, (ctr.shape[0], wh.shape[0], 2)), np.broadcast_to(wh[None], (ctr.shape[0], wh.shape[0], 2))], axis=-1)
        a = torch.from_numpy(anchors.reshape(-1, 4).astype(np.float32)).to(device)
        return torch.clamp(a, min=0.0, max=1.0) if self.clip else a

    @torch.no_grad()
    def forward(self, fm_height: int, fm_width: int, fm_output_stride: int, device="cpu", *args, **kwargs) -> Tensor:
        key = "h_{}_w_{}_os_{}".format(fm_height, fm_width, fm_output_stride)
        if key not in self.anchors_dict:
            self.anchors_dict[key] = self._generate_anchors(fm_height, fm_width, fm_output_stride, device=device)
        return self.anchors_dict[key].to(device)


class SingleShotMaskDetector(nn.Module):
    """cvnets/models/detection/ssd.py:33-352 — `SSD <https://arxiv.org/abs/1512.02325>`_ on a MobileViT-family encoder"""

    coordinates = 4

    def __init__(self, opts, encoder: nn.Module, *args, **kwargs) -> None:
        super().__init__()
        conf = encoder.model_conf_dict
        self.encoder = encoder
        self.n_detection_classes = opt(opts, "model.detection.n_classes", 80)
        self.enc_l5_channels, self.enc_l4_channels, self.enc_l3_channels = conf["layer5"]["out"], conf["layer4"]["out"], conf["layer3"]["out"]
        self.anchor_box_generator = SSDAnchorGenerator(opts)
        osar = self.anchor_box_generator.output_strides_aspect_ratio
        output_strides = list(osar.keys())
        self.encoder.classifier = None
        self.encoder.conv_1x1_exp = None
        proj_channels = list(opt(opts, "model.detection.ssd.proj_channels", [512, 256, 256, 128, 128, 64]))
        proj_channels = proj_channels + [128] * (len(output_strides) - len(proj_channels))
        if opt(opts, "model.detection.ssd.use_fpn", False):
            raise NotImplementedError("SSD with FPN is not on the HIP path")
        extra_layers, enc_channels_list = {}, []
        in_channels = self.enc_l5_channels
        for idx, os_ in enumerate(output_strides):
            out_channels = proj_channels[idx]
            if os_ == 8:
                enc_channels_list.append(self.enc_l3_channels)
            elif os_ == 16:
                enc_channels_list.append(self.enc_l4_channels)
            elif os_ == 32:
                enc_channels_list.append(self.enc_l5_channels)
            elif os_ > 32 and os_ != -1:
                extra_layers["os_{}".format(os_)] = SeparableConv2d(opts=opts, in_channels=in_channels, out_channels=out_channels, kernel_size=3,
                                                                    use_act=True, use_norm=True, stride=2)
                enc_channels_list.append(out_channels)
                in_channels = out_channels
            elif os_ == -1:
                extra_layers["os_{}".format(os_)] = nn.Sequential(
                    AdaptiveAvgPool2d(output_size=1),
                    ConvLayer2d(opts=opts, in_channels=in_channels, out_channels=out_channels, kernel_size=1, use_act=True, use_norm=False))
                enc_channels_list.append(out_channels)
                in_channels = out_channels
            else:
                raise NotImplementedError
        self.extra_layers = None if not extra_layers else nn.ModuleDict(extra_layers)
        self.fpn = None
        self.conf_threshold = opt(opts, "model.detection.ssd.conf_threshold", 0.01)
        self.nms_threshold = opt(opts, "model.detection.ssd.nms_iou_threshold", 0.5)
        self.top_k = opt(opts, "model.detection.ssd.top_k", 400)
        self.objects_per_image = opt(opts, "model.detection.ssd.objects_per_image", 200)
        n_anchors = self.anchor_box_generator.num_anchors_per_os()
        self.ssd_heads = nn.ModuleList()
        for os_, in_dim, proj_dim, na, step in zip(output_strides, enc_channels_list, proj_channels, n_anchors, self.anchor_box_generator.step):
            self.ssd_heads += [SSDHead(opts=opts, in_channels=in_dim, n_classes=self.n_detection_classes, n_coordinates=self.coordinates, n_anchors=na,
                                       proj_channels=proj_dim, kernel_size=3 if os_ != -1 else 1, stride=step)]
        self.anchors_aspect_ratio = n_anchors
        self.output_strides = output_strides
        self.step = self.anchor_box_generator.step

    def get_backbone_features(self, x: Tensor) -> Dict[str, Tensor]:
        enc = self.encoder.extract_end_points_all(x)
        end_points: Dict = dict()
        for os_ in self.output_strides:
            if os_ == 8:
                end_points["os_8"] = enc.pop("out_l3")
            elif os_ == 16:
                end_points["os_16"] = enc.pop("out_l4")
            elif os_ == 32:
                end_points["os_32"] = enc.pop("out_l5")
        if self.extra_layers is not None:
            x = end_points["os_{}".format(self.output_strides[len(end_points) - 1])]
            for os_, extra_layer in self.extra_layers.items():
                x = extra_layer(x)
                end_points[os_] = x
        return end_points

    def ssd_forward(self, end_points: Dict[str, Tensor], device="cpu", *args, **kwargs):
        locations, confidences, anchors = [], [], []
        for os_, head in zip(self.output_strides, self.ssd_heads):
            x = end_points["os_{}".format(os_)]
            fm_h, fm_w = x.shape[2:]
            loc, pred = head(x)
            locations.append(loc)
            confidences.append(pred)
            anchors.append(self.anchor_box_generator(fm_height=fm_h, fm_width=fm_w, fm_output_stride=os_, device=device))
        return torch.cat(confidences, dim=1), torch.cat(locations, dim=1), torch.cat(anchors, dim=0).unsqueeze(0)

    def forward(self, x: Union[Tensor, Dict]) -> Dict:
        image = x["image"] if isinstance(x, dict) else x
        confidences, locations, anchors = self.ssd_forward(self.get_backbone_features(image), device=image.device)
        out = {"scores": confidences, "boxes": locations}
        if not self.training:
            out["anchors"] = anchors  # box decoding + NMS (ssd.py:372-381, torchvision ops) stay on the reference side
        return out


def build_ssd(opts, encoder: str = "mobilevit", head_activation: str = "relu") -> SingleShotMaskDetector:
    """config/detection/ssd_coco/mobilevit{,_v2}.yaml: encoder with its own activation; extra layers and heads use model.activation.name"""
    import copy

    from .models import MobileViT, MobileViTv2

    enc = {"mobilevit": MobileViT, "mobilevit_v2": MobileViTv2}[encoder](opts)
    head_opts = copy.copy(opts)
    setattr(head_opts, "model.activation.name", head_activation)
    return SingleShotMaskDetector(head_opts, encoder=enc)
