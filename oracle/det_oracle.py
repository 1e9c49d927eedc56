"""ORACLE (test infrastructure only — never imported by the product): CPU / fp32 functional restatement of the SSD detection head over
a state_dict, pinned bit-exact against the reference classes by oracle/make_golden.py --detection.

Follows  cvnets/models/detection/ssd.py:300-352 (get_backbone_features, ssd_forward), cvnets/modules/ssd_heads.py:117-132 (SSDHead.forward),
         cvnets/layers/conv_layer.py:474-591 (SeparableConv2d: depthwise conv -> BatchNorm, pointwise conv -> norm -> act).
"""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


def _bn(sd, prefix: str, y: Tensor, training: bool, bn_state: Dict, momentum: float = 0.1) -> Tensor:
    rm, rv = sd[prefix + ".running_mean"].clone(), sd[prefix + ".running_var"].clone()
    y = F.batch_norm(y, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], training, momentum, 1e-5)
    if training:
        bn_state[prefix + ".running_mean"], bn_state[prefix + ".running_var"] = rm, rv
    return y


def separable_conv(sd, prefix: str, x: Tensor, stride: int, use_norm: bool, use_act: bool, training: bool, bn_state: Dict) -> Tensor:
    w = sd[prefix + ".dw_conv.block.conv.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2, groups=w.shape[0])
    y = _bn(sd, prefix + ".dw_conv.block.norm", y, training, bn_state)
    y = F.conv2d(y, sd[prefix + ".pw_conv.block.conv.weight"], sd.get(prefix + ".pw_conv.block.conv.bias"))
    if use_norm:
        y = _bn(sd, prefix + ".pw_conv.block.norm", y, training, bn_state)
    return F.relu(y) if use_act else y


def ssd_head(sd, prefix: str, x: Tensor, n_classes: int, kernel_size: int, training: bool, bn_state: Dict) -> Tuple[Tensor, Tensor]:
    B = x.shape[0]
    if prefix + ".proj_layer.block.conv.weight" in sd:
        x = F.conv2d(x, sd[prefix + ".proj_layer.block.conv.weight"])
        x = F.relu(_bn(sd, prefix + ".proj_layer.block.norm", x, training, bn_state))
    if kernel_size == 1:
        x = F.conv2d(x, sd[prefix + ".loc_cls_layer.block.conv.weight"], sd[prefix + ".loc_cls_layer.block.conv.bias"])
    else:
        x = separable_conv(sd, prefix + ".loc_cls_layer", x, 1, False, False, training, bn_state)
    x = x.permute(0, 2, 3, 1).contiguous().view(B, -1, 4 + n_classes)
    return x[..., :4], x[..., 4:]


def ssd_forward(sd, enc_out: Dict[str, Tensor], output_strides: List[int], n_classes: int = 81, training: bool = True):
    """returns (scores [B, A, n_classes], boxes [B, A, 4], updated BatchNorm running statistics)"""
    bn_state: Dict[str, Tensor] = {}
    end_points = {}
    for os_ in output_strides:
        if os_ == 16:
            end_points[16] = enc_out["out_l4"]
        elif os_ == 32:
            end_points[32] = enc_out["out_l5"]
    x = end_points[32]
    for os_ in output_strides:
        if os_ > 32:
            x = separable_conv(sd, f"extra_layers.os_{os_}", x, 2, True, True, training, bn_state)
            end_points[os_] = x
        elif os_ == -1:
            x = F.relu(F.conv2d(F.adaptive_avg_pool2d(x, 1), sd["extra_layers.os_-1.1.block.conv.weight"]))
            end_points[os_] = x
    locs, confs = [], []
    for i, os_ in enumerate(output_strides):
        loc, conf = ssd_head(sd, f"ssd_heads.{i}", end_points[os_], n_classes, 3 if os_ != -1 else 1, training, bn_state)
        locs.append(loc)
        confs.append(conf)
    return torch.cat(confs, 1), torch.cat(locs, 1), bn_state
