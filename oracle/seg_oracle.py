"""ORACLE (test infrastructure only — never imported by the product): CPU / fp32 functional restatement of the DeepLabv3 segmentation
head over a state_dict, pinned bit-exact against the reference classes by oracle/make_golden.py --segmentation.

Follows  cvnets/modules/pspnet_module.py:17-114 and cvnets/models/segmentation/heads/pspnet.py:19-115 (PSPNet),
         cvnets/modules/aspp_block.py:22-248 (ASPP.forward :118-123, ASPPPooling.forward :238-243),
         cvnets/models/segmentation/heads/deeplabv3.py:122-126 (forward_seg_head),
         cvnets/models/segmentation/heads/base_seg_head.py:92-112 (forward: up-sampling, auxiliary head),
         loss_fn/segmentation/cross_entropy.py:96-116 (_compute_loss) and :172-176 (total = seg + aux_weight * aux).
"""
from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


def _conv_bn_relu(sd, prefix: str, x: Tensor, *, dilation: int = 1, training: bool, bn_state: Dict, momentum: float = 0.1) -> Tensor:
    w = sd[prefix + ".block.conv.weight"]
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride=1, padding=(k // 2) * dilation, dilation=dilation)
    rm, rv = sd[prefix + ".block.norm.running_mean"].clone(), sd[prefix + ".block.norm.running_var"].clone()
    y = F.batch_norm(y, rm, rv, sd[prefix + ".block.norm.weight"], sd[prefix + ".block.norm.bias"], training, momentum, 1e-5)
    if training:
        bn_state[prefix + ".block.norm.running_mean"] = rm
        bn_state[prefix + ".block.norm.running_var"] = rv
    return F.relu(y)


def aspp(sd, prefix: str, x: Tensor, rates: Tuple[int, int, int], training: bool, bn_state: Dict) -> Tensor:
    outs = [_conv_bn_relu(sd, prefix + ".convs.0", x, training=training, bn_state=bn_state)]
    for i, r in enumerate(rates):
        outs.append(_conv_bn_relu(sd, f"{prefix}.convs.{i + 1}", x, dilation=r, training=training, bn_state=bn_state))
    pooled = F.adaptive_avg_pool2d(x, 1)
    pooled = _conv_bn_relu(sd, prefix + ".convs.4.aspp_pool.conv_1x1", pooled, training=training, bn_state=bn_state)
    outs.append(F.interpolate(pooled, size=x.shape[-2:], mode="bilinear", align_corners=False))
    return _conv_bn_relu(sd, prefix + ".project", torch.cat(outs, dim=1), training=training, bn_state=bn_state)  # Dropout2d: p = 0 in parity runs


def deeplabv3_head(sd, prefix: str, enc_out: Dict[str, Tensor], rates=(12, 24, 36), output_stride: int = 8, training: bool = True, use_aux: bool = True):
    """returns (mask logits [B, n_classes, H, W], auxiliary logits or None, updated BatchNorm running statistics)"""
    bn_state: Dict[str, Tensor] = {}
    y = aspp(sd, prefix + ".aspp.aspp_layer", enc_out["out_l5"], tuple(rates), training, bn_state)
    y = F.conv2d(y, sd[prefix + ".classifier.block.conv.weight"], sd[prefix + ".classifier.block.conv.bias"])
    if output_stride != 1:
        y = F.interpolate(y, scale_factor=float(output_stride), mode="bilinear", align_corners=True)
    aux = None
    if use_aux and training:
        a = _conv_bn_relu(sd, prefix + ".aux_head.0", enc_out["out_l4"], training=training, bn_state=bn_state)
        aux = F.conv2d(a, sd[prefix + ".aux_head.2.block.conv.weight"], sd[prefix + ".aux_head.2.block.conv.bias"])
    return y, aux, bn_state


def psp(sd, prefix: str, x: Tensor, pool_sizes, training: bool, bn_state: Dict) -> Tensor:
    """cvnets/modules/pspnet_module.py:91-103"""
    outs = [x]
    for i, ps in enumerate(pool_sizes):
        y = F.adaptive_avg_pool2d(x, ps)
        y = _conv_bn_relu(sd, f"{prefix}.psp_branches.{i}.1", y, training=training, bn_state=bn_state)
        outs.append(F.interpolate(y, size=x.shape[-2:], mode="bilinear", align_corners=True))
    return _conv_bn_relu(sd, prefix + ".fusion.0", torch.cat(outs, dim=1), training=training, bn_state=bn_state)  # Dropout2d: p = 0 in parity runs


def pspnet_head(sd, prefix: str, enc_out: Dict[str, Tensor], pool_sizes=(1, 2, 3, 6), output_stride: int = 16, training: bool = True, use_aux: bool = True):
    """cvnets/models/segmentation/heads/pspnet.py:99-115 + base_seg_head.py:92-112"""
    bn_state: Dict[str, Tensor] = {}
    y = psp(sd, prefix + ".psp_layer", enc_out["out_l5"], tuple(pool_sizes), training, bn_state)
    y = F.conv2d(y, sd[prefix + ".classifier.block.conv.weight"], sd[prefix + ".classifier.block.conv.bias"])
    if output_stride != 1:
        y = F.interpolate(y, scale_factor=float(output_stride), mode="bilinear", align_corners=True)
    aux = None
    if use_aux and training:
        a = _conv_bn_relu(sd, prefix + ".aux_head.0", enc_out["out_l4"], training=training, bn_state=bn_state)
        aux = F.conv2d(a, sd[prefix + ".aux_head.2.block.conv.weight"], sd[prefix + ".aux_head.2.block.conv.bias"])
    return y, aux, bn_state


def seg_loss(mask_logits: Tensor, aux_logits, target: Tensor, aux_weight: float = 0.4, ignore_index: int = 255) -> Tensor:
    def one(pred):
        if pred.shape[-2:] != target.shape[-2:]:
            pred = F.interpolate(pred, size=target.shape[-2:], mode="bilinear", align_corners=True)
        return F.cross_entropy(pred, target, ignore_index=ignore_index)
    loss = one(mask_logits)
    if aux_logits is not None:
        loss = loss + aux_weight * one(aux_logits)
    return loss
