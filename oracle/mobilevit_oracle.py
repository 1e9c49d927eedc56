"""ORACLE — test infrastructure, NOT product code.

CPU / fp32 / plain-PyTorch restatement of the CVNets MobileViT hot path (the path named by
BASELINE.json:north_star).  It is written functionally over a ``state_dict`` (no nn.Module
tree) so that it shares no code with the product package ``cvnets_amd``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity pinning: ``oracle/make_golden.py`` imports the *reference itself* (``/root/reference``
through ``oracle/ref_shim``) in the authoring container, checks this restatement against it to
fp32 round-off on identical weights/inputs (forward, loss, every parameter gradient and BN
running statistics), and commits the reference's outputs as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` re-checks the restatement against those fixtures everywhere.

Every function cites the reference file:line it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# model configuration  (cvnets/models/classification/config/mobilevit.py:11-208)
# --------------------------------------------------------------------------------------
def mobilevit_config(mode: str) -> Dict:
    mode = mode.lower()
    if mode == "xx_small":
        exp = 2
        chans = dict(l1=16, l2=24, l3=48, l4=64, l5=80)
        tdim = dict(l3=64, l4=80, l5=96)
    elif mode == "x_small":
        exp = 4
        chans = dict(l1=32, l2=48, l3=64, l4=80, l5=96)
        tdim = dict(l3=96, l4=120, l5=144)
    elif mode == "small":
        exp = 4
        chans = dict(l1=32, l2=64, l3=96, l4=128, l5=160)
        tdim = dict(l3=144, l4=192, l5=240)
    else:
        raise NotImplementedError(mode)
    return {
        "exp": exp,
        "layer1": dict(out=chans["l1"], blocks=1, stride=1),
        "layer2": dict(out=chans["l2"], blocks=3, stride=2),
        "layer3": dict(out=chans["l3"], tdim=tdim["l3"], ffn=2 * tdim["l3"], nblk=2),
        "layer4": dict(out=chans["l4"], tdim=tdim["l4"], ffn=2 * tdim["l4"], nblk=4),
        "layer5": dict(out=chans["l5"], tdim=tdim["l5"], ffn=2 * tdim["l5"], nblk=3),
        "last_exp": 4,
    }


def make_divisible(v, divisor=8, min_value=None):
    # cvnets/utils/math_utils.py (make_divisible), used at cvnets/modules/mobilenetv2.py:176
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
class BNState:
    """Collects updated BatchNorm running statistics (train mode) keyed by state_dict prefix."""

    def __init__(self):
        self.running: Dict[str, Tensor] = {}


def conv_bn_act(
    sd: Dict[str, Tensor],
    prefix: str,
    x: Tensor,
    stride: int = 1,
    groups: int = 1,
    dilation: int = 1,
    use_norm: bool = True,
    use_act: bool = True,
    training: bool = True,
    bn_state: Optional[BNState] = None,
    momentum: float = 0.1,
    eps: float = 1e-5,
    act: str = "swish",
) -> Tensor:
    """ConvLayer2d.forward = Sequential{conv, norm?, act?}  (cvnets/layers/conv_layer.py:254-255).

    conv: nn.Conv2d, padding=(k-1)//2*dilation (conv_layer.py:182-185), bias only without BN
    (conv_layer.py:157-167).  norm: nn.BatchNorm2d train-mode batch statistics
    (cvnets/layers/normalization/batch_norm.py:14-49).  act: nn.SiLU (activation/swish.py).
    """
    w = sd[prefix + ".block.conv.weight"]
    b = sd.get(prefix + ".block.conv.bias")
    k = w.shape[-1]
    pad = int((k - 1) / 2) * dilation
    y = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dilation, groups=groups)
    if use_norm:
        g = sd[prefix + ".block.norm.weight"]
        be = sd[prefix + ".block.norm.bias"]
        rm = sd[prefix + ".block.norm.running_mean"].detach().clone()
        rv = sd[prefix + ".block.norm.running_var"].detach().clone()
        y = F.batch_norm(y, rm, rv, g, be, training=training, momentum=momentum, eps=eps)
        if bn_state is not None and training:
            bn_state.running[prefix + ".block.norm.running_mean"] = rm
            bn_state.running[prefix + ".block.norm.running_var"] = rv
    if use_act:
        y = F.silu(y) if act == "swish" else F.gelu(y)
    return y


def inverted_residual(sd, prefix, x, cin, cout, stride, expand, training, bn_state) -> Tensor:
    """InvertedResidual.forward  (cvnets/modules/mobilenetv2.py:141-235)."""
    hidden = make_divisible(int(round(cin * expand)), 8)
    y = x
    if expand != 1:
        y = conv_bn_act(sd, prefix + ".block.exp_1x1", y, training=training, bn_state=bn_state)
    y = conv_bn_act(sd, prefix + ".block.conv_3x3", y, stride=stride, groups=hidden,
                    training=training, bn_state=bn_state)
    y = conv_bn_act(sd, prefix + ".block.red_1x1", y, use_act=False, training=training, bn_state=bn_state)
    if stride == 1 and cin == cout:
        return x + y
    return y


def layer_norm(sd, prefix, x: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm.forward  (cvnets/layers/normalization/layer_norm.py:51-72), INCLUDING the
    channel-first branch taken whenever x.shape[1] == C and ndim > 2 (``(x-u)/(std+eps)``)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    c = w.shape[0]
    if x.ndim > 2 and x.shape[1] == c:
        s, u = torch.std_mean(x, dim=1, keepdim=True, unbiased=False)
        x = (x - u) / (s + eps)
        shape = [1, c] + [1] * (x.ndim - 2)
        return torch.addcmul(b.reshape(shape), x, w.reshape(shape))
    return F.layer_norm(x, (c,), w, b, eps)


def multi_head_attention(sd, prefix, x: Tensor, num_heads: int,
                         attn_mask: Optional[Tensor] = None,
                         key_padding_mask: Optional[Tensor] = None, x_kv: Optional[Tensor] = None) -> Tensor:
    """MultiHeadAttention.forward_default (cvnets/layers/multi_head_attention.py:135-239): the self-attention branch (:148-157) and, with
    `x_kv` [B, T, C], the cross-attention branch (:158-185: query from the first C rows of qkv_proj on x, key / value from the other 2C
    rows on x_kv)."""
    b, s, c = x.shape
    hd = c // num_heads
    w, bias = sd[prefix + ".qkv_proj.weight"], sd.get(prefix + ".qkv_proj.bias")
    if x_kv is None:
        qkv = F.linear(x, w, bias)
        qkv = qkv.reshape(b, s, 3, num_heads, hd).transpose(1, 3).contiguous()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        t = x_kv.shape[1]
        q = F.linear(x, w[:c], None if bias is None else bias[:c])
        q = q.reshape(b, s, num_heads, hd).transpose(1, 2).contiguous()
        kv = F.linear(x_kv, w[c:], None if bias is None else bias[c:])
        kv = kv.reshape(b, t, 2, num_heads, hd).transpose(1, 3).contiguous()
        k, v = kv[:, :, 0], kv[:, :, 1]
    q = q * (hd ** -0.5)
    attn = torch.matmul(q, k.transpose(-1, -2))
    if attn_mask is not None:
        attn = attn + attn_mask.unsqueeze(1)
    if key_padding_mask is not None:
        attn = attn.masked_fill(key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf"))
    attn = torch.softmax(attn.float(), dim=-1).to(attn.dtype)
    out = torch.matmul(attn, v)
    out = out.transpose(1, 2).reshape(b, s, -1)
    return F.linear(out, sd[prefix + ".out_proj.weight"], sd.get(prefix + ".out_proj.bias"))


def transformer_encoder(sd, prefix, x: Tensor, num_heads: int, act: str = "swish",
                        ln_eps: float = 1e-5, attn_mask=None, key_padding_mask=None, drop: Optional[Dict[str, Tensor]] = None,
                        x_prev: Optional[Tensor] = None) -> Tensor:
    """TransformerEncoder.forward, drop_path Identity (cvnets/modules/transformer.py:129-156).  Dropout: p = 0 unless `drop` holds the
    multiplicative factors (0 or 1 / (1 - p), shaped like x) of the two `Dropout(p=dropout)` layers, transformer.py:82 (after the
    attention, key prefix + ".mha") and transformer.py:94 (after the second FFN linear, key prefix + ".ffn"); the generator that drew
    them is outside the oracle (the tests export the draws of the path under test).  ffn_dropout (transformer.py:92) stays 0."""
    res = x
    y = layer_norm(sd, prefix + ".pre_norm_mha.0", x, ln_eps)
    # x_prev (transformer.py:131,143-150) is handed to the attention AS GIVEN: only the query side is normalised
    y = multi_head_attention(sd, prefix + ".pre_norm_mha.1", y, num_heads, attn_mask, key_padding_mask, x_kv=x_prev)
    if drop is not None:
        y = y * drop[prefix + ".mha"]
    x = y + res
    y = layer_norm(sd, prefix + ".pre_norm_ffn.0", x, ln_eps)
    y = F.linear(y, sd[prefix + ".pre_norm_ffn.1.weight"], sd[prefix + ".pre_norm_ffn.1.bias"])
    y = F.silu(y) if act == "swish" else F.gelu(y)
    y = F.linear(y, sd[prefix + ".pre_norm_ffn.4.weight"], sd[prefix + ".pre_norm_ffn.4.bias"])
    if drop is not None:
        y = y * drop[prefix + ".ffn"]
    return x + y


def unfolding(fm: Tensor, ph: int, pw: int) -> Tuple[Tensor, Dict]:
    """MobileViTBlock.unfolding  (cvnets/modules/mobilevit_block.py:186-231)."""
    b, c, oh, ow = fm.shape
    nh_ = int(math.ceil(oh / ph) * ph)
    nw_ = int(math.ceil(ow / pw) * pw)
    interpolate = False
    if nw_ != ow or nh_ != oh:
        fm = F.interpolate(fm, size=(nh_, nw_), mode="bilinear", align_corners=False)
        interpolate = True
    npw, nph = nw_ // pw, nh_ // ph
    n = nph * npw
    p = ph * pw
    r = fm.reshape(b * c * nph, ph, npw, pw).transpose(1, 2)
    r = r.reshape(b, c, n, p).transpose(1, 3)
    patches = r.reshape(b * p, n, -1)
    info = dict(orig_size=(oh, ow), batch_size=b, interpolate=interpolate, total_patches=n,
                num_patches_w=npw, num_patches_h=nph)
    return patches, info


def folding(patches: Tensor, info: Dict, ph: int, pw: int) -> Tensor:
    """MobileViTBlock.folding  (cvnets/modules/mobilevit_block.py:233-267)."""
    p = ph * pw
    patches = patches.contiguous().view(info["batch_size"], p, info["total_patches"], -1)
    b, _, n, c = patches.size()
    nph, npw = info["num_patches_h"], info["num_patches_w"]
    patches = patches.transpose(1, 3)
    fm = patches.reshape(b * c * nph, npw, ph, pw).transpose(1, 2)
    fm = fm.reshape(b, c, nph * ph, npw * pw)
    if info["interpolate"]:
        fm = F.interpolate(fm, size=info["orig_size"], mode="bilinear", align_corners=False)
    return fm


def mobilevit_block(sd, prefix, x: Tensor, n_blocks: int, num_heads: int, training: bool,
                    bn_state, ph: int = 2, pw: int = 2, taps: Optional[Dict] = None, drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MobileViTBlock.forward_spatial  (cvnets/modules/mobilevit_block.py:269-288)."""
    res = x
    fm = conv_bn_act(sd, prefix + ".local_rep.conv_3x3", x, training=training, bn_state=bn_state)
    fm = conv_bn_act(sd, prefix + ".local_rep.conv_1x1", fm, use_norm=False, use_act=False)
    patches, info = unfolding(fm, ph, pw)
    for i in range(n_blocks):
        patches = transformer_encoder(sd, f"{prefix}.global_rep.{i}", patches, num_heads, drop=drop)
    patches = layer_norm(sd, f"{prefix}.global_rep.{n_blocks}", patches)
    if taps is not None:
        taps[prefix + ".patches"] = patches
    fm = folding(patches, info, ph, pw)
    fm = conv_bn_act(sd, prefix + ".conv_proj", fm, training=training, bn_state=bn_state)
    fm = conv_bn_act(sd, prefix + ".fusion", torch.cat((res, fm), dim=1), training=training, bn_state=bn_state)
    return fm


def mobilevit_block_temporal(sd, prefix, x: Tensor, x_prev: Optional[Tensor], n_blocks: int, num_heads: int, training: bool,
                            bn_state, ph: int = 2, pw: int = 2) -> Tuple[Tensor, Tensor]:
    """MobileViTBlock.forward_temporal  (cvnets/modules/mobilevit_block.py:289-314): forward_spatial with every TransformerEncoder reading
    its keys / values from `x_prev` (the patches [B*P, N, d] returned for the previous frame; None = self-attention), returning the
    patches after the last global layer beside the feature map.  Pinned on the reference by oracle/make_temporal_fixture.py."""
    res = x
    pre = prefix + "." if prefix else ""  # "" = the state dict of a bare block
    fm = conv_bn_act(sd, pre + "local_rep.conv_3x3", x, training=training, bn_state=bn_state)
    fm = conv_bn_act(sd, pre + "local_rep.conv_1x1", fm, use_norm=False, use_act=False)
    patches, info = unfolding(fm, ph, pw)
    for i in range(n_blocks):
        patches = transformer_encoder(sd, f"{pre}global_rep.{i}", patches, num_heads, x_prev=x_prev)
    patches = layer_norm(sd, f"{pre}global_rep.{n_blocks}", patches)
    fm = folding(patches, info, ph, pw)
    fm = conv_bn_act(sd, pre + "conv_proj", fm, training=training, bn_state=bn_state)
    fm = conv_bn_act(sd, pre + "fusion", torch.cat((res, fm), dim=1), training=training, bn_state=bn_state)
    return fm, patches


def mobilevit_forward(sd: Dict[str, Tensor], x: Tensor, mode: str = "small", num_heads: int = 4,
                      training: bool = True, bn_state: Optional[BNState] = None,
                      taps: Optional[Dict] = None, drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MobileViT.forward -> BaseImageEncoder.forward_classifier/extract_features
    (cvnets/models/classification/mobilevit.py:26-125; base_image_encoder.py:261-283);
    classifier = GlobalPool(mean) -> [Dropout] -> LinearLayer (mobilevit.py:106-119, global_pool.py:60-71).
    Dropout layers are p=0 (parity configuration) unless `drop` supplies the factors every Dropout layer multiplies by: see
    transformer_encoder for the keys of the transformer layers, "classifier" for mobilevit.py:110-113."""
    cfg = mobilevit_config(mode)
    exp = cfg["exp"]
    x = conv_bn_act(sd, "conv_1", x, stride=2, training=training, bn_state=bn_state)
    if taps is not None:
        taps["conv_1"] = x
    cin = 16
    for li, name in ((1, "layer1"), (2, "layer2")):
        c = cfg[name]
        for i in range(c["blocks"]):
            st = c["stride"] if i == 0 else 1
            x = inverted_residual(sd, f"layer_{li}.{i}", x, cin, c["out"], st, exp, training, bn_state)
            cin = c["out"]
        if taps is not None:
            taps[f"layer_{li}"] = x
    for li, name in ((3, "layer3"), (4, "layer4"), (5, "layer5")):
        c = cfg[name]
        x = inverted_residual(sd, f"layer_{li}.0", x, cin, c["out"], 2, exp, training, bn_state)
        cin = c["out"]
        x = mobilevit_block(sd, f"layer_{li}.1", x, c["nblk"], num_heads, training, bn_state, taps=taps, drop=drop)
        if taps is not None:
            taps[f"layer_{li}"] = x
    x = conv_bn_act(sd, "conv_1x1_exp", x, training=training, bn_state=bn_state)
    x = torch.mean(x, dim=[-2, -1])
    if drop is not None and "classifier" in drop:
        x = x * drop["classifier"]
    return F.linear(x, sd["classifier.fc.weight"], sd["classifier.fc.bias"])


def cross_entropy(logits: Tensor, target: Tensor, label_smoothing: float = 0.1) -> Tensor:
    """loss_fn/classification/cross_entropy.py:65-92 (training branch: F.cross_entropy with
    label smoothing, mean reduction)."""
    return F.cross_entropy(logits, target, label_smoothing=label_smoothing)


def train_step(sd: Dict[str, Tensor], x: Tensor, y: Tensor, mode: str = "small",
               label_smoothing: float = 0.1, drop: Optional[Dict[str, Tensor]] = None):
    """One fwd + loss + bwd of the hot path (engine/training_engine.py:257-287 without the
    optimizer).  Returns (logits, loss, grads-by-name, updated BN running stats)."""
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
              for k, v in sd.items()}
    st = BNState()
    logits = mobilevit_forward(params, x, mode=mode, training=True, bn_state=st, drop=drop)
    loss = cross_entropy(logits, y, label_smoothing)
    names = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [params[k] for k in names])
    return logits.detach(), loss.detach(), dict(zip(names, grads)), st.running


# --------------------------------------------------------------------------------------
# ViT  (cvnets/models/classification/vit.py:33-649)
# --------------------------------------------------------------------------------------
VIT_CFG = {"tiny": (192, 12, 3), "small": (384, 12, 6), "base": (768, 12, 12)}  # config/vit.py:12-99: (embed, layers, heads); ffn = 4*embed


def positional_embedding(sd, prefix: str, seq_len: int) -> Tensor:
    """LearnablePositionalEmbedding.forward (cvnets/layers/positional_embedding.py:81-104): bilinear resize of the
    [1,1,N,E] table along the sequence axis when seq_len != num_embeddings."""
    pe = sd[prefix + ".pos_embed.pos_embed"]
    n, e = pe.shape[2], pe.shape[3]
    if seq_len != n:
        pe = F.interpolate(pe, size=(seq_len, e), mode="bilinear")
    return pe.reshape(1, seq_len, e)


def vit_forward(sd: Dict[str, Tensor], x: Tensor, mode: str = "tiny", training: bool = True,
                bn_state: Optional[BNState] = None, prefix: str = "", head: str = "classifier") -> Tensor:
    """VisionTransformer.forward -> forward_classifier -> extract_features -> _features_from_transformer ->
    extract_patch_embeddings (vit.py:480-610), dropout p = 0, default (batch-first) MHA."""
    e, n_layers, heads = VIT_CFG[mode]
    p = prefix
    y = conv_bn_act(sd, p + "patch_emb.0", x, stride=4, training=training, bn_state=bn_state, act="gelu")
    y = conv_bn_act(sd, p + "patch_emb.1", y, stride=2, training=training, bn_state=bn_state, act="gelu")
    y = conv_bn_act(sd, p + "patch_emb.2", y, stride=2, use_norm=False, use_act=False)
    b = y.shape[0]
    t = y.flatten(2).transpose(1, 2).contiguous()
    t = positional_embedding(sd, p + "pos_embed", t.shape[1]).to(t.dtype) + t
    t = torch.cat((sd[p + "cls_token"].expand(b, -1, -1), t), dim=1)
    for i in range(n_layers):
        t = transformer_encoder(sd, f"{p}transformer.{i}", t, heads, act="gelu", ln_eps=1e-6)
    t = F.layer_norm(t, (e,), sd[p + "post_transformer_norm.weight"], sd[p + "post_transformer_norm.bias"], 1e-6)
    if head == "projection":  # SimpleImageProjectionHead.forward (cvnets/image_projection_layers/simple_projection_head.py:74-85)
        return F.normalize(t[:, 0] @ sd[p + "classifier.proj"], dim=-1)
    return F.linear(t[:, 0], sd[p + "classifier.weight"], sd[p + "classifier.bias"])


def generic_train_step(forward_fn, sd: Dict[str, Tensor], x: Tensor, y: Tensor, label_smoothing: float = 0.1, **kw):
    """fwd + label-smoothed CE + bwd for any of the functional models above; returns (logits, loss, grads, BN running stats)."""
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    st = BNState()
    logits = forward_fn(params, x, training=True, bn_state=st, **kw)
    loss = cross_entropy(logits, y, label_smoothing)
    names = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(params[k]) for g, k in zip(grads, names)]
    return logits.detach(), loss.detach(), dict(zip(names, grads)), st.running


# --------------------------------------------------------------------------------------
# MobileViTv2  (cvnets/models/classification/mobilevit_v2.py:20-226, config/mobilevit_v2.py:11-77)
# --------------------------------------------------------------------------------------
def mobilevit_v2_config(wm: float) -> Dict:
    l0 = int(make_divisible(min(max(32 * wm, 16), 64), divisor=8, min_value=16))
    return {"layer0": l0,
            "layer1": dict(out=int(make_divisible(64 * wm, divisor=16)), blocks=1, stride=1),
            "layer2": dict(out=int(make_divisible(128 * wm, divisor=8)), blocks=2, stride=2),
            "layer3": dict(out=int(make_divisible(256 * wm, divisor=8)), attn=int(make_divisible(128 * wm, divisor=8)), nblk=2),
            "layer4": dict(out=int(make_divisible(384 * wm, divisor=8)), attn=int(make_divisible(192 * wm, divisor=8)), nblk=4),
            "layer5": dict(out=int(make_divisible(512 * wm, divisor=8)), attn=int(make_divisible(256 * wm, divisor=8)), nblk=3),
            "exp": 2}


def group_norm1(sd, prefix: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm2D_NCHW = nn.GroupNorm(num_groups=1)  (cvnets/layers/normalization/layer_norm.py:75-108)."""
    return F.group_norm(x, 1, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def linear_self_attention(sd, prefix: str, x: Tensor) -> Tensor:
    """LinearSelfAttention._forward_self_attn on the unfolded [B, C, P, N] tensor (cvnets/layers/linear_attention.py:147-162)."""
    c = x.shape[1]
    qkv = F.conv2d(x, sd[prefix + ".qkv_proj.block.conv.weight"], sd[prefix + ".qkv_proj.block.conv.bias"])
    query, key, value = torch.split(qkv, [1, c, c], dim=1)
    scores = F.softmax(query, dim=-1)
    context = torch.sum(key * scores, dim=-1, keepdim=True)
    out = F.relu(value) * context.expand_as(value)
    return F.conv2d(out, sd[prefix + ".out_proj.block.conv.weight"], sd[prefix + ".out_proj.block.conv.bias"])


def linear_attn_ffn(sd, prefix: str, x: Tensor) -> Tensor:
    """LinearAttnFFN.forward, dropout p = 0  (cvnets/modules/transformer.py:248-264)."""
    x = x + linear_self_attention(sd, prefix + ".pre_norm_attn.1", group_norm1(sd, prefix + ".pre_norm_attn.0", x))
    y = group_norm1(sd, prefix + ".pre_norm_ffn.0", x)
    y = F.silu(F.conv2d(y, sd[prefix + ".pre_norm_ffn.1.block.conv.weight"], sd[prefix + ".pre_norm_ffn.1.block.conv.bias"]))
    y = F.conv2d(y, sd[prefix + ".pre_norm_ffn.3.block.conv.weight"], sd[prefix + ".pre_norm_ffn.3.block.conv.bias"])
    return x + y


def mobilevit_block_v2(sd, prefix: str, x: Tensor, n_blocks: int, training: bool, bn_state, ph: int = 2, pw: int = 2) -> Tensor:
    """MobileViTBlockv2.forward_spatial with unfolding_pytorch / folding_pytorch (cvnets/modules/mobilevit_block.py:526-626)."""
    b, c_in, h, w = x.shape
    if h % ph != 0 or w % pw != 0:  # resize_input_if_needed (:595-603)
        x = F.interpolate(x, size=(int(math.ceil(h / ph) * ph), int(math.ceil(w / pw) * pw)), mode="bilinear", align_corners=True)
    fm = conv_bn_act(sd, prefix + ".local_rep.0", x, groups=c_in, training=training, bn_state=bn_state)
    fm = conv_bn_act(sd, prefix + ".local_rep.1", fm, use_norm=False, use_act=False)
    bb, c, hh, ww = fm.shape
    patches = F.unfold(fm, kernel_size=(ph, pw), stride=(ph, pw)).reshape(bb, c, ph * pw, -1)
    for i in range(n_blocks):
        patches = linear_attn_ffn(sd, f"{prefix}.global_rep.{i}", patches)
    patches = group_norm1(sd, f"{prefix}.global_rep.{n_blocks}", patches)
    fm = F.fold(patches.reshape(bb, c * ph * pw, -1), output_size=(hh, ww), kernel_size=(ph, pw), stride=(ph, pw))
    return conv_bn_act(sd, prefix + ".conv_proj", fm, use_act=False, training=training, bn_state=bn_state)


def mobilevit_v2_forward(sd: Dict[str, Tensor], x: Tensor, width_multiplier: float = 1.0, training: bool = True,
                         bn_state: Optional[BNState] = None) -> Tensor:
    """MobileViTv2.forward (mobilevit_v2.py:29-96 + base_image_encoder.py:261-283); conv_1x1_exp is Identity,
    classifier = GlobalPool(mean) -> LinearLayer."""
    cfg = mobilevit_v2_config(width_multiplier)
    x = conv_bn_act(sd, "conv_1", x, stride=2, training=training, bn_state=bn_state)
    cin = cfg["layer0"]
    for li in (1, 2):
        c = cfg[f"layer{li}"]
        for i in range(c["blocks"]):
            x = inverted_residual(sd, f"layer_{li}.{i}", x, cin, c["out"], c["stride"] if i == 0 else 1, cfg["exp"], training, bn_state)
            cin = c["out"]
    for li in (3, 4, 5):
        c = cfg[f"layer{li}"]
        x = inverted_residual(sd, f"layer_{li}.0", x, cin, c["out"], 2, cfg["exp"], training, bn_state)
        cin = c["out"]
        x = mobilevit_block_v2(sd, f"layer_{li}.1", x, c["nblk"], training, bn_state)
    x = torch.mean(x, dim=[-2, -1])
    return F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])


# --------------------------------------------------------------------------------------
# CLIP  (cvnets/models/multi_modal_img_text/clip.py:144-210; cvnets/text_encoders/transformer.py:321-426;
#        loss_fn/multi_modal_img_text/contrastive_loss_clip.py:35-103)
# --------------------------------------------------------------------------------------
def text_transformer_forward(sd: Dict[str, Tensor], prefix: str, tokens: Tensor, n_layers: int, heads: int, causal: bool = True,
                             padding_idx: Optional[int] = 0) -> Tensor:
    """TextTransformer.forward -> encode_text (transformer.py:354-426): token + positional embedding (position `padding_idx` of the
    table is re-zeroed in place on every call, positional_embedding.py:84-86), causal pre-norm GELU transformer, final LayerNorm,
    EOT (= arg-max token id) row, projection, L2 normalisation."""
    emb = F.embedding(tokens, sd[prefix + "embedding_layer.weight"], padding_idx)
    pe = sd[prefix + "positional_embedding.pos_embed.pos_embed"]
    if padding_idx is not None:
        with torch.no_grad():
            pe[:, :, padding_idx, ...] = 0.0
    s = tokens.shape[1]
    if s != pe.shape[2]:
        pe = F.interpolate(pe, size=(s, pe.shape[3]), mode="bilinear")
    x = emb + pe.reshape(1, s, -1).to(emb.dtype)
    mask = None
    if causal:
        mask = torch.full((s, s), float("-inf"), device=tokens.device).triu_(1).unsqueeze(0).expand(tokens.shape[0], -1, -1)
    for i in range(n_layers):
        x = transformer_encoder(sd, f"{prefix}transformer.{i}", x, heads, act="gelu", attn_mask=mask)
    x = F.layer_norm(x, (x.shape[-1],), sd[prefix + "final_layer_norm.weight"], sd[prefix + "final_layer_norm.bias"], 1e-5)
    x = x[torch.arange(tokens.shape[0], device=tokens.device), tokens.argmax(dim=-1)]
    return F.normalize(x @ sd[prefix + "projection_layer"], dim=-1)


def clip_forward(sd: Dict[str, Tensor], image: Tensor, tokens: Tensor, vit_mode: str, text_layers: int, text_heads: int,
                 training: bool = True, bn_state: Optional[BNState] = None):
    """CLIP.forward (clip.py:144-210), training branch: (image embeddings, text embeddings, clamped exp(logit_scale))."""
    img = vit_forward(sd, image, mode=vit_mode, training=training, bn_state=bn_state, prefix="image_encoder.", head="projection")
    txt = text_transformer_forward(sd, "text_encoder.", tokens, text_layers, text_heads)
    return img, txt, torch.clamp(sd["logit_scale"].exp(), 0, 100.0)


def contrastive_loss_clip(img: Tensor, txt: Tensor, logit_scale: Tensor, all_img: Optional[Tensor] = None, all_txt: Optional[Tensor] = None,
                          rank: int = 0):
    """ContrastiveLossClip._forward_clip (contrastive_loss_clip.py:35-103); all_* = features gathered over ranks (None: 1 rank)."""
    all_img = img if all_img is None else all_img
    all_txt = txt if all_txt is None else all_txt
    logits_per_image = logit_scale * (img @ all_txt.transpose(0, 1))
    logits_per_text = logit_scale * (txt @ all_img.transpose(0, 1))
    n = logits_per_image.shape[0]
    labels = torch.arange(n, dtype=torch.long, device=img.device) + n * rank
    text_loss = F.cross_entropy(logits_per_text, labels) * 0.5
    image_loss = F.cross_entropy(logits_per_image, labels) * 0.5
    return image_loss + text_loss, image_loss, text_loss


def clip_train_step(sd: Dict[str, Tensor], image: Tensor, tokens: Tensor, **kw):
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    st = BNState()
    img, txt, scale = clip_forward(params, image, tokens, training=True, bn_state=st, **kw)
    loss, _, _ = contrastive_loss_clip(img, txt, scale)
    names = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(params[k]) for g, k in zip(grads, names)]
    return img.detach(), txt.detach(), loss.detach(), dict(zip(names, grads)), st.running
