"""Host-side mirror of the CLIP path: ``TextTransformer`` (cvnets/text_encoders/transformer.py:23-500),
``SimpleImageProjectionHead`` (cvnets/image_projection_layers/simple_projection_head.py:17-85), ``CLIP``
(cvnets/models/multi_modal_img_text/clip.py:24-228) and ``ContrastiveLossClip``
(loss_fn/multi_modal_img_text/contrastive_loss_clip.py:20-172) — same constructor signatures, attribute trees and state_dict keys;
forward/backward run the HIP kernels (token embedding, causal fused attention, GEMMs, LayerNorm, EOT gather, L2 normalise); the losses live in losses.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
from torch import Tensor, nn

from . import ops
from .layers import Dropout, Embedding, PositionalEmbedding, get_normalization_layer, opt
from .modules import TransformerEncoder


class TextTransformer(nn.Module):
    def __init__(self, opts, projection_dim: int, *args, **kwargs) -> None:
        model_dim = opt(opts, "model.text.transformer.model_dim", 512)
        no_scale_embedding = opt(opts, "model.text.transformer.no_scale_embedding", False)
        no_pos_embedding = opt(opts, "model.text.transformer.no_pos_embedding", False)
        embed_dropout = opt(opts, "model.text.transformer.embed_dropout", 0.0)
        dropout = opt(opts, "model.text.transformer.dropout", 0.0)
        attn_dropout = opt(opts, "model.text.transformer.attn_dropout", 0.0)
        ffn_dropout = opt(opts, "model.text.transformer.ffn_dropout", 0.0)
        norm_layer = opt(opts, "model.text.transformer.norm_layer", None)
        if norm_layer is None:
            raise ValueError("Normalization layer can not be None in {}".format(self.__class__.__name__))
        vocab_size = opt(opts, "dataset.text_vocab_size", None)
        if opt(opts, "common.debug_mode", False):
            vocab_size = 100
        if vocab_size is None:
            raise ValueError("Vocabulary size can't be None or -1 in {}".format(self.__class__.__name__))
        super().__init__()
        self.opts = opts
        self.projection_dim = projection_dim
        self.is_master_node = True
        self.vocab_size = vocab_size
        padding_index = opt(opts, "dataset.padding_index", None)
        self.embedding_layer = Embedding(opts=opts, embedding_dim=model_dim, padding_idx=padding_index, num_embeddings=self.vocab_size)
        self.embed_scale = 1.0 if no_scale_embedding else model_dim ** -0.5
        context_length = opt(opts, "dataset.text_context_length", None)
        if opt(opts, "common.debug_mode", False):
            context_length = 77
        assert context_length is not None, "Context length can't be None. Please set dataset.text_context_length"
        self.positional_embedding = None if no_pos_embedding else PositionalEmbedding(
            opts=opts, num_embeddings=context_length, embedding_dim=model_dim, padding_idx=padding_index,
            is_learnable=not opt(opts, "model.text.transformer.sinusoidal_pos_emb", False))
        self.embedding_dropout = Dropout(p=embed_dropout)
        n_layers = opt(opts, "model.text.transformer.n_transformer_layers", 6)
        ffn_multipliers = opt(opts, "model.text.transformer.ffn_multiplier_per_layer", 4.0)
        if isinstance(ffn_multipliers, (float, int)):
            ffn_multipliers = [ffn_multipliers] * n_layers
        if not isinstance(ffn_multipliers, Sequence) or len(ffn_multipliers) != n_layers:
            raise ValueError("We need one FFN multiplier per transformer layer")
        ffn_dims = [int(math.ceil(model_dim * m / 16.0) * 16.0) for m in ffn_multipliers]
        mha_heads = opt(opts, "model.text.transformer.n_heads_per_layer", 8)
        if isinstance(mha_heads, int):
            mha_heads = [mha_heads] * n_layers
        if not isinstance(mha_heads, Sequence) or len(mha_heads) != n_layers:
            raise ValueError("We need the number of MHA heads for each transformer layer")
        self.transformer = nn.ModuleList([
            TransformerEncoder(opts=opts, embed_dim=model_dim, num_heads=mha_heads[i], ffn_latent_dim=ffn_dims[i], attn_dropout=attn_dropout,
                               ffn_dropout=ffn_dropout, dropout=dropout, transformer_norm_layer=norm_layer) for i in range(n_layers)])
        self.final_layer_norm = get_normalization_layer(opts, num_features=model_dim, norm_type=norm_layer)
        self.projection_layer = nn.Parameter(torch.empty(model_dim, self.projection_dim))
        self.model_dim = model_dim
        self.reset_parameters_clip_style()
        self.gradient_ckpt = opt(opts, "model.text.transformer.gradient_checkpoint", False)  # memory-only knob: nothing to do with 288 GB
        self.use_pytorch_mha = False
        self.causal_masking = opt(opts, "model.text.transformer.causal_masking", False)
        self.classes_per_split_zero_shot = max(1, int(opt(opts, "model.text.transformer.classes_per_split_zero_shot", 1)))

    def reset_parameters_clip_style(self):
        nn.init.normal_(self.embedding_layer.weight, mean=0.0, std=0.02)
        attn_std = self.model_dim ** -0.5
        proj_std = attn_std * ((2 * len(self.transformer)) ** -0.5)
        fc_std = (2 * self.model_dim) ** -0.5
        for block in self.transformer:
            nn.init.normal_(block.pre_norm_mha[1].qkv_proj.weight, mean=0.0, std=attn_std)
            nn.init.normal_(block.pre_norm_mha[1].out_proj.weight, mean=0.0, std=proj_std)
            nn.init.normal_(block.pre_norm_ffn[1].weight, mean=0.0, std=fc_std)
            nn.init.normal_(block.pre_norm_ffn[4].weight, mean=0.0, std=proj_std)
        nn.init.normal_(self.projection_layer, mean=0.0, std=attn_std)

    def forward_embedding(self, text_tokens: Tensor) -> Tensor:
        pos = self.positional_embedding.table(text_tokens.shape[1]) if self.positional_embedding is not None else None
        token_emb = self.embedding_layer(text_tokens, pos=pos)  # lookup + positional add in one kernel
        return self.embedding_dropout(token_emb)

    def encode_text(self, text_tokens: Tensor, key_padding_mask: Optional[Tensor] = None, return_all_tokens: bool = False, *args,
                    **kwargs) -> Tensor:
        B, S = text_tokens.shape
        t = self.forward_embedding(text_tokens).view(B * S, self.model_dim)
        causal = bool(self.causal_masking)
        if causal:
            key_padding_mask = None  # text_encoders/transformer.py:376-380
        seqmap = (B, S, 1, 1, S, 1, S)
        for layer in self.transformer:
            t = layer.forward_tokens(t, seqmap, causal=causal, key_padding_mask=key_padding_mask)
        n = self.final_layer_norm
        t = ops.layer_norm_tokens(t, n, seqmap)
        if return_all_tokens:
            return t.view(B, S, self.model_dim)
        rows = torch.arange(B, device=text_tokens.device) * S + text_tokens.argmax(dim=-1)  # plumbing: EOT index arithmetic
        eot = ops.RowsGatherIdx.apply(t, rows)
        emb = ops.linear(eot, self.projection_layer.t())  # x @ P
        return ops.l2_normalize(emb)

    def forward_zero_shot(self, text_tokens: Tensor, key_padding_mask: Optional[Tensor] = None, *args, **kwargs) -> Tensor:
        raise NotImplementedError("zero-shot class-template averaging is an evaluation utility, not the HIP training hot path")

    def forward(self, text_tokens: Tensor, key_padding_mask: Optional[Tensor] = None, *args, **kwargs) -> Tensor:
        if text_tokens.dim() == 2:
            return self.encode_text(text_tokens=text_tokens, key_padding_mask=key_padding_mask, *args, **kwargs)
        if text_tokens.dim() == 3:
            b, n, _ = text_tokens.shape
            kpm = key_padding_mask.reshape(b * n, -1) if key_padding_mask is not None else None
            return self.encode_text(text_tokens=text_tokens.reshape(b * n, -1), key_padding_mask=kpm, *args, **kwargs).reshape(b, n, -1)
        if text_tokens.dim() == 4:
            return self.forward_zero_shot(text_tokens, key_padding_mask)
        raise NotImplementedError


class SimpleImageProjectionHead(nn.Module):
    def __init__(self, opts, in_dim: int, out_dim: int, *args, **kwargs) -> None:
        super().__init__()
        scale = in_dim ** -0.5
        self.use_identity = bool(opt(opts, "model.image_projection_head.simple_projection_nc2nc.identity_if_same_size", False)) and in_dim == out_dim
        if not self.use_identity:
            self.proj = nn.Parameter(scale * torch.randn(size=(in_dim, out_dim)))
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.feature_normalization = not opt(opts, "model.image_projection_head.simple_projection_nc2nc.no_feature_normalization", False)

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        assert x.dim() == 2, "Input should be 2-dimensional (Batch x in_dim). Got: {}".format(x.shape)
        if not self.use_identity:
            x = ops.linear(x.contiguous(), self.proj.t())  # x @ proj
        if self.feature_normalization:
            x = ops.l2_normalize(x)
        return x


class CLIP(nn.Module):
    def __init__(self, opts, image_encoder: nn.Module, text_encoder: nn.Module, *args, **kwargs) -> None:
        super().__init__()
        self.lr_multiplier_img_encoder = opt(opts, "model.multi_modal_image_text.lr_multiplier_img_encoder", 1.0)
        self.lr_multiplier_text_encoder = opt(opts, "model.multi_modal_image_text.lr_multiplier_text_encoder", 1.0)
        self.image_encoder = image_encoder
        self.text_encoder = text_encoder
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1.0 / 0.07))
        self.use_distributed = opt(opts, "ddp.use_distributed", False)
        self.cache_text_features_zero_shot = opt(opts, "model.multi_modal_image_text.clip.cache_text_features_zero_shot", False)
        self.cached_text_features = None

    def _exponentiate_and_clip_logits(self, max_scale: float = 100.0):
        return torch.clamp(self.logit_scale.exp(), 0, max_scale)  # plumbing: one learnable scalar

    def forward(self, input: Dict, *args, **kwargs) -> Dict:
        images, text_tokens, padding_mask = input.get("image", None), input.get("text", None), input.get("padding_mask", None)
        image_embeddings = self.image_encoder(images)
        if not isinstance(image_embeddings, Tensor):
            raise NotImplementedError("dict outputs of the image encoder (neural augmentation) are not on the HIP hot path")
        if text_tokens.dim() == 4:
            raise NotImplementedError("zero-shot evaluation is not on the HIP training hot path")
        ops.pack_all(self.text_encoder)  # the image encoder packs its own weights; without this the text tower would run on the packs of
        #                                  whatever step its per-layer cache was filled in (the fused optimizer writes through raw pointers)
        text_embeddings = self.text_encoder(text_tokens=text_tokens, key_padding_mask=padding_mask)
        return {"image": image_embeddings, "text": text_embeddings, "logit_scale": self._exponentiate_and_clip_logits(),
                "zero_shot_image_logits": None, "augmented_tensor": None}

    @classmethod
    def build_model(cls, opts, *args, **kwargs) -> "CLIP":
        from .models import VisionTransformer

        projection_dim = opt(opts, "model.multi_modal_image_text.clip.projection_dim", -1)
        if projection_dim < 1:
            raise ValueError("Projection dimension should be > 1. Got: {}".format(projection_dim))
        if opt(opts, "model.classification.name", "vit") != "vit":
            raise NotImplementedError("CLIP image encoders other than ViT are not on the HIP hot path")
        image_encoder = VisionTransformer(opts)
        text_encoder = TextTransformer(opts, projection_dim=projection_dim)
        image_encoder.classifier = SimpleImageProjectionHead(opts, in_dim=image_encoder.classifier.in_features, out_dim=projection_dim)
        return cls(opts, image_encoder=image_encoder, text_encoder=text_encoder)


def build_clip(opts=None, **overrides) -> CLIP:
    from .layers import default_opts

    if opts is None:
        base = {"model.classification.name": "vit", "model.classification.vit.mode": "base", "model.classification.vit.norm_layer": "layer_norm_fp32",
                "model.activation.name": "gelu", "model.multi_modal_image_text.clip.projection_dim": 512,
                "model.text.transformer.causal_masking": True, "model.text.transformer.model_dim": 512,
                "model.text.transformer.n_transformer_layers": 12, "model.text.transformer.ffn_multiplier_per_layer": 4.0,
                "model.text.transformer.n_heads_per_layer": 8, "model.text.transformer.norm_layer": "layer_norm_fp32",
                "dataset.text_vocab_size": 49408, "dataset.text_context_length": 77, "dataset.padding_index": 0}
        base.update(overrides)
        opts = default_opts(**base)
    return CLIP.build_model(opts)
