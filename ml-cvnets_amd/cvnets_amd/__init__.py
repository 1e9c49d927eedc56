"""cvnets_amd — MI355X-native implementation of the CVNets backbone forward/backward hot path.

Host side: Python mirrors of cvnets.layers / cvnets.modules / MobileViT with identical signatures and
state_dict keys; compute: hand-written HIP kernels for gfx950 in libcvnets_hip.so (C ABI: include/cvnets_hip.h).
"""
from . import _lib, ops  # noqa: F401
from .layers import (BatchNorm2d, Conv2d, ConvLayer2d, Dropout, GELU, GlobalPool, Identity, LayerNorm, LinearLayer,  # noqa: F401
                     MultiHeadAttention, PositionalEmbedding, Swish, build_activation_layer, default_opts, get_normalization_layer)
from .models import MobileViT, VisionTransformer, build_mobilevit, build_vit, get_configuration  # noqa: F401
from .modules import InvertedResidual, MobileViTBlock, TransformerEncoder  # noqa: F401
from .ops import compute_dtype, set_compute_dtype  # noqa: F401

__version__ = "0.1.0"
