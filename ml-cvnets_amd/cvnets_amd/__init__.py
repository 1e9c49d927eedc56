"""cvnets_amd — MI355X-native implementation of the CVNets backbone forward/backward hot path.

Host side: Python mirrors of cvnets.layers / cvnets.modules / MobileViT with identical signatures and
state_dict keys; compute: hand-written HIP kernels for gfx950 in libcvnets_hip.so (C ABI: include/cvnets_hip.h).
"""
from . import _lib, ops, optim  # noqa: F401
from .layers import (BatchNorm2d, Conv2d, ConvLayer2d, Dropout, GELU, GlobalPool, Identity, Embedding, LayerNorm, LayerNorm2D_NCHW,  # noqa: F401
                     LayerNormFP32, LinearLayer, LinearSelfAttention, MultiHeadAttention, PositionalEmbedding, Swish, build_activation_layer, default_opts, get_normalization_layer)
from .models import (MobileViT, MobileViTv2, VisionTransformer, build_mobilevit, build_mobilevit_v2, build_vit,  # noqa: F401
                     get_configuration)
from .clip import CLIP, SimpleImageProjectionHead, TextTransformer, build_clip  # noqa: F401
from .losses import ContrastiveLossClip, CrossEntropy  # noqa: F401
from .modules import InvertedResidual, LinearAttnFFN, MobileViTBlock, MobileViTBlockv2, TransformerEncoder  # noqa: F401
from .ops import compute_dtype, set_compute_dtype  # noqa: F401
from .detection import SeparableConv2d, SingleShotMaskDetector, SSDAnchorGenerator, SSDHead, build_ssd  # noqa: F401
from .segmentation import ASPP, PSP, DeeplabV3, PSPNet, SegEncoderDecoder, build_deeplabv3_mobilevit, build_segmentation  # noqa: F401

__version__ = "0.1.0"
