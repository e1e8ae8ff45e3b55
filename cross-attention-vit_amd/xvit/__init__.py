"""xvit — MI355X-native hot path of the cross-attention 3-D ViT, behind the reference's
own nn.Module signatures (vsahni3/cross-attention-ViT: model_cross.py, model.py).

    from xvit.model_cross import ModelCross          # drop-in for reference model_cross.ModelCross
    from xvit.modelv3 import ModelVIT                # drop-in for reference modelv3.ModelVIT (concatenated-token baseline)
    from xvit.model import Encoder                   # drop-in for reference model.Encoder
"""
from . import _lib  # noqa: F401
from .functional import invalidate_shadows  # noqa: F401
from .model import Block, Encoder, Mlp, MultiHeadAttention  # noqa: F401
from .modelv3 import ModelVIT, Transformer  # noqa: F401
from .model_cross import (Attention, CrossAttention, CrossAttentionBlock, FeedForward, ModelCross,  # noqa: F401
                          MultiScaleBlock, PreNorm, SelfAttentionBlock)
from .metrics import BinaryEpochMetrics  # noqa: F401
