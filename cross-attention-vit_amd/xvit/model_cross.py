"""Reference module name -> xvit implementation (`from xvit.model_cross import ModelCross`)."""
from .cross_vit import (Attention, CrossAttention, CrossAttentionBlock, FeedForward, ModelCross,  # noqa: F401
                        MultiScaleBlock, PreNorm, SelfAttentionBlock)
