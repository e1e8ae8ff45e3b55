"""xvit — MI355X-native hot path of the cross-attention 3-D ViT, behind the reference's
own nn.Module signatures (vsahni3/cross-attention-ViT: model_cross.py, model.py)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
