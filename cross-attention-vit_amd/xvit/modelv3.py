"""Reference module name -> xvit implementation (`from xvit.modelv3 import ModelVIT`)."""
from .vit import ModelVIT, Transformer  # noqa: F401
