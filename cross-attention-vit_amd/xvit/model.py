"""Reference module name -> xvit implementation (`from xvit.model import Encoder`)."""
from .encoder import Block, Encoder, Mlp, MultiHeadAttention  # noqa: F401
