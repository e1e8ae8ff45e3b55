"""Drop-in replacements for the encoder classes of the reference's model.py (:107-214):
Mlp, MultiHeadAttention, Block, Encoder — separate biased query/key/value/out projections,
scores divided by sqrt(head size), LayerNorm eps 1e-6.  Same constructor (`config.hidden_size`,
`config.transformer[...]`), forward signature and state_dict keys; arithmetic on the HIP kernels
(the three projections run as ONE GEMM against the row-concatenated bf16 weight copy)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as XF


def _p(module, drop):
    return float(drop.p) if module.training else 0.0


class Mlp(nn.Module):
    """model.py:107-122."""

    def __init__(self, config):
        super().__init__()
        t = config.transformer
        self.fc1 = nn.Linear(config.hidden_size, t["mlp_dim"])
        self.fc2 = nn.Linear(t["mlp_dim"], config.hidden_size)
        self.dropout = nn.Dropout(t["dropout_rate"])

    def forward(self, x):
        return XF.FeedForwardFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, _p(self, self.dropout))


class MultiHeadAttention(nn.Module):
    """model.py:124-178."""

    def __init__(self, config):
        super().__init__()
        t = config.transformer
        self.num_attention_heads = t["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        for name in ("query", "key", "value"):
            setattr(self, name, nn.Linear(config.hidden_size, self.all_head_size))
        self.out = nn.Linear(config.hidden_size, config.hidden_size)
        self.attn_dropout = nn.Dropout(t["attention_dropout_rate"])
        self.proj_dropout = nn.Dropout(t["attention_dropout_rate"])

    def forward(self, x):
        w = torch.cat((self.query.weight, self.key.weight, self.value.weight), dim=0)
        b = torch.cat((self.query.bias, self.key.bias, self.value.bias), dim=0)
        qkv = XF.LinearFn.apply(x, w, b, False)
        # model.py:169: dropout on the attention probabilities, inside the fused kernel (counter-hash mask, regenerated in backward)
        ctxl = XF.AttentionCoreFn.apply(qkv, self.num_attention_heads, 1.0 / float(self.attention_head_size) ** 0.5, _p(self, self.attn_dropout))
        return XF.LinearFn.apply(ctxl, self.out.weight, self.out.bias, True, _p(self, self.proj_dropout))


class Block(nn.Module):
    """model.py:181-201 — one fused autograd node (XF.EncoderBlockFn)."""

    def __init__(self, config):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.multi_head = MultiHeadAttention(config)
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn = Mlp(config)

    def forward(self, x):
        a, f = self.multi_head, self.ffn
        return XF.EncoderBlockFn.apply(
            x, self.attention_norm.weight, self.attention_norm.bias, a.query.weight, a.query.bias, a.key.weight, a.key.bias,
            a.value.weight, a.value.bias, a.out.weight, a.out.bias, self.ffn_norm.weight, self.ffn_norm.bias,
            f.fc1.weight, f.fc1.bias, f.fc2.weight, f.fc2.bias, a.num_attention_heads, self.attention_norm.eps,
            _p(self, a.proj_dropout), _p(self, f.dropout), _p(self, a.attn_dropout))


class Encoder(nn.Module):
    """model.py:203-214."""

    def __init__(self, config):
        super().__init__()
        self.encoder_norm = nn.LayerNorm(config.hidden_size, eps=1e-6)
        self.layers = nn.Sequential(*(Block(config) for _ in range(config.transformer["num_layers"])))

    def forward(self, x):
        x = self.layers(x)
        # fp32 out, like the reference's LayerNorm on an fp32 stream
        return XF.LayerNormFn.apply(x, self.encoder_norm.weight, self.encoder_norm.bias, self.encoder_norm.eps).float()
