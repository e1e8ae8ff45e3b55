"""Drop-in replacement for the reference's modelv3.ModelVIT — its own comparison arm in every run
(main_mist.py:158-159): the patch tokens of ALL modalities concatenated into one sequence (N = M*P + 1)
through one stack of the same pre-norm blocks as model_cross (modelv3.py:18-67 == model_cross.py:11-61).
Same constructor (`config.num_layers` deep), forward signature and state_dict keys
(`transformer.layers.{l}.{0,2}.*`, `mlp_head.{0,1,4}.*`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as XF
from .cross_vit import Attention, FeedForward, PreNorm, _Base, _lin, _p
from .metrics import EpochStatsMixin


class _NoDrop(nn.Module):
    """Placeholder for torchvision.ops.StochasticDepth(p=0, mode="row") (modelv3.py:74-82): the reference
    hard-codes every rate to 0, which makes it the identity; kept so the child indices 0..3 match."""

    def forward(self, x):
        return x


class Transformer(nn.Module):
    """modelv3.py:68-87: layers[l] = [PreNorm(Attention), drop-path(0), PreNorm(FeedForward), drop-path(0)]."""

    def __init__(self, config):
        super().__init__()
        dh = config.hidden_dim // config.num_heads
        self.layers = nn.ModuleList(
            nn.ModuleList([PreNorm(config, Attention(config, dim_head=dh)), _NoDrop(), PreNorm(config, FeedForward(config)), _NoDrop()])
            for _ in range(config.num_layers))

    def forward(self, x):
        for attn, _, ff, _ in self.layers:
            x = XF.SelfAttentionBlockFn.apply(
                x, attn.norm.weight, attn.norm.bias, attn.fn.to_qkv.weight, attn.fn.to_out[0].weight, attn.fn.to_out[0].bias,
                ff.norm.weight, ff.norm.bias, ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias,
                attn.fn.heads, attn.norm.eps, _p(self, ff.fn.net[2]))
        return x


class ModelVIT(EpochStatsMixin, _Base):
    """modelv3.py:90-147.  forward(img [B, M, 1, D, H, W], labels [B]) -> (logits, loss)."""

    def __init__(self, config):
        super().__init__()
        if any(s % p for s, p in zip(config.img_size, config.patch_size)):
            raise AssertionError('image dimensions must be divisible by the patch size')
        P = 1
        for s, p in zip(config.img_size, config.patch_size):
            P *= s // p
        pd = config.patch_size[0] * config.patch_size[1] * config.patch_size[2]
        d = config.hidden_dim
        self.patch_size = tuple(config.patch_size)
        for k in ("lr", "weight_decay", "optim_params"):
            setattr(self, k, getattr(config, k))
        self.pos_embedding = nn.Parameter(torch.empty(1, P * config.num_modalities + 1, d))
        self.patch_to_embedding = _lin(pd, d)
        self.cls_token = nn.Parameter(torch.empty(1, 1, d))
        self.dropout = nn.Dropout(config.dropout)
        self.transformer = Transformer(config)
        self.to_cls_token = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(d), _lin(d, config.mlp_dim), nn.GELU(), nn.Dropout(config.dropout),
                                      _lin(config.mlp_dim, config.num_classes), nn.Dropout(config.dropout))
        self.initialize_model()

    def forward(self, img, labels):
        if img.is_cuda and torch.is_grad_enabled():
            XF.arena_begin(img.device)                   # the backward's small zeroed vectors: one fill per step (functional._zeros)
        x = XF.PatchEmbedFn.apply(img, self.patch_to_embedding.weight, self.patch_to_embedding.bias, self.cls_token, self.pos_embedding,
                                  self.patch_size, _p(self, self.dropout), True)
        x = self.transformer(x)
        h = self.mlp_head
        logits_m = XF.HeadFn.apply(x, h[0].weight, h[0].bias, h[1].weight, h[1].bias, h[4].weight, h[4].bias, h[0].eps, _p(self, h[3]))
        return XF.MeanCrossEntropyFn.apply(logits_m.unsqueeze(0), labels, 0.0)

    @staticmethod
    def init_weights(module):
        if isinstance(module, nn.Linear):
            nn.init.xavier_uniform_(module.weight)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight)
            nn.init.zeros_(module.bias)

    def initialize_model(self):
        self.apply(ModelVIT.init_weights)
        nn.init.normal_(self.pos_embedding, mean=0.0, std=0.02)
        nn.init.normal_(self.cls_token, mean=0.0, std=0.02)

    def log(self, *args, **kwargs):
        sup = getattr(super(), "log", None)
        if sup is not None:
            return sup(*args, **kwargs)

    def training_step(self, batch, batch_idx):
        x, labels = batch
        logits, loss = self(x, labels)
        self.log('train_loss', loss, on_epoch=True, on_step=False, sync_dist=True)
        self.log_stats('train', logits, labels)
        return loss

    def validation_step(self, batch, batch_idx):
        x, labels = batch
        logits, loss = self(x, labels)
        self.log('val_loss', loss, on_epoch=True, on_step=False, sync_dist=True)
        self.log_stats('val', logits, labels)

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=self.optim_params["T_max"], eta_min=self.optim_params["eta_min"])
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "epoch"}}
