"""Drop-in replacements for the reference's model_cross.py modules (re-exported by
xvit/model_cross.py under the reference's module name).

Same class names, constructor arguments, forward signatures and state_dict keys as
/root/reference/model_cross.py; the arithmetic runs on the HIP kernels of
libxvit_hip.so.  Parameters are ordinary fp32 nn.Parameters (so torch.optim.Adam, DDP and
Lightning checkpoints work unchanged); nn.Linear / nn.LayerNorm objects are used ONLY as
parameter containers with the reference's names — their torch forward is never called.

`config` is duck-typed: any object with the attributes config2.py / main_mist.py set.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import functional as XF
from .metrics import EpochStatsMixin

try:  # the reference derives ModelCross from lightning.LightningModule (model_cross.py:152)
    import lightning as _L  # type: ignore

    _Base = _L.LightningModule
except Exception:  # lightning is not installed on this image: plain nn.Module, same methods
    _Base = nn.Module


def _lin(i, o, bias=True):
    return nn.Linear(i, o, bias=bias)


def _mlp_container(config, out_dim):
    """Parameter container with the reference's child indices: 0 = Linear(d, f), 3 = Linear(f, out)."""
    return nn.Sequential(_lin(config.hidden_dim, config.mlp_dim), nn.GELU(), nn.Dropout(config.dropout),
                         _lin(config.mlp_dim, out_dim), nn.Dropout(config.dropout))


def _p(module, drop):
    """Active dropout probability of an nn.Dropout child: its p in training mode, else 0 (nn.Dropout semantics).
    Masks come from a counter hash (xvit_dropout), so they cannot bit-match torch's Philox stream: parity is
    defined at p = 0 / eval; in training the layers are statistically equivalent."""
    return float(drop.p) if module.training else 0.0


class PreNorm(nn.Module):
    """model_cross.py:11-17."""

    def __init__(self, config, fn):
        super().__init__()
        self.norm = nn.LayerNorm(config.hidden_dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(XF.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps), **kwargs)


class FeedForward(nn.Module):
    """model_cross.py:19-31."""

    def __init__(self, config):
        super().__init__()
        self.net = _mlp_container(config, config.hidden_dim)

    def forward(self, x):
        return XF.FeedForwardFn.apply(x, self.net[0].weight, self.net[0].bias, self.net[3].weight, self.net[3].bias, _p(self, self.net[2]))


class Attention(nn.Module):
    """model_cross.py:33-61."""

    def __init__(self, config, dim_head):
        super().__init__()
        d, H = config.hidden_dim, config.num_heads
        if dim_head * H != d:
            raise AssertionError(f"dim_head * num_heads ({dim_head}*{H}) must equal hidden_dim ({d})")
        self.heads, self.scale = H, dim_head ** -0.5
        self.to_qkv = _lin(d, 3 * d, bias=False)                       # fused q|k|v, no bias
        single = H == 1 and dim_head == d
        self.to_out = nn.Identity() if single else nn.Sequential(_lin(d, d), nn.Dropout(config.dropout))

    def forward(self, x):
        qkv = XF.LinearFn.apply(x, self.to_qkv.weight, None, False)
        out = XF.AttentionCoreFn.apply(qkv, self.heads, self.scale)
        if isinstance(self.to_out, nn.Identity):
            return out
        return XF.LinearFn.apply(out, self.to_out[0].weight, self.to_out[0].bias, True, _p(self, self.to_out[1]))


class SelfAttentionBlock(nn.Module):
    """model_cross.py:64-72 — one fused autograd node (XF.SelfAttentionBlockFn)."""

    def __init__(self, config):
        super().__init__()
        self.attn = PreNorm(config, Attention(config, dim_head=(config.hidden_dim // config.num_heads)))
        self.ffn = PreNorm(config, FeedForward(config))
        self.feeds_a_block = False     # set by MultiScaleBlock for the blocks behind another one in a branch (XF.attach_b16)

    def forward(self, x):
        a, f = self.attn, self.ffn
        if isinstance(a.fn.to_out, nn.Identity):  # single-head degenerate case: unfused composition
            x = a(x) + x
            return f(x) + x
        return XF.SelfAttentionBlockFn.apply(
            x, a.norm.weight, a.norm.bias, a.fn.to_qkv.weight, a.fn.to_out[0].weight, a.fn.to_out[0].bias,
            f.norm.weight, f.norm.bias, f.fn.net[0].weight, f.fn.net[0].bias, f.fn.net[3].weight, f.fn.net[3].bias,
            a.fn.heads, a.norm.eps, _p(self, f.fn.net[2]), self.feeds_a_block and not self._forward_hooks and not self._forward_pre_hooks)


class CrossAttention(nn.Module):
    """model_cross.py:74-102: the query is the CLS row only."""

    def __init__(self, config):
        super().__init__()
        d = config.hidden_dim
        self.num_heads = config.num_heads
        self.scale = (d // self.num_heads) ** -0.5
        for name in ("wq", "wk", "wv", "proj"):
            setattr(self, name, _lin(d, d))
        self.attn_drop, self.proj_drop = nn.Dropout(config.dropout), nn.Dropout(config.dropout)

    def forward(self, x):
        B, N, C = x.shape
        q = XF.LinearFn.apply(x[:, 0], self.wq.weight, self.wq.bias, False)
        k = XF.LinearFn.apply(x, self.wk.weight, self.wk.bias, False)
        v = XF.LinearFn.apply(x, self.wv.weight, self.wv.bias, False)
        o = XF.ClsAttentionCoreFn.apply(q, torch.cat((k, v), dim=-1), self.num_heads, self.scale, _p(self, self.attn_drop))
        return XF.LinearFn.apply(o, self.proj.weight, self.proj.bias, True, _p(self, self.proj_drop)).reshape(B, 1, C)


def _fusion_args(blk):
    a, f = blk.attn, blk.ffn
    c = a.fn
    return (a.norm.weight, a.norm.bias, c.wq.weight, c.wq.bias, c.wk.weight, c.wk.bias, c.wv.weight, c.wv.bias,
            c.proj.weight, c.proj.bias, f.norm.weight, f.norm.bias, f.fn.net[0].weight, f.fn.net[0].bias,
            f.fn.net[3].weight, f.fn.net[3].bias, c.num_heads, a.norm.eps)


class CrossAttentionBlock(nn.Module):
    """model_cross.py:104-114: [B, N, d] -> [B, 1, d]."""

    def __init__(self, config, act_layer=nn.GELU):
        super().__init__()
        self.attn = PreNorm(config, CrossAttention(config))
        self.ffn = PreNorm(config, FeedForward(config))

    def forward(self, x):
        return XF.CrossFusionFn.apply(x, x, *_fusion_args(self), False, _p(self, self.attn.fn.attn_drop))


class _FanOut(torch.autograd.Function):
    """x -> n aliases of x whose gradients are summed HERE, on this node's stream.  A tensor consumed by nodes on different
    side streams otherwise has its gradients accumulated inside the autograd engine's input buffer, across streams (the
    engine's own event hand-off); under HIP-graph capture that cross-stream accumulation is what crashed
    hipStreamEndCapture.  With one gradient slot per consumer the engine only orders each producer stream before this
    node — the pattern the branch fork already uses."""

    @staticmethod
    def forward(ctx, x, n, cls_first=False, want_b16=False):
        """cls_first: reader 0 (the modality's own fusion in a cls-only block, which uses nothing but the CLS rows) gets x[:, :1].
        want_b16: x was produced by a SelfAttentionBlock: hand the summed gradient on in bf16 as well (XF.attach_b16)."""
        ctx.cls_first = cls_first
        ctx.want_b16 = bool(want_b16) and XF.B16_HANDOFF
        outs = [x.view_as(x) for _ in range(n)]
        if cls_first:
            outs[0] = x[:, :1]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        narrow = gs[0] if ctx.cls_first else None          # [B, 1, d]: joins the CLS rows only — no zero-filled full tensor, no full add
        gs = [g for g in (gs[1:] if ctx.cls_first else gs) if g is not None]
        if not gs:
            return None, None, None
        total, tb = gs[0], None
        if len(gs) == 2 and ctx.want_b16 and gs[0].is_cuda and gs[0].dtype == torch.float32 and gs[1].dtype == torch.float32 \
                and gs[0].shape == gs[1].shape and gs[0].is_contiguous() and gs[1].is_contiguous() and gs[0].numel() % 8 == 0:
            total, tb = XF.ops.add_cast(gs[0], gs[1])      # the sum and its bf16 copy in one pass (the reader is a block's backward)
        else:
            for g in gs[1:]:
                total = total + g
            if len(gs) == 1 and ctx.want_b16:
                tb = XF.b16_of(total, (total.shape[0] * total.shape[1], total.shape[2])) if total.dim() == 3 else None   # a fusion's dcat came in both dtypes
        if narrow is not None:
            if len(gs) == 1 and not total.is_contiguous():
                total, tb = total.contiguous(), None
            # in place: `total` is a fusion's freshly written dcat (this node is its only reader) or the sum above; the B rows of the bf16 copy follow
            if total.is_cuda and total.dtype == torch.float32 and narrow.dtype == torch.float32:
                XF.ops.rows_combine(total[:, 0], a=total[:, 0], b=narrow[:, 0], dst2=tb.view(total.shape)[:, 0] if tb is not None else None)
            else:
                total[:, :1] += narrow
                tb = None
        XF.keep(total, *gs, narrow)
        if tb is not None:
            total = XF.attach_b16(total, tb)
        return total, None, None, None


_SIDE_STREAMS: dict = {}
# per-context override of XVIT_STREAMS (xvit.graph.GraphedStep captures with its own mode without touching the process environment,
# which another thread running a model would see)
import contextvars  # noqa: E402

STREAM_MODE: "contextvars.ContextVar[str | None]" = contextvars.ContextVar("xvit_stream_mode", default=None)


def _side_streams(device, n, kind="branches"):
    """Per-device pool of side streams for the modality branches / fusions (shared by every model in the process).
    XVIT_STREAM_POOLS=split gives the fusions their own streams instead of re-forking the branch streams."""
    key = (device, n, kind if os.environ.get("XVIT_STREAM_POOLS", "shared") == "split" else "")
    if key not in _SIDE_STREAMS:
        if os.environ.get("XVIT_CU_SPLIT", "0") == "1":   # each branch owns 1/n of the CUs (see xvit/cu_mask.py)
            from . import cu_mask
            n_cu = torch.cuda.get_device_properties(device).multi_processor_count
            _SIDE_STREAMS[key] = [cu_mask.masked_stream(device, bits) for bits in cu_mask.split_masks(n_cu, n)]
        else:
            _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE_STREAMS[key]


class MultiScaleBlock(nn.Module):
    """model_cross.py:116-148: list of M token tensors in, list out."""

    def __init__(self, config, act_layer=nn.GELU):
        super().__init__()
        self.attn_order = config.attn_order        # {str(cls modality): str(token modality)}
        branch = lambda: nn.Sequential(*(SelfAttentionBlock(config) for _ in range(config.num_self_blocks)))  # noqa: E731
        self.blocks = nn.ModuleList(branch() for _ in range(config.num_modalities))   # separate weights per modality
        self.fusion = nn.ModuleList(CrossAttentionBlock(config) for _ in self.attn_order)

    def _parallel(self, thunks, tensors, kind="branches"):
        """Run independent pieces of work (one per modality) each on its own HIP stream so their kernels interleave
        on the chip (one GEMM's tail round / HBM-bound epilogue overlaps the other's MFMA phase, and the strings of
        tiny CLS-row launches of the two fusions overlap each other).  `thunks[i]` is None for a pass-through (nothing
        to launch: no fork).  `tensors` are the inputs the side streams read.  autograd replays backward on the same
        streams."""
        mode = STREAM_MODE.get() or os.environ.get("XVIT_STREAMS", "1")     # "1": branches and fusions, "branches": branches only, "0": one stream
        work = [i for i, f in enumerate(thunks) if f is not None]
        outs = [None] * len(thunks)
        if len(work) < 2 or not tensors[0].is_cuda or mode == "0" or (mode == "branches" and kind != "branches"):
            for i in work:
                outs[i] = thunks[i]()
            return outs
        dev = tensors[0].device
        cur = torch.cuda.current_stream(dev)
        streams = _side_streams(dev, len(thunks), kind)
        # While a HIP graph is being captured record_stream is not available (its deferred-event bookkeeping on private-pool
        # blocks is what capture_end tripped over); what it protects — a block freed by its owner stream's pool while another
        # stream still reads it — is prevented by keeping every tensor that crosses streams alive until the capture ends
        # (XF.keep, see functional.py).
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            XF.keep(*tensors)
        for i in work:
            st = streams[i]
            st.wait_stream(cur)
            if not capturing:
                for t in tensors:
                    t.record_stream(st)
            with torch.cuda.stream(st):
                outs[i] = thunks[i]()
        for i in work:
            cur.wait_stream(streams[i])
            if not capturing:
                outs[i].record_stream(cur)
            else:
                XF.keep(outs[i])
        return outs

    def _branches(self, x):
        """The M modality branches are independent until fusion (separate weights, :122)."""
        return self._parallel([(lambda x_=x_, block=block: block(x_)) for x_, block in zip(x, self.blocks)], list(x))

    def _branch_modules(self, i):
        """The modules of branch i as a list, built once (nn.Module.modules() walks and de-duplicates the tree at every call: 0.6 ms per
        step at the reference's batch); rebuilt when the branch's length changes."""
        cache = self.__dict__.setdefault("_branch_mods", {})
        ent = cache.get(i)
        if ent is None or ent[0] != len(self.blocks[i]):
            ent = cache[i] = (len(self.blocks[i]), list(self.blocks[i].modules()))
        return ent[1]

    def forward(self, x, cls_only=False, exclusive=False):
        """cls_only (used by ModelCross for its last block, whose outputs are read through their CLS rows only): the
        fusions return [B, 1, d] instead of re-attaching the new CLS token to a copy of the patch tokens.
        exclusive (ModelCross only): every output has exactly one consumer, whose backward returns a fresh gradient tensor —
        together with "the branch output was produced here and no forward hook saw it" that lets a fusion splice its CLS
        rows into the branch output in place (XF.CrossFusionFn); a direct caller gets the reference's copying cat."""
        for seq in self.blocks:          # a block whose input is another block's output returns its input gradient in bf16 too
            prev = None
            for blk in seq:
                if isinstance(blk, SelfAttentionBlock):
                    blk.feeds_a_block = isinstance(prev, SelfAttentionBlock) and not prev._forward_hooks
                prev = blk
        attn = self._branches(x)
        # every reader of a branch output (its own fusion, other fusions that take its patch tokens, the pass-through) gets
        # its own alias, so the gradients are summed by _FanOut instead of across streams inside the engine
        M = len(self.blocks)
        readers = [[("own", i)] if str(i) in self.attn_order else [("pass", i)] for i in range(M)]
        for k, v in self.attn_order.items():
            if int(v) != int(k):
                readers[int(v)].append(("tok", int(k)))
        alias = {}
        for i, a in enumerate(attn):
            fan = len(readers[i]) > 1 and a.is_cuda and a.requires_grad and os.environ.get("XVIT_FANOUT", "1") == "1"
            # cls-only block: the own fusion reads nothing but the CLS rows of its modality: hand it those alone
            cls_first = fan and cls_only and readers[i][0] == ("own", i) and self.attn_order.get(str(i)) != str(i) and os.environ.get("XVIT_CLS_NARROW", "1") == "1"
            last_blk = self.blocks[i][-1] if len(self.blocks[i]) > 0 else None
            from_block = isinstance(last_blk, SelfAttentionBlock) and not last_blk._forward_hooks and not self.blocks[i]._forward_hooks
            outs_i = _FanOut.apply(a, len(readers[i]), cls_first, from_block) if fan else [a] * len(readers[i])
            for r, t in zip(readers[i], outs_i):
                alias[(i,) + r] = t
        thunks = []
        cross_count = 0
        for i in range(len(self.blocks)):
            if str(i) in self.attn_order:
                j = int(self.attn_order[str(i)])
                blk = self.fusion[cross_count]
                # cls of i + patch tokens of j -> new cls, re-attached to i's own patch tokens (:140-142)
                xi = alias[(i, "own", i)]
                xj = xi if j == i else alias[(j, "tok", i)]
                own = exclusive and len(self.blocks[i]) > 0 and not any(m._forward_hooks for m in self._branch_modules(i))
                thunks.append(lambda xi=xi, xj=xj, blk=blk, own=own: XF.CrossFusionFn.apply(xi, xj, *_fusion_args(blk), not cls_only, _p(blk, blk.attn.fn.attn_drop), own))
                cross_count += 1
            else:
                thunks.append(None)                                    # no fusion for this modality: its tokens pass through (:146)
        outs = self._parallel(thunks, list(attn), kind="fusion")   # the fusions only read the branch outputs: independent of each other
        return [alias[(i, "pass", i)] if o is None else o for i, o in enumerate(outs)]


class ModelCross(EpochStatsMixin, _Base):
    """model_cross.py:152-308.  forward(img [B, M, 1, D, H, W], labels [B]) -> (logits, loss)."""

    def __init__(self, config):
        super().__init__()
        grid = [s // p for s, p in zip(config.img_size, config.patch_size)]
        if any(s % p for s, p in zip(config.img_size, config.patch_size)):
            raise AssertionError('image dimensions must be divisible by the patch size')
        P, pd, d, M = grid[0] * grid[1] * grid[2], config.patch_size[0] * config.patch_size[1] * config.patch_size[2], config.hidden_dim, config.num_modalities
        self.patch_size = tuple(config.patch_size)
        self.num_modalities = M
        for k in ("lr", "weight_decay", "optim_params", "label_smoothing"):
            setattr(self, k, getattr(config, k))
        # shared by every modality (model_cross.py:167-169)
        self.pos_embedding = nn.Parameter(torch.empty(1, P + 1, d))
        self.patch_to_embedding = _lin(pd, d)
        self.cls_token = nn.Parameter(torch.empty(1, 1, d))
        self.dropout = nn.Dropout(config.dropout)
        self.transformer = nn.Sequential(*(MultiScaleBlock(config) for _ in range(config.num_multi_blocks)))
        self.norm = nn.ModuleList(nn.LayerNorm(d) for _ in range(M))
        self.mlp_head = nn.ModuleList(_mlp_container(config, config.num_classes) for _ in range(M))
        self.initialize_model()

    def _sync_flat_weights(self):
        """Keep the weight matrices in one flat fp32 buffer + one flat bf16 operand copy (XF.FlatWeights);
        re-cast the latter with a single launch whenever any weight changed since the last forward."""
        grp = getattr(self, "_flat", None)
        if grp is None or not grp.intact() or grp.params[0].device != self.pos_embedding.device:
            if grp is not None:
                XF.SHADOWS.detach(grp)
            grp = XF.FlatWeights(list(self.parameters()))
            object.__setattr__(self, "_flat", grp)
            XF.SHADOWS.attach(grp)
        grp.refresh(force=XF.SHADOWS.force)
        XF.SHADOWS.force = False

    def forward(self, img, labels):
        if self.pos_embedding.is_cuda and os.environ.get("XVIT_FLAT_WEIGHTS", "1") != "0":
            self._sync_flat_weights()
        if img.shape[1] != self.num_modalities:
            raise ValueError(f"expected {self.num_modalities} modalities, got {img.shape[1]}")
        if img.is_cuda and torch.is_grad_enabled():
            XF.arena_begin(img.device)                   # the backward's small zeroed vectors: one fill per step (functional._zeros)
        tokens = XF.PatchEmbedFn.apply(img, self.patch_to_embedding.weight, self.patch_to_embedding.bias,
                                       self.cls_token, self.pos_embedding, self.patch_size, _p(self, self.dropout))
        x = list(tokens)
        for k, blk in enumerate(self.transformer):                       # nn.Sequential of MultiScaleBlocks (model_cross.py:171)
            last = k == len(self.transformer) - 1                        # only x[m][:, 0] of the last block is read below (model_cross.py:203) ...
            observed = bool(blk._forward_hooks) or bool(self.transformer._forward_hooks)   # ... unless a hook wants the reference's full output
            # the block's outputs go to the next block's branches (SelfAttentionBlockFn: fresh gradients) or to the heads; with a
            # hook on the block, or a next block without self blocks, somebody else may hold / alias them
            nxt = None if last else self.transformer[k + 1]
            sole = not observed and (last or all(len(b) > 0 for b in nxt.blocks))
            x = blk(x, cls_only=last and not observed, exclusive=sole)
        heads = [(lambda m=m: XF.HeadFn.apply(x[m], self.norm[m].weight, self.norm[m].bias, self.mlp_head[m][0].weight, self.mlp_head[m][0].bias,
                                              self.mlp_head[m][3].weight, self.mlp_head[m][3].bias, self.norm[m].eps, _p(self, self.mlp_head[m][2])))
                 for m in range(self.num_modalities)]
        if len(self.transformer) > 0 and os.environ.get("XVIT_HEAD_STREAMS", "0") == "1":    # experiment: each head on its modality's stream
            per_mod = self.transformer[-1]._parallel(heads, list(x), kind="fusion")
        else:
            per_mod = [h() for h in heads]
        logits, loss = XF.MeanCrossEntropyFn.apply(torch.stack(per_mod), labels, self.label_smoothing)
        return logits, loss

    # ---- initialisation (model_cross.py:214-241) -------------------------------------------
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
        self.apply(ModelCross.init_weights)
        nn.init.normal_(self.pos_embedding, mean=0.0, std=0.02)
        nn.init.normal_(self.cls_token, mean=0.0, std=0.02)

    # ---- training-loop surface used by main_mist.py / Lightning (model_cross.py:243-308) -----
    def log(self, *args, **kwargs):  # no-op unless the Lightning base provides it
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
