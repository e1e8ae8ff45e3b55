"""Autograd layer: one torch.autograd.Function per reference block, each with a hand-written
backward chain over the HIP kernels (fused residual + LayerNorm gradients, GELU' in the dgrad
epilogue, wgrad by k-strided MFMA tiles, flash-attention recompute).  torch provides the
tensors, the stream and the autograd graph edges between blocks — nothing else.

Numerics: operands bf16, accumulation fp32, residual stream / LN statistics / parameter
gradients fp32.  Weights are fp32 master parameters; their bf16 operand copies ("shadows")
are re-cast whenever a parameter's version changes (i.e. after every optimizer step).
"""
from __future__ import annotations

import os
import weakref

import torch
from torch.autograd import Function

from . import ops

# ------------------------------------------------------------------------------------------
# bf16 shadows of fp32 master weights
# ------------------------------------------------------------------------------------------


class _ShadowCache:
    """bf16 operand copies of fp32 master weights, keyed by the Parameter OBJECT (weakly held, so
    a recycled id() or device address can never return another tensor's copy) and refreshed when
    its version counter or storage changes.  Non-Parameter tensors are cast fresh every time."""

    def __init__(self):
        self._map = {}
        self.casts = 0
        self._groups = []   # weakref.ref(FlatWeights): the owner model holds the only strong reference (model._flat)
        self.force = False

    @property
    def groups(self):
        """Live FlatWeights groups; a discarded model's group (and its flat fp32/bf16 buffers) dies with the model."""
        live = [g for g in (r() for r in self._groups) if g is not None]
        if len(live) != len(self._groups):
            self._groups = [weakref.ref(g) for g in live]
        return live

    @staticmethod
    def _cast(params):
        with torch.no_grad():
            if len(params) == 1:
                return ops.cast_bf16(params[0].detach().contiguous())
            rows = sum(p.shape[0] for p in params)
            out = torch.empty(rows, *params[0].shape[1:], dtype=torch.bfloat16, device=params[0].device)
            r = 0
            for p in params:
                ops.cast_bf16(p.detach().contiguous(), out[r:r + p.shape[0]])
                r += p.shape[0]
            return out

    def get(self, *params):
        """bf16 copy of one parameter, or of several concatenated along dim 0."""
        for grp in self.groups:
            v = grp.lookup(params)
            if v is not None:
                return v
        self.casts += len(params)
        if not all(isinstance(p, torch.nn.Parameter) for p in params):
            self.casts += 0
            return self._cast(params)
        key = tuple(id(p) for p in params)
        tag = tuple((p._version, p.data_ptr()) for p in params)
        ent = self._map.get(key)
        if ent is not None and ent[1] == tag and all(r() is p for r, p in zip(ent[0], params)):
            self.casts -= len(params)
            return ent[2]
        out = self._cast(params)
        if len(self._map) > 4096:  # dead models: drop entries whose parameters are gone
            self._map = {k: v for k, v in self._map.items() if all(r() is not None for r in v[0])}
        self._map[key] = (tuple(weakref.ref(p) for p in params), tag, out)
        return out

    def invalidate(self):
        self._map.clear()
        self.force = True   # flat groups re-cast at their owner's next forward

    def attach(self, group):
        self._groups = [weakref.ref(g) for g in self.groups if g.intact() and g is not group] + [weakref.ref(group)]

    def detach(self, group):
        self._groups = [weakref.ref(g) for g in self.groups if g is not group]

    def drop(self, params):
        """Forget the cached copies of `params` (FusedAdam rewrites parameters through raw pointers, which does not bump
        their version counters: without this the GEMMs would keep training against the pre-step bf16 copies)."""
        ids = {id(p) for p in params}
        if ids:
            self._map = {k: v for k, v in self._map.items() if not ids.intersection(k)}


class FlatWeights:
    """All 2-D weight matrices of one model as views into ONE fp32 buffer (p.data is re-pointed, the
    Parameter objects, their names and their optimizer state are untouched) with a parallel bf16 buffer:
    the per-step operand cast is a single launch over the flat buffer instead of one per matrix, and
    row-concatenations of neighbours (wk|wv, query|key|value) are plain adjacent views."""

    def __init__(self, params):
        self.params = [p for p in params if p.dim() == 2]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        self.flat32 = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat16 = torch.empty(total, dtype=torch.bfloat16, device=dev)
        self.index, self.view16 = {}, []
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, offs)):
                v = self.flat32[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.view16.append(self.flat16[o:o + p.numel()].view(p.shape))
                self.index[id(p)] = i
        self.ptrs = [p.data_ptr() for p in self.params]
        self.by_ptr = {q: i for i, q in enumerate(self.ptrs)}
        self.stamp = None

    def intact(self):
        """False once any parameter's storage was replaced (e.g. module.to(...))."""
        return all(p.data_ptr() == q for p, q in zip(self.params, self.ptrs))

    def refresh(self, force=False):
        stamp = sum(p._version for p in self.params)
        if force or stamp != self.stamp:
            ops.cast_bf16(self.flat32, self.flat16)
            self.stamp = stamp

    def lookup(self, params):
        """bf16 view for one member, or for members that are neighbours with nothing between them."""
        idx = [self.index.get(id(p)) for p in params]
        if any(i is None or self.params[i] is not p for i, p in zip(idx, params)):
            # not the registered Parameter objects: accept leaf ALIASES of them (same storage and shape — xvit.graph.GraphedStep
            # runs the model on detached aliases so that its captured backward owns fresh AccumulateGrad nodes)
            idx = [self.by_ptr.get(p.data_ptr()) for p in params]
            if any(i is None or self.params[i].shape != p.shape or p.dtype != torch.float32 for i, p in zip(idx, params)):
                return None
        if len(idx) == 1:
            return self.view16[idx[0]]
        if any(b != a + 1 for a, b in zip(idx, idx[1:])) or any(self.params[i].numel() % 8 for i in idx[:-1]):
            return None
        first = self.view16[idx[0]]
        rows = sum(self.params[i].shape[0] for i in idx)
        return torch.as_strided(first, (rows,) + tuple(first.shape[1:]), first.stride(), first.storage_offset())


SHADOWS = _ShadowCache()


def invalidate_shadows():
    """Drop every cached bf16 weight copy (they are re-cast on next use).  A training loop
    does not need this: optimizer steps bump the parameter version.  bench.py calls it every
    step because it skips the optimizer but must still pay for the per-step cast."""
    SHADOWS.invalidate()


# ------------------------------------------------------------------------------------------
# HIP-graph capture and tensors that cross streams.  The caching allocator hands a freed block back to the pool of the
# stream it was ALLOCATED on, at once; a tensor that another stream reads is protected in eager mode by record_stream (reuse
# waits for that stream's work).  Under capture record_stream is skipped (its deferred-event bookkeeping on private-pool
# blocks crashed hipStreamEndCapture), so a block freed while a parallel graph branch still reads it could be handed to the
# owner stream's next allocation — in a replay the two branches then race on it (seen: the fusion of modality 1 reading the
# branch-0 tokens in its LayerNorm backward while the fusion of modality 0, a parallel branch, was already writing a new
# tensor there).  Allocation decisions are made at capture time only, so the cure is to keep every such tensor ALIVE until
# the capture ends: keep() parks a reference; xvit.graph.GraphedStep drops them after capture_end.
# ------------------------------------------------------------------------------------------

_CAPTURE_KEEP: list = []


def keep(*tensors):
    """While a HIP graph is being captured: park references to tensors that are (or may be) touched by more than one stream."""
    if tensors and torch.cuda.is_current_stream_capturing():
        _CAPTURE_KEEP.extend(t for t in tensors if t is not None)
    return tensors[0] if len(tensors) == 1 else tensors


def release_capture_keep():
    _CAPTURE_KEEP.clear()


# ------------------------------------------------------------------------------------------
# Gradients handed from one backward node to the next in BOTH dtypes.  A block's backward starts by casting its incoming fp32
# gradient to bf16 (the operand of its first GEMMs): a pass over 297 MB at configs[1].  The node that produced the gradient — the next
# block's LayerNorm backward, a fusion's, or the fan-out sum — can write the bf16 copy in the pass it makes anyway.  The copy rides
# on the gradient tensor as an attribute (the engine hands the very tensor object to the next node when that node is its only
# reader; any accumulation or hook in between yields a new tensor without it and the consumer casts as before).
# ------------------------------------------------------------------------------------------

B16_HANDOFF = os.environ.get("XVIT_B16_HANDOFF", "1") == "1"


def attach_b16(t, tb):
    """`tb`: bf16 copy of the gradient `t` about to be returned to autograd."""
    if tb is not None and B16_HANDOFF:
        t._xvit_b16 = tb
        keep(tb)
    return t


def b16_of(t, shape):
    """The bf16 copy that came with gradient `t` (same device, the expected 2-D shape), or None."""
    tb = getattr(t, "_xvit_b16", None) if B16_HANDOFF else None
    if tb is not None and tb.dtype == torch.bfloat16 and tb.device == t.device and tb.numel() == shape[0] * shape[1] and tb.is_contiguous():
        return tb.view(shape)
    return None


# ------------------------------------------------------------------------------------------
# dropout: counter-based masks.  A site's mask is a pure function of (seed, element index), so the
# backward pass regenerates it instead of storing it; seeds derive from torch's global seed and
# a call counter (reproducible under torch.manual_seed, different at every call).
# ------------------------------------------------------------------------------------------

_DROP_CALLS = 0


def drop_seeds(n):
    global _DROP_CALLS
    out = []
    for _ in range(n):
        _DROP_CALLS += 1
        out.append(((torch.initial_seed() & 0xFFFFFFFF) * 0x9E3779B1 + _DROP_CALLS * 0x85EBCA77) & 0x7FFFFFFFFFFFFFFF)
    return tuple(out)


def _dp(p, seed):
    """(p, seed) pair for the kernels, or None when dropout is off."""
    return (p, seed) if p > 0.0 else None


# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------


# depth cap of the split: with 9 tiles (a d x d gradient) 16 splits fill 144 of the 256 CUs, 28 fill 252: 106 -> 89 us at K = 64638
# (tools/gemm_model_bench.py, XVIT_WGRAD_SPLIT_CAP=16 / 20 / 24 / 28: 106.2 / 94.6 / 90.8 / 88.6 us)
_WGRAD_SPLIT_CAP = int(os.environ.get("XVIT_WGRAD_SPLIT_CAP", "28"))


def _wgrad_split(m, n, k):
    """Split-K factor for a TN wgrad: enough workgroups to fill 256 CUs, but >= 16 K-steps each on the 256x256 tiles (at the
    reference's batch 8 — 64 K-steps in all — 16 splits of 4 steps cost 29 us per d x d gradient where 4 of 16 cost 21: prologue,
    epilogue and slab traffic per split; tools/gemm_model_bench.py 8)."""
    if m >= 256 and n >= 256:   # 256x256 tiles, one block per CU
        tiles = ((m + 255) // 256) * ((n + 255) // 256)
        return max(1, min(256 // max(tiles, 1), ((k + 63) // 64) // 16, _WGRAD_SPLIT_CAP))
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    return max(1, min(512 // max(tiles, 1), ((k + 63) // 64) // 8, 32))


_WGRAD_STREAMS = {}


def _wgrad_stream(cur):
    """Companion stream of `cur` for weight-gradient GEMMs (opt-in with XVIT_WGRAD_STREAM=1: measured neutral, 18.14 vs 18.36 ms, once the modality streams overlap)."""
    if os.environ.get("XVIT_WGRAD_STREAM", "0") != "1" or ops.PROFILE is not None:
        return None
    key = (cur.device, cur.cuda_stream)
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=cur.device)
    return _WGRAD_STREAMS[key]


# Data-parallel runs: xvit.ddp.BucketedGradReducer registers, for every weight matrix, the view of its all-reduce bucket under the
# data_ptr of the fp32 master AND of its bf16 operand copy; a weight-gradient kernel then writes straight into the bucket (no pack
# copy of 373 MB per step at configs[1]) and the tensor autograd hands on IS that view.
GRAD_SINK = None


def _grad_out(like, shape, device):
    """Destination of a weight gradient: the reducer's bucket view registered for `like` (a master weight or its bf16 copy), else a fresh tensor."""
    if GRAD_SINK is not None and like is not None:
        v = GRAD_SINK.get(like.data_ptr())
        if v is not None and tuple(v.shape) == tuple(shape):
            return v.detach()                 # a fresh alias: autograd may take it over (its use count is 1)
    return torch.empty(*shape, dtype=torch.float32, device=device)


def _wgrad(dy_b, x_b, like=None):
    """dW[out, in] = dy^T x over all rows (tokens); fp32.  Weight gradients are off the critical path of a block's
    backward (nothing downstream reads them), so large ones are issued on a companion stream and overlap the
    HBM-bound kernels of the dgrad chain; `_join_wgrads()` re-joins before the Function returns.  `like`: the weight (or its
    bf16 copy) this is the gradient of — see GRAD_SINK."""
    out_f, in_f, k = dy_b.shape[1], x_b.shape[1], dy_b.shape[0]
    cur = torch.cuda.current_stream(dy_b.device)
    ws = _wgrad_stream(cur) if k >= 1024 else None
    if ws is None:
        dW = _grad_out(like, (out_f, in_f), dy_b.device)
        ops.gemm(ops.TN, dy_b, x_b, dW, split_k=_wgrad_split(out_f, in_f, k))
        return dW
    ws.wait_stream(cur)                       # operands are produced on `cur`
    with torch.cuda.stream(ws):
        dW = _grad_out(like, (out_f, in_f), dy_b.device)
        ops.gemm(ops.TN, dy_b, x_b, dW, split_k=_wgrad_split(out_f, in_f, k))
    return dW


def _join_wgrads(device):
    """Make the current stream wait for its companion wgrad stream (operand tensors may be freed afterwards)."""
    cur = torch.cuda.current_stream(device)
    ws = _WGRAD_STREAMS.get((cur.device, cur.cuda_stream))
    if ws is not None:
        cur.wait_stream(ws)


def _skinny_split(m, n, k):
    """Few-row GEMMs (the 1-token cross-attention path, heads) are bound by the latency of their serial
    K loop, not by FLOPs: cut K so ~100 workgroups stream the weight matrix concurrently."""
    if m > 128:
        return 1
    tiles = (n + 127) // 128
    return max(1, min(((k + 63) // 64) // 4, 96 // tiles))


def _linear(x_b, w_s, *, bias=None, residual=None, act=ops.ACT_NONE, aux=None, out_dtype=torch.bfloat16, dropout=None, **kw):
    y = torch.empty(x_b.shape[0], w_s.shape[0], dtype=out_dtype, device=x_b.device)
    return ops.gemm(ops.NT, x_b, w_s, y, bias=bias, residual=residual, act=act, aux=aux, dropout=dropout,
                    split_k=_skinny_split(x_b.shape[0], w_s.shape[0], w_s.shape[1]), **kw)


def _dgrad(dy_b, w_s, *, act=ops.ACT_NONE, aux=None, colsum=None, dropout=None, aux_mode=0):
    """dx = dy W (+ GELU' / dropout epilogue).  `colsum` (bias gradient of the Linear that produced dx's pre-image)
    is accumulated by the GEMM epilogue itself on every tile path."""
    m, n, k = dy_b.shape[0], w_s.shape[1], w_s.shape[0]
    dx = torch.empty(m, n, dtype=torch.bfloat16, device=dy_b.device)
    fused = None if ops.DETERMINISTIC else colsum          # the epilogue's column sums meet in fp32 atomics
    ops.gemm(ops.NN, dy_b, w_s, dx, act=act, aux=aux, colsum=fused, dropout=dropout, split_k=_skinny_split(m, n, k), aux_mode=aux_mode)
    if colsum is not None and fused is None:
        ops.colsum(dx, out=colsum, accumulate=True)        # fixed-order two-pass sum of the stored (bf16) values
    return dx


def _masked(t_b, p, seed):
    """In-place dropout mask on a freshly produced bf16 gradient tensor (p == 0: untouched)."""
    return ops.dropout(t_b, p, seed, out=t_b) if p > 0.0 else t_b


# One zero fill per step instead of ~28: the backward chains need many small zeroed fp32 vectors (the targets of atomically accumulated bias /
# LayerNorm gradients, dpos, dcls, ...).  ModelCross.forward opens a fresh 4 MB arena (one fill, on the caller's stream, before any fork: it
# happens-before every backward kernel); _zeros() hands out 256-byte aligned slices of it and falls back to torch.zeros when it is used up.
# A slice is never handed out twice; gradients that ARE slices keep the arena's storage alive as long as they live.
# An arena is only used in the regime it was opened in: one opened eagerly is not touched by a HIP-graph capture (its fill would not be
# part of the graph: replays would accumulate into memory zeroed once), and one opened inside a capture lives in that graph's pool
# (xvit.graph.GraphedStep drops it when the capture ends).  A model that never opens one (any module used outside ModelCross / ModelVIT)
# gets torch.zeros per request.
_ARENA = [None, 0, False]
ARENA_FLOATS = 1 << 20


def arena_begin(device):
    _ARENA[0] = torch.zeros(ARENA_FLOATS, dtype=torch.float32, device=device)
    _ARENA[1] = 0
    _ARENA[2] = torch.cuda.is_current_stream_capturing()


def arena_reset():
    _ARENA[0] = None


def _zeros(n, ref, shape=None):
    a = _ARENA[0]
    if a is not None and a.device == ref.device and _ARENA[1] + n <= a.numel() and _ARENA[2] == torch.cuda.is_current_stream_capturing():
        o = _ARENA[1]
        _ARENA[1] = o + (n + 63) // 64 * 64
        out = a[o:o + n]
    else:
        out = torch.zeros(n, dtype=torch.float32, device=ref.device)
    return out if shape is None else out.view(shape)


def _f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------
# pre-norm transformer block:  x + attn(LN(x)) ;  x + ffn(LN(x))
#   reference model_cross.py:64-72 (SelfAttentionBlock) and model.py:181-201 (Block)
# ------------------------------------------------------------------------------------------


# The all-token FFNs save gelu'(pre-activation) instead of the pre-activation itself (xvit_gemm aux_mode 1): the forward epilogue
# has the exponential at hand anyway, and the GELU' dgrad epilogue becomes a multiply.  XVIT_GELU_AUX=z restores the old form.
AUX_MODE = 0 if os.environ.get("XVIT_GELU_AUX", "deriv") == "z" else 1
# The fusion's K/V path: "lowrank" never applies wk / wv to the N tokens (xvit_head_* + two batched GEMMs over hn, see
# csrc/head_linear.hip); "dense" is the reference's literal order (kv = hn Wkv^T + b, then the CLS-query attention kernel).
# "auto" (default): low-rank wherever the GPU is the bound — XATTN_AUTO_ROWS token rows or more, or a step being captured into a HIP
# graph — and the literal order on small eager batches, which are bound by the host: it is 15 launches per fusion instead of 22
# (the reference's run shape at batch 8, eager: 13.0 vs 14.7 ms per step; captured: 8.59 vs 8.44 — tools/config_step_bench.py mist).
XATTN_FORM = os.environ.get("XVIT_XATTN_FORM", "auto")
XATTN_AUTO_ROWS = 8192


def _xattn_lowrank_ok(H, d, rows):
    """Shapes the low-rank form is built for (otherwise the dense form runs): 64-wide heads, H <= 16 (one 16-column operand).  Dropout on
    the probabilities is part of it (the kept weights feed the row sums; bv is weighted by their sum: csrc/head_linear.hip)."""
    if XATTN_FORM == "dense" or not (d == 64 * H and H <= 16 and d <= 1024):
        return False
    return XATTN_FORM == "lowrank" or rows >= XATTN_AUTO_ROWS or torch.cuda.is_current_stream_capturing()


def _attn_fwd(qkv, B, N, H, scale, p=0.0, seed=0):
    """Forward attention of the all-token blocks.  XVIT_ATTN_FP8=1 (opt-in, SURVEY.md 8 / configs[4]) runs QK^T and P.V on the
    MX-fp8 matrix instruction where the kernel applies (d_h = 64, no probability dropout): ~5e-2 output error instead of 2e-3
    (tests/test_attn_fp8_gpu.py), 7 % less forward-attention time at N = 4097 and MORE time at N = 513.
    The backward always runs the bf16 kernels on the saved o / lse."""
    if p == 0.0 and os.environ.get("XVIT_ATTN_FP8", "0") == "1" and qkv.shape[1] // (3 * H) == 64:
        return ops.attn_fwd_fp8(qkv, B, N, H, scale)
    return ops.attn_fwd(qkv, B, N, H, scale, dropout=(p, seed))


def block_forward(x, B, N, H, eps, scale, ln1w, ln1b, wqkv_s, bqkv, wo_s, bo, ln2w, ln2b, w1_s, b1, w2_s, b2, p_out=0.0, p_ffn=0.0, seeds=(0, 0, 0), p_attn=0.0, seed_attn=0):
    """x fp32 [B*N, d] -> (x2 fp32 [B*N, d], saved activations).  Dropout sites (reference
    model_cross.py:47,25,27 / model.py:177,114,116): after the out-projection (p_out), after GELU and
    after the second FFN Linear (p_ffn) — all fused into the producing GEMM's epilogue."""
    h1, mu1, rs1 = ops.layernorm_fwd(x, ln1w, ln1b, eps)
    qkv = _linear(h1, wqkv_s, bias=bqkv)
    o, lse = _attn_fwd(qkv, B, N, H, scale, p_attn, seed_attn)       # model.py:169: dropout on the probabilities
    x1 = _linear(o, wo_s, bias=bo, residual=x, out_dtype=torch.float32, dropout=_dp(p_out, seeds[0]))
    h2, mu2, rs2 = ops.layernorm_fwd(x1, ln2w, ln2b, eps)
    z = torch.empty(x.shape[0], w1_s.shape[0], dtype=torch.bfloat16, device=x.device)
    a = _linear(h2, w1_s, bias=b1, act=ops.ACT_GELU, aux=z, dropout=_dp(p_ffn, seeds[1]), aux_mode=AUX_MODE)     # z: gelu'(pre-activation)
    x2 = _linear(a, w2_s, bias=b2, residual=x1, out_dtype=torch.float32, dropout=_dp(p_ffn, seeds[2]))
    return x2, (x, mu1, rs1, h1, qkv, o, lse, x1, mu2, rs2, h2, z, a)


def block_backward(dy, saved, B, N, H, scale, ln1w, wqkv_s, has_bqkv, wo_s, ln2w, w1_s, w2_s, p_out=0.0, p_ffn=0.0, seeds=(0, 0, 0), p_attn=0.0, seed_attn=0,
                   dy_b16=None, want_dx_b16=False):
    """dy fp32 [B*N, d] -> (dx, grads dict).  Bias gradients cost no extra pass: b2 and bo fall out of the
    LN2 backward (column sums of its dres and dx), b1 out of the GELU' dgrad epilogue.  dy_b16: the bf16 copy of dy when the producer
    supplied one (b16_of); want_dx_b16: also return dx in bf16 (g["dx_b16"]) for the block in front."""
    x, mu1, rs1, h1, qkv, o, lse, x1, mu2, rs2, h2, z, a = saved
    d, f = x.shape[1], z.shape[1]
    zero = _zeros(6 * d + f, x)                               # every atomically-accumulated vector of this block
    g = dict(zip(("ln2w", "ln2b", "bo", "b2", "ln1w", "ln1b"), zero[:6 * d].split(d)))
    g["b1"] = zero[6 * d:]
    dyb = _masked(dy_b16 if dy_b16 is not None else ops.cast_bf16(dy), p_ffn, seeds[2])        # d(FFN out) = dy * mask
    # FFN
    dz = _dgrad(dyb, w2_s, act=ops.ACT_DGELU, aux=z, colsum=g["b1"], dropout=_dp(p_ffn, seeds[1]), aux_mode=AUX_MODE)
    g["w2"] = _wgrad(dyb, a, w2_s)
    dh2 = _dgrad(dz, w1_s)
    g["w1"] = _wgrad(dz, h2, w1_s)
    # without dropout the two bias gradients are column sums LN2's backward produces anyway
    dx1, dx1b = ops.layernorm_bwd(dh2, x1, mu2, rs2, ln2w, g["ln2w"], g["ln2b"], dres=dy, want_bf16=True,
                                  dxsum=g["bo"] if p_out == 0.0 else None, dressum=g["b2"] if p_ffn == 0.0 else None)
    if p_ffn > 0.0:
        ops.colsum(dyb, out=g["b2"], accumulate=True)
    if p_out > 0.0:
        dx1b = _masked(dx1b, p_out, seeds[0])
        ops.colsum(dx1b, out=g["bo"], accumulate=True)
    # attention
    do = _dgrad(dx1b, wo_s)
    g["wo"] = _wgrad(dx1b, o, wo_s)
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale, dropout=(p_attn, seed_attn))
    dh1 = _dgrad(dqkv, wqkv_s)
    g["wqkv"] = _wgrad(dqkv, h1, wqkv_s)
    if has_bqkv:
        g["bqkv"] = ops.colsum(dqkv)
    dx, g["dx_b16"] = ops.layernorm_bwd(dh1, x, mu1, rs1, ln1w, g["ln1w"], g["ln1b"], dres=dx1, want_bf16=want_dx_b16)
    _join_wgrads(x.device)
    return dx, g


class SelfAttentionBlockFn(Function):
    """model_cross.SelfAttentionBlock: fused-qkv (no bias), eps 1e-5, scale dh**-0.5."""

    @staticmethod
    def forward(ctx, x, ln1w, ln1b, wqkv, wo, bo, ln2w, ln2b, w1, b1, w2, b2, H, eps, p=0.0, feeds_a_block=False):
        """feeds_a_block: the gradient this node returns for x goes to another SelfAttentionBlockFn (the block in front of it in a
        branch): hand it on in bf16 as well (attach_b16)."""
        B, N, d = x.shape
        scale = (d // H) ** -0.5
        ctx.feeds_a_block = bool(feeds_a_block)
        sh = (SHADOWS.get(wqkv), SHADOWS.get(wo), SHADOWS.get(w1), SHADOWS.get(w2))
        seeds = drop_seeds(3) if p > 0.0 else (0, 0, 0)
        x2, saved = block_forward(_f32c(x).reshape(B * N, d), B, N, H, eps, scale, ln1w, ln1b, sh[0], None, sh[1], bo, ln2w, ln2b, sh[2], b1, sh[3], b2,
                                  p, p, seeds)
        ctx.drop = (p, p, seeds)
        ctx.meta = (B, N, H, scale, x.dtype)
        ctx.save_for_backward(ln1w, ln2w, *sh, *saved)
        return x2.reshape(B, N, d)

    @staticmethod
    def backward(ctx, dy):
        B, N, H, scale, xdt = ctx.meta
        ln1w, ln2w, wqkv_s, wo_s, w1_s, w2_s, *saved = ctx.saved_tensors
        d = ln1w.shape[0]
        dx, g = block_backward(_f32c(dy).reshape(B * N, -1), saved, B, N, H, scale, ln1w, wqkv_s, False, wo_s, ln2w, w1_s, w2_s, *ctx.drop,
                               dy_b16=b16_of(dy, (B * N, d)), want_dx_b16=ctx.feeds_a_block and B16_HANDOFF)
        keep(dx, dy)
        out = attach_b16(dx.reshape(B, N, -1).to(xdt), g["dx_b16"])
        return (out, g["ln1w"], g["ln1b"], g["wqkv"], g["wo"], g["bo"], g["ln2w"], g["ln2b"], g["w1"], g["b1"], g["w2"], g["b2"], None, None, None, None)


class EncoderBlockFn(Function):
    """model.Block: separate biased query/key/value, eps 1e-6, scores / sqrt(dh)."""

    @staticmethod
    def forward(ctx, x, ln1w, ln1b, wq, bq, wk, bk, wv, bv, wo, bo, ln2w, ln2b, w1, b1, w2, b2, H, eps, p_out=0.0, p_ffn=0.0, p_attn=0.0):
        B, N, d = x.shape
        scale = 1.0 / float(d // H) ** 0.5
        sh = (SHADOWS.get(wq, wk, wv), SHADOWS.get(wo), SHADOWS.get(w1), SHADOWS.get(w2))
        bqkv = torch.cat((bq, bk, bv)).detach()
        seeds = drop_seeds(3) if (p_out > 0.0 or p_ffn > 0.0) else (0, 0, 0)
        seed_attn = drop_seeds(1)[0] if p_attn > 0.0 else 0
        x2, saved = block_forward(_f32c(x).reshape(B * N, d), B, N, H, eps, scale, ln1w, ln1b, sh[0], bqkv, sh[1], bo, ln2w, ln2b, sh[2], b1, sh[3], b2,
                                  p_out, p_ffn, seeds, p_attn, seed_attn)
        ctx.drop = (p_out, p_ffn, seeds, p_attn, seed_attn)
        ctx.meta = (B, N, H, scale, d, x.dtype)
        ctx.save_for_backward(ln1w, ln2w, *sh, *saved)
        return x2.reshape(B, N, d)

    @staticmethod
    def backward(ctx, dy):
        B, N, H, scale, d, xdt = ctx.meta
        ln1w, ln2w, wqkv_s, wo_s, w1_s, w2_s, *saved = ctx.saved_tensors
        dx, g = block_backward(_f32c(dy).reshape(B * N, -1), saved, B, N, H, scale, ln1w, wqkv_s, True, wo_s, ln2w, w1_s, w2_s, *ctx.drop)
        wq, wk, wv = g["wqkv"].split(d, dim=0)
        bq, bk, bv = g["bqkv"].split(d)
        return (dx.reshape(B, N, -1).to(xdt), g["ln1w"], g["ln1b"], wq, bq, wk, bk, wv, bv, g["wo"], g["bo"], g["ln2w"], g["ln2b"], g["w1"], g["b1"], g["w2"], g["b2"], None, None, None, None, None)


# ------------------------------------------------------------------------------------------
# cross-attention fusion (model_cross.py:104-114 CrossAttentionBlock, :135-142 routing)
# ------------------------------------------------------------------------------------------


def cross_forward(xi, xj, B, N, H, eps, ln1w, ln1b, wq, bq, wkv_s, bkv, wp, bp, ln2w, ln2b, w1, b1, w2, b2, p=0.0, seeds=(0, 0, 0, 0), pack_cls=False, wk=None, wv=None, bv=None):
    """xi, xj fp32 [B*N, d] (cls taken from xi, patch tokens from xj) -> (y2 fp32 [B, d], saved).

    Two precisions on purpose.  The key/value projection runs over all N tokens: bf16 operands on the MFMA tile kernels
    (wkv_s = bf16 copy of wk|wv).  Everything downstream of the CLS query is ONE row per sample — wq, the attention
    output, proj, LayerNorm, the FFN (wq, wp, w1, w2 = the fp32 master weights) — and runs fp32 operands
    (xvit_linear_f32): that row feeds 2 small logits through ~10 stages with nothing to average bf16 storage rounding
    over, which cost 5e-3 on the CLS rows and ~2e-2 on the logits for <0.1 % of the FLOPs.  bf16 copies of the
    activations are emitted alongside for the backward chain (unchanged: bf16 operands, fp32 accumulation)."""
    d = xi.shape[1]
    scale = (d // H) ** -0.5
    hn, mu, rs = ops.layernorm_fwd(xj, ln1w, ln1b, eps, x_alt=xi, seq_len=N)
    # un-normed CLS rows (row 0 of the concat; the residual, :112): rows 0, N, 2N, .. of the token tensor (ld = N*d), or xi itself when the
    # caller hands just the packed [B, d] CLS rows
    cls_in = xi if xi.shape[0] == B else xi.reshape(B, N * d)[:, :d]
    hn0f, _, _, _ = ops.layernorm_fwd_f32(cls_in, ln1w, ln1b, eps, want_bf16=False)
    qf, qb, _ = ops.linear_f32(hn0f, wq, bq, want_bf16=True)
    lowrank = wk is not None and _xattn_lowrank_ok(H, d, B * N)
    if lowrank:
        # scores = hn . (q_h Wk_h), out = Wv_h (sum_n p hn) + bv: wk / wv meet one row per (sample, head), never the N tokens
        hn3 = hn.view(B, N, d)
        R = torch.empty(2 * H, B, d, dtype=torch.float32, device=xi.device)       # (U | Y): the first half now, the second in the backward
        Ub = torch.empty(B, 16, d, dtype=torch.bfloat16, device=xi.device)        # GEMM operand: H rows, zero-padded to 16 by the kernel
        ops.head_rows(qf, wk, R[:H].transpose(0, 1), H, out_bf16=Ub)
        sc = torch.empty(B, N, 16, dtype=torch.float32, device=xi.device)
        ops.gemm(ops.NT, hn3, Ub, sc)                                              # [N, d] x [d, 16] per sample
        S = torch.empty(B, 16, d, dtype=torch.float32, device=xi.device)
        if p > 0.0:   # attn_drop (model_cross.py:97): the row sums run over the KEPT weights; stat = (rz, rz / (1 - p), that times their sum)
            e, rz, ek = ops.cls_softmax_fwd(sc, H, scale, dropout=(p, seeds[0]))
            ops.gemm(ops.TN, ek, hn3, S)
            ocf, oc = ops.head_cols(S, wv, H, row_scale=rz[1], bias=bv, bias_scale=rz[2], want_bf16=True)
        else:
            e, rz = ops.cls_softmax_fwd(sc, H, scale)
            ops.gemm(ops.TN, e, hn3, S)                                            # [16, N] x [N, d] per sample
            ocf, oc = ops.head_cols(S, wv, H, row_scale=rz, bias=bv, want_bf16=True)
        kv, pr = (R, e, rz, S, qf), None                                           # what the backward needs instead of kv / p
    else:
        kv = _linear(hn, wkv_s, bias=bkv)
        # dropout sites (model_cross.py:97,101,25,27): probabilities, proj output, after GELU, FFN output
        oc, pr, ocf = ops.cls_xattn_fwd(qf, kv, B, N, H, scale, dropout=(p, seeds[0]), want_f32=True)
    y, _, _ = ops.linear_f32(ocf, wp, bp, residual=cls_in, dropout=_dp(p, seeds[1]))
    h2f, h2, mu2, rs2 = ops.layernorm_fwd_f32(y, ln2w, ln2b, eps)
    af, a, z = ops.linear_f32(h2f, w1, b1, act=ops.ACT_GELU, want_z=True, want_bf16=True, dropout=_dp(p, seeds[2]))
    y2, _, _ = ops.linear_f32(af, w2, b2, residual=y, dropout=_dp(p, seeds[3]))
    # the backward needs xi only for its CLS rows (row 0 of the normed concat): a packed [B, d] copy when the caller is about
    # to overwrite them in place (CrossFusionFn), else xi itself
    keep_xi = ops.rows_combine(torch.empty(B, d, dtype=torch.float32, device=xi.device), a=cls_in) if pack_cls else xi
    if lowrank:
        return y2, (keep_xi, xj, mu, rs, hn, *kv, qb, oc, y, mu2, rs2, h2, z, a)
    return y2, (keep_xi, xj, mu, rs, hn, kv, qb, oc, pr, y, mu2, rs2, h2, z, a)


def cross_backward(dy2, saved, B, N, H, ln1w, wq_s, wkv_s, wp_s, ln2w, w1_s, w2_s, pd=0.0, seeds=(0, 0, 0, 0), wk=None, wv=None, want_dcat_b16=False):
    """dy2 fp32 [B, d] -> (dcat fp32 [B*N, d] = grad of the normed concat input, dcls_res fp32 [B, d], grads); want_dcat_b16: g["dcat_b16"] is
    dcat's bf16 copy, written by the same LayerNorm backward (attach_b16)."""
    lowrank = len(saved) == 18                                            # the forward ran the low-rank form: (R, e, rz, S, qf) instead of (kv, p)
    if lowrank:
        xi, xj, mu, rs, hn, R, e, rz, S, qf, q, oc, y, mu2, rs2, h2, z, a = saved
    else:
        xi, xj, mu, rs, hn, kv, q, oc, p, y, mu2, rs2, h2, z, a = saved       # xi: the token tensor or a packed copy of its CLS rows
    d, f = xi.shape[1], z.shape[1]
    scale = (d // H) ** -0.5
    zero = _zeros(7 * d + f, xi)
    g = dict(zip(("ln2w", "ln2b", "bp", "b2", "ln1w", "ln1b", "bk0"), zero[:7 * d].split(d)))
    g["b1"] = zero[7 * d:]
    bk0 = g.pop("bk0")                                                    # stays zero: wk.bias has no gradient (softmax shift invariance)
    dyb = _masked(ops.cast_bf16(dy2), pd, seeds[3])
    dz = _dgrad(dyb, w2_s, act=ops.ACT_DGELU, aux=z, colsum=g["b1"], dropout=_dp(pd, seeds[2]))
    g["w2"] = _wgrad(dyb, a, w2_s)
    dh2 = _dgrad(dz, w1_s)
    g["w1"] = _wgrad(dz, h2, w1_s)
    nd = pd == 0.0
    dy, dyb1 = ops.layernorm_bwd(dh2, y, mu2, rs2, ln2w, g["ln2w"], g["ln2b"], dres=dy2, want_bf16=True,
                                 dxsum=g["bp"] if nd else None, dressum=g["b2"] if nd else None)
    if not nd:
        ops.colsum(dyb, out=g["b2"], accumulate=True)
        dyb1 = _masked(dyb1, pd, seeds[1])
        ops.colsum(dyb1, out=g["bp"], accumulate=True)
    g["wp"] = _wgrad(dyb1, oc, wp_s)
    if lowrank:
        # d(attention output) in fp32: it meets Wv_h as an fp32 row (xvit_head_rows)
        doc = torch.empty(B, d, dtype=torch.float32, device=xi.device)
        ops.gemm(ops.NN, dyb1, wp_s, doc, split_k=_skinny_split(B, d, d))
        hn3 = hn.view(B, N, d)
        Yb = torch.empty(B, 16, d, dtype=torch.bfloat16, device=xi.device)
        ops.head_rows(doc, wv, R[H:].transpose(0, 1), H, out_bf16=Yb)             # Y[b, h] = dO_h Wv_h: dp[b, n, h] = hn[b, n] . Y[b, h]
        dp = torch.empty(B, N, 16, dtype=torch.float32, device=xi.device)
        ops.gemm(ops.NT, hn3, Yb, dp)
        drop = pd > 0.0                                                            # rz is then the forward's stat block (rz, rz / (1 - p), bv's weight)
        coef, dsb = ops.cls_softmax_bwd(e, rz[0] if drop else rz, dp, H, scale, dropout=(pd, seeds[0]))   # (ds | p') per token and head
        dhn = ops.xattn_kv_dgrad(coef, R, B, N, H, d)                              # dhn[n] = sum_h ds U_h + p Y_h
        T = torch.empty(B, 16, d, dtype=torch.float32, device=xi.device)
        ops.gemm(ops.TN, dsb, hn3, T)                                              # T[b, h] = sum_n ds hn[b, n]
        dq, _ = ops.head_cols(T, wk, H)                                            # dq_h = Wk_h T_h  (the bk term carries sum_n ds = 0)
        g["wk"] = ops.head_wgrad(qf, T, H, out=_grad_out(wk, (d, d), xi.device))
        g["wv"] = ops.head_wgrad(doc, S, H, row_scale=rz[1] if drop else rz, out=_grad_out(wv, (d, d), xi.device))
        g["bv"] = ops.head_bias_grad(doc, rz[2], H) if drop else ops.colsum(doc)
        g["bk"] = bk0                                                              # analytically zero: sum_n ds[n] = 0
        return _cross_backward_tail(g, dq, dhn, hn, q, xi, xj, mu, rs, ln1w, wq_s, B, N, d, dy, want_dcat_b16)
    # the reference's literal order (XVIT_XATTN_FORM=dense, more than 16 heads, head widths other than 64): the K/V projection's
    # gradient through the [B N, 2 d] tensor — a K = 2 d dgrad GEMM, a wgrad GEMM and a column-sum pass
    doc = _dgrad(dyb1, wp_s)
    dq, dkv = ops.cls_xattn_bwd(q, kv, p, doc, B, N, H, scale, dropout=(pd, seeds[0]))
    dhn = _dgrad(dkv, wkv_s)                                # [B*N, d] bf16
    g["wkv"] = _wgrad(dkv, hn)
    g["bkv"] = ops.colsum(dkv)
    g["wk"], g["wv"] = g["wkv"].split(d, dim=0)
    g["bk"], g["bv"] = g["bkv"].split(d)
    return _cross_backward_tail(g, dq, dhn, hn, q, xi, xj, mu, rs, ln1w, wq_s, B, N, d, dy, want_dcat_b16)


def _cross_backward_tail(g, dq, dhn, hn, q, xi, xj, mu, rs, ln1w, wq_s, B, N, d, dy, want_dcat_b16=False):
    """The query path's gradient joins the CLS rows of dhn, then the LayerNorm over the normed concat."""
    dqb = ops.cast_bf16(dq)
    dhq = _dgrad(dqb, wq_s)                                 # [B, d] bf16: the query path reaches row 0 only
    dhn0 = dhn.reshape(B, N * d)[:, :d]
    ops.rows_combine(dhn0, a=dhn0, b=dhq)                   # B rows: merge the two paths into the CLS rows (fp32 add, one bf16 rounding)
    hn0 = hn.reshape(B, N * d)[:, :d]
    g["wq"] = _wgrad(dqb, hn0, wq_s)
    g["bq"] = ops.colsum(dq)
    dcat, g["dcat_b16"] = ops.layernorm_bwd(dhn, xj, mu, rs, ln1w, g["ln1w"], g["ln1b"], x_alt=xi, seq_len=N, want_bf16=want_dcat_b16)
    _join_wgrads(xi.device)
    return dcat, dy, g


class CrossFusionFn(Function):
    """out_i = cat(CrossAttentionBlock(cat(cls_i, patches_j)), patches_i)  (model_cross.py:140-142)."""

    @staticmethod
    def forward(ctx, xi, xj, ln1w, ln1b, wq, bq, wk, bk, wv, bv, wp, bp, ln2w, ln2b, w1, b1, w2, b2, H, eps, concat, p=0.0, exclusive=False):
        B, N, d = xj.shape
        narrow = xi.shape[1] == 1 and N > 1                    # just the CLS rows of modality i (cls-only blocks, cross_vit._FanOut)
        if narrow and concat:
            raise ValueError("CrossFusionFn: a [B, 1, d] xi cannot be concatenated with patch tokens it does not carry")
        sh = (SHADOWS.get(wq), SHADOWS.get(wk, wv), SHADOWS.get(wp), SHADOWS.get(w1), SHADOWS.get(w2))
        bkv = torch.cat((bk, bv)).detach()
        xi2, xj2 = _f32c(xi).reshape(B if narrow else B * N, d), _f32c(xj).reshape(B * N, d)
        seeds = drop_seeds(4) if p > 0.0 else (0, 0, 0, 0)
        # concat: the output is x_i with its CLS rows replaced by the fused token (model_cross.py:142) — by default a copy, as
        # the reference's torch.cat.  `exclusive` (set by MultiScaleBlock when ModelCross drives it, never by a direct caller):
        # x_i was produced inside the block, nobody else holds it or reads its CLS rows (the other fusions take its PATCH rows,
        # the producing block saves its input, not its output, no forward hook saw it), and this node's output has ONE consumer
        # whose backward returns a fresh gradient tensor.  Then the new rows are written IN PLACE and the output aliases x_i (no
        # 166 MB copy per fusion; the write goes through .data: autograd must not take x_i's patch rows, saved by the other
        # fusion, for modified), the backward keeps a packed copy of the original CLS rows and replaces the CLS rows of the
        # incoming gradient where they are.  XVIT_CLS_INPLACE=0 forces the copying form.
        inplace = concat and exclusive and os.environ.get("XVIT_CLS_INPLACE", "1") == "1"
        y2, saved = cross_forward(xi2, xj2, B, N, H, eps, ln1w, ln1b, wq.detach(), bq, sh[1], bkv, wp.detach(), bp, ln2w, ln2b, w1.detach(), b1,
                                  w2.detach(), b2, p, seeds, pack_cls=inplace, wk=wk.detach(), wv=wv.detach(), bv=bv.detach())
        ctx.drop = (p, seeds)
        ctx.meta = (B, N, H, d, concat, inplace, narrow)
        ctx.save_for_backward(ln1w, ln2w, *sh, wk.detach(), wv.detach(), *saved)
        if not concat:
            return y2.reshape(B, 1, d)
        if inplace:
            ops.rows_combine(xi2.data.reshape(B, N * d)[:, :d], a=y2)
            return xi2.reshape(B, N, d)
        out = xi2.clone().reshape(B, N, d)
        out[:, 0] = y2
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, H, d, concat, inplace, narrow = ctx.meta
        ln1w, ln2w, wq_s, wkv_s, wp_s, w1_s, w2_s, wk_m, wv_m, *saved = ctx.saved_tensors
        dout = _f32c(dout)
        dy2 = ops.rows_combine(torch.empty(B, d, dtype=torch.float32, device=dout.device), a=dout[:, 0])
        # a cls-only fusion's patch-row gradient is (up to the CLS rows) the whole gradient of the partner's last block: hand it on in bf16 too
        dcat, dcls_res, g = cross_backward(dy2, saved, B, N, H, ln1w, wq_s, wkv_s, wp_s, ln2w, w1_s, w2_s, *ctx.drop, wk=wk_m, wv=wv_m,
                                           want_dcat_b16=not concat and B16_HANDOFF)
        dcat = dcat.reshape(B, N, d)
        # cls row -> x_i (normed-concat path + the un-normed residual path); patch rows -> x_j
        if concat:
            # exclusive form: the incoming gradient has this node as its only reader (see forward), so its CLS rows are replaced
            # where they are; otherwise the caller's grad_outputs / a gradient shared with another node stays untouched
            dxi = dout if inplace else dout.clone()
        elif narrow:
            dxi = torch.empty(B, 1, d, dtype=torch.float32, device=dout.device)       # xi was the CLS rows alone: so is its gradient
        else:
            dxi = torch.zeros(B, N, d, dtype=torch.float32, device=dout.device)
        ops.rows_combine(dxi[:, 0], a=dcat[:, 0], b=dcls_res)
        if getattr(dxi, "_xvit_b16", None) is not None:
            del dxi._xvit_b16                                  # its CLS rows just changed: a bf16 copy that came with it is stale
        dxj = dcat
        db16 = g["dcat_b16"]
        ops.rows_combine(dxj[:, 0], dst2=db16.view(B, N, d)[:, 0] if db16 is not None else None)     # the CLS rows belong to x_i: zero here
        if db16 is not None:
            dxj = attach_b16(dxj, db16)
        wk, wv, bk, bv = g["wk"], g["wv"], g["bk"], g["bv"]
        keep(dxi, dxj, dout)
        return (dxi, dxj, g["ln1w"], g["ln1b"], g["wq"], g["bq"], wk, bk, wv, bv, g["wp"], g["bp"], g["ln2w"], g["ln2b"], g["w1"], g["b1"], g["w2"], g["b2"], None, None, None, None, None)


# ------------------------------------------------------------------------------------------
# patch embedding of all modalities at once (model_cross.py:191-199)
# ------------------------------------------------------------------------------------------


class PatchEmbedFn(Function):
    """img [B, M, 1, D, H, W] -> tokens fp32 [M, B, N, d] = cat(cls, patches W^T + b) + pos."""

    @staticmethod
    def forward(ctx, img, w, b, cls, pos, patch, p=0.0, concat=False):
        Bn, M = img.shape[0], img.shape[1]
        d, pd = w.shape
        w_s = SHADOWS.get(w)
        # a bf16 volume tensor feeds the GEMM's loaders directly (no [rows, pd] patch matrix in HBM: -2 GB of traffic and
        # -1 GB resident per step at configs[1]); other inputs (fp32 volumes, ModelVIT's concatenated sequence, small or odd
        # geometries) go through the patchify kernel, which also converts
        fused = not concat and ops.patch_embed_supported(img, patch, d, cls_rows=1)
        if fused:
            N = 1 + (img.shape[3] // patch[0]) * (img.shape[4] // patch[1]) * (img.shape[5] // patch[2])
            pos2 = pos.detach().reshape(N, d)
            x = ops.patch_embed_fwd(img, patch, w_s, b, pos2, cls_rows=1)
            patches = img
        else:
            if concat:   # ModelVIT (modelv3.py:123-137): one sequence per sample, cls + the patches of every modality
                patches = ops.patchify(img.contiguous(), patch, concat=True)                    # [B*(M*P+1), pd]
                M = 1
            else:        # ModelCross (model_cross.py:191-199): one sequence per (modality, sample)
                patches = ops.patchify(img.contiguous(), patch, pad_cls_row=True).reshape(-1, pd)   # [M*B*N, pd], row 0 of each sample = 0
            N = patches.shape[0] // (M * Bn)
            pos2 = pos.detach().reshape(N, d)
            x = torch.empty(M * Bn * N, d, dtype=torch.float32, device=img.device)
            ops.gemm(ops.NT, patches, w_s, x, bias=b, residual=pos2, res_row_mod=N, res_row_off=0)
        ops.cls_row_fwd(cls.detach().reshape(d), pos2, x, M * Bn, N, d)
        seed = drop_seeds(1)[0] if p > 0.0 else 0
        if p > 0.0:                       # self.dropout on the embedded tokens (model_cross.py:198)
            ops.dropout(x, p, seed, out=x)
        ctx.meta = (M, Bn, N, d, p, seed, patch if fused else None)
        ctx.save_for_backward(patches, w_s)     # fused: the volume tensor itself
        if concat:
            return x.reshape(Bn, N, d)
        # one output per modality (views of one buffer): slicing a stacked [M, B, N, d] output in the caller would cost
        # autograd two full-size zero fills, two slice copies and a full-size add per step (select_backward)
        return tuple(x.reshape(M, Bn, N, d).unbind(0))

    @staticmethod
    def backward(ctx, *dxs):
        M, Bn, N, d, p, seed, fused_patch = ctx.meta
        patches, w_s = ctx.saved_tensors
        dev = patches.device
        dpos = _zeros(N * d, patches, (N, d))
        dcls = _zeros(d, patches)
        if len(dxs) == 1 or p > 0.0:         # single sequence (ModelVIT), or dropout (its mask is keyed by the index in the stacked tensor)
            dx = dxs[0] if len(dxs) == 1 else torch.stack([_f32c(g) for g in dxs])
            dx2 = _f32c(dx).reshape(M * Bn * N, d)
            if p > 0.0:
                dx2 = ops.dropout(dx2, p, seed)
            dxb = ops.cast_bf16(dx2)
            ops.embed_bwd(dx2, dpos, dcls, M * Bn, N, d)
        else:                                # per-modality gradients: cast each into its slice of ONE bf16 operand, no stacking copy
            dxb = torch.empty(M * Bn * N, d, dtype=torch.bfloat16, device=dev)
            for m, g in enumerate(dxs):
                g2 = _f32c(g).reshape(Bn * N, d)
                ops.cast_bf16(g2, dxb[m * Bn * N:(m + 1) * Bn * N])
                ops.embed_bwd(g2, dpos, dcls, Bn, N, d)      # accumulates into dpos / dcls
        if fused_patch is not None:
            dW = ops.patch_embed_wgrad(patches, fused_patch, dxb, cls_rows=1, out=_grad_out(w_s, tuple(w_s.shape), dev))     # contraction over the patch rows only
        else:
            dW = _wgrad(dxb, patches, w_s)   # the zero CLS rows of `patches` drop the CLS-row gradients
        db = ops.colsum(dpos[1:])            # bias reaches the P patch rows of every sample
        _join_wgrads(dev)
        return None, dW, db, dcls.reshape(1, 1, d), dpos.reshape(1, N, d), None, None, None


# ------------------------------------------------------------------------------------------
# heads + loss (model_cross.py:203-211)
# ------------------------------------------------------------------------------------------


class HeadFn(Function):
    """logits_m = Linear(GELU(Linear(LN(x)[:, 0]))): only the CLS row of the final norm is used."""

    @staticmethod
    def forward(ctx, x, lnw, lnb, w0, b0, w3, b3, eps, p=0.0):
        B, N, d = x.shape
        x2 = _f32c(x).reshape(B, N * d)[:, :d]              # CLS rows, ld = N*d
        # one row per sample: fp32 operands throughout (see cross_forward); bf16 copies only feed the backward chain
        hf, h, mu, rs = ops.layernorm_fwd_f32(x2, lnw, lnb, eps)
        w0_s = SHADOWS.get(w0)
        seeds = drop_seeds(2) if p > 0.0 else (0, 0)
        af, a, z = ops.linear_f32(hf, w0.detach(), b0.detach(), act=ops.ACT_GELU, want_z=True, want_bf16=True, dropout=_dp(p, seeds[0]))   # mlp_head[m][0..2]
        logits, _, _ = ops.linear_f32(af, w3.detach(), b3.detach())
        if p > 0.0:                                                                     # mlp_head[m][4]: dropout on the logits
            ops.dropout(logits, p, seeds[1], out=logits)
        ctx.drop = (p, seeds)
        ctx.meta = (B, N, d)
        ctx.save_for_backward(x2, mu, rs, h, z, a, lnw, w0_s, w3)
        return logits

    @staticmethod
    def backward(ctx, dl):
        B, N, d = ctx.meta
        x2, mu, rs, h, z, a, lnw, w0_s, w3 = ctx.saved_tensors
        p, seeds = ctx.drop
        dl = _f32c(dl)
        if p > 0.0:
            dl = ops.dropout(dl, p, seeds[1])
        dW3, db3 = _zeros(w3.numel(), dl, tuple(w3.shape)), _zeros(w3.shape[0], dl)
        dz = _masked(ops.small_linear_bwd(dl, a, w3.detach(), dW3, db3, z=z), p, seeds[0])
        dh = _dgrad(dz, w0_s)
        dW0 = _wgrad(dz, h, w0_s)
        db0 = ops.colsum(dz)
        dg, dbeta = _zeros(d, dl), _zeros(d, dl)
        dxc, _ = ops.layernorm_bwd(dh, x2, mu, rs, lnw, dg, dbeta)
        _join_wgrads(dl.device)
        dx = _zeros(B * d, dl, (B, 1, d)) if N == 1 else torch.zeros(B, N, d, dtype=torch.float32, device=dl.device)
        dx[:, 0] = dxc
        keep(dx)
        return dx, dg, dbeta, dW0, db0, dW3, db3, None, None


class MeanCrossEntropyFn(Function):
    """(logits_m [M,B,C], labels) -> (mean logits [B,C], CE loss)."""

    @staticmethod
    def forward(ctx, logits_m, labels, smoothing):
        logits, loss, dl = ops.mean_ce(_f32c(logits_m), labels.to(torch.int64).contiguous(), float(smoothing))
        ctx.save_for_backward(dl)
        ctx.M = logits_m.shape[0]
        return logits, loss

    @staticmethod
    def backward(ctx, dlogits, dloss):
        (dl,) = ctx.saved_tensors
        g = dl * dloss
        if dlogits is not None:
            g = g + (dlogits / ctx.M).unsqueeze(0)
        return g, None, None


# ------------------------------------------------------------------------------------------
# fine-grained Functions for the stand-alone module facades (PreNorm / Attention / FeedForward /
# CrossAttention / Mlp / MultiHeadAttention used outside their parent block)
# ------------------------------------------------------------------------------------------


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        shape = x.shape
        x2 = _f32c(x).reshape(-1, shape[-1])
        y, mu, rs = ops.layernorm_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, mu, rs, w)
        ctx.shape = shape
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mu, rs, w = ctx.saved_tensors
        d = x2.shape[1]
        dyb = dy.reshape(-1, d)
        dyb = dyb if dyb.dtype == torch.bfloat16 else ops.cast_bf16(_f32c(dyb))
        dg, db = _zeros(d, x2), _zeros(d, x2)
        dx, _ = ops.layernorm_bwd(dyb.contiguous(), x2, mu, rs, w, dg, db)
        return dx.reshape(ctx.shape), dg, db, None


def _as_bf16_2d(x):
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype == torch.bfloat16:
        return x2.contiguous()
    return ops.cast_bf16(_f32c(x2))


class LinearFn(Function):
    """y = x W^T + b; x any float dtype (cast to bf16), y bf16 or fp32."""

    @staticmethod
    def forward(ctx, x, w, b, out_f32, p=0.0):
        x2 = _as_bf16_2d(x)
        w_s = SHADOWS.get(w)
        seed = drop_seeds(1)[0] if p > 0.0 else 0
        y = _linear(x2, w_s, bias=b, out_dtype=torch.float32 if out_f32 else torch.bfloat16, dropout=_dp(p, seed))
        ctx.save_for_backward(x2, w_s)
        ctx.meta = (x.shape, x.dtype, b is not None, p, seed)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w_s = ctx.saved_tensors
        shape, xdtype, has_b, p, seed = ctx.meta
        dyb = _as_bf16_2d(dy)
        if p > 0.0:
            dyb = ops.dropout(dyb, p, seed)
        dx = _dgrad(dyb, w_s).reshape(shape)
        dW = _wgrad(dyb, x2)
        _join_wgrads(dyb.device)
        return dx.to(xdtype), dW, (ops.colsum(dyb) if has_b else None), None, None


class FeedForwardFn(Function):
    """Linear -> exact GELU -> Linear (model_cross.py:19-31, model.py:107-122), fp32 out."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p=0.0):
        x2 = _as_bf16_2d(x)
        w1_s, w2_s = SHADOWS.get(w1), SHADOWS.get(w2)
        z = torch.empty(x2.shape[0], w1.shape[0], dtype=torch.bfloat16, device=x.device)
        seeds = drop_seeds(2) if p > 0.0 else (0, 0)
        a = _linear(x2, w1_s, bias=b1, act=ops.ACT_GELU, aux=z, dropout=_dp(p, seeds[0]), aux_mode=AUX_MODE)
        y = _linear(a, w2_s, bias=b2, out_dtype=torch.float32, dropout=_dp(p, seeds[1]))
        ctx.save_for_backward(x2, z, a, w1_s, w2_s)
        ctx.meta = (x.shape, x.dtype, p, seeds)
        return y.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, z, a, w1_s, w2_s = ctx.saved_tensors
        shape, xdtype, p, seeds = ctx.meta
        dyb = _as_bf16_2d(dy)
        if p > 0.0:
            dyb = ops.dropout(dyb, p, seeds[1])
        dz = _dgrad(dyb, w2_s, act=ops.ACT_DGELU, aux=z, dropout=_dp(p, seeds[0]), aux_mode=AUX_MODE)
        dx = _dgrad(dz, w1_s).reshape(shape)
        dW1, dW2 = _wgrad(dz, x2), _wgrad(dyb, a)
        _join_wgrads(dyb.device)
        return dx.to(xdtype), dW1, ops.colsum(dz), dW2, ops.colsum(dyb), None


class AttentionCoreFn(Function):
    """softmax(q k^T scale) v on a fused [B, N, 3d] qkv tensor."""

    @staticmethod
    def forward(ctx, qkv, H, scale, p=0.0):
        B, N, d3 = qkv.shape
        q2 = _as_bf16_2d(qkv)
        seed = drop_seeds(1)[0] if p > 0.0 else 0
        o, lse = _attn_fwd(q2, B, N, H, scale, p, seed)
        ctx.save_for_backward(q2, o, lse)
        ctx.meta = (B, N, H, scale, qkv.dtype, p, seed)
        return o.reshape(B, N, d3 // 3)

    @staticmethod
    def backward(ctx, do):
        q2, o, lse = ctx.saved_tensors
        B, N, H, scale, dt, p, seed = ctx.meta
        dqkv = ops.attn_bwd(q2, o, _as_bf16_2d(do), lse, B, N, H, scale, dropout=(p, seed))
        return dqkv.reshape(B, N, -1).to(dt), None, None, None


class ClsAttentionCoreFn(Function):
    """one CLS query per (b, h) against N keys: q [B, d], kv [B, N, 2d] -> [B, d]."""

    @staticmethod
    def forward(ctx, q, kv, H, scale, pd=0.0):
        B, N, d2 = kv.shape
        qb, kvb = _as_bf16_2d(q), _as_bf16_2d(kv)
        seed = drop_seeds(1)[0] if pd > 0.0 else 0
        o, p = ops.cls_xattn_fwd(qb, kvb, B, N, H, scale, dropout=(pd, seed))
        ctx.save_for_backward(qb, kvb, p)
        ctx.meta = (B, N, H, scale, q.dtype, kv.dtype, pd, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        qb, kvb, p = ctx.saved_tensors
        B, N, H, scale, qdt, kvdt, pd, seed = ctx.meta
        dq, dkv = ops.cls_xattn_bwd(qb, kvb, p, _as_bf16_2d(do), B, N, H, scale, dropout=(pd, seed))
        return dq.to(qdt), dkv.reshape(B, N, -1).to(kvdt), None, None, None
