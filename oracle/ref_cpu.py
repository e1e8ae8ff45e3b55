"""CPU oracle: a from-scratch functional restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may call it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own modules
(``/root/reference/model_cross.py`` and the four encoder classes of ``model.py``) in the
build container, loads the seeded weights generated here, and (a) asserts this
restatement equals the reference to fp32 round-off for outputs *and* parameter
gradients, (b) writes the reference's outputs to ``tests/golden/*.npz``.
``tests/test_oracle.py`` re-checks this file against those fixtures on every run.

Everything is written as pure functions over a flat ``dict[str, Tensor]`` that uses the
reference's ``state_dict`` key names, so a reference checkpoint drives it unchanged.
Works in fp32 or fp64 (dtype follows the inputs); gradients come from torch autograd
over these functions.

Reference lines restated (relative to /root/reference):
  patchify            model_cross.py:193      layer_norm        model_cross.py:14-17
  patch embed/cls/pos model_cross.py:194-197  feed_forward      model_cross.py:19-31
  self_attention      model_cross.py:50-61    self_block        model_cross.py:69-72
  cls_cross_attention model_cross.py:88-102   cross_block       model_cross.py:111-114
  multi_scale_block   model_cross.py:128-148  model_cross_fwd   model_cross.py:186-212
  encoder (model.py)  model.py:107-214
"""
from __future__ import annotations

import hashlib
import math
from types import SimpleNamespace

import torch

# --------------------------------------------------------------------------------------
# configs (BASELINE.json `configs`, SURVEY.md §8(a) shape ladder)
# --------------------------------------------------------------------------------------


def make_config(name: str = "tiny", **over) -> SimpleNamespace:
    """Duck-typed config carrying the attribute names the reference reads
    (config2.py:3-28 + main_mist.py:59-77)."""
    base = dict(
        num_classes=2, dropout=0.0, lr=1e-4, weight_decay=0.0,
        optim_params={"T_max": 1, "eta_min": 0.0}, label_smoothing=0.0,
        num_modalities=2, attn_order={"0": "1", "1": "0"},
    )
    table = {
        # config[0]: CPU-plumbing shape
        "tiny": dict(hidden_dim=192, mlp_dim=768, num_heads=3, num_multi_blocks=2,
                     num_self_blocks=2, img_size=(32, 32, 2), patch_size=(8, 8, 2)),
        # a mid-size shape with tails everywhere (N=65), two MSBs, 3-ring of modalities
        "small": dict(hidden_dim=256, mlp_dim=512, num_heads=4, num_multi_blocks=2,
                      num_self_blocks=1, img_size=(32, 32, 16), patch_size=(8, 8, 8),
                      num_modalities=3, attn_order={"0": "1", "1": "2", "2": "0"}),
        # config[1]/[3]: headline shape
        "base": dict(hidden_dim=768, mlp_dim=3072, num_heads=12, num_multi_blocks=2,
                     num_self_blocks=2, img_size=(128, 128, 128), patch_size=(16, 16, 16)),
        # config[2]: UCSF-PDGM shape, 4-ring
        "ucsf": dict(hidden_dim=768, mlp_dim=3072, num_heads=12, num_multi_blocks=2,
                     num_self_blocks=2, img_size=(240, 240, 240), patch_size=(16, 16, 16),
                     num_modalities=4, attn_order={"0": "1", "1": "2", "2": "3", "3": "0"}),
        # the reference's own run shape (config2.py:5-22 + main_mist.py:71): 3-ring, thin W patches (8-voxel runs), d_h = 64 at 16 heads
        "mist": dict(hidden_dim=1024, mlp_dim=4096, num_heads=16, num_multi_blocks=2,
                     num_self_blocks=2, img_size=(128, 128, 64), patch_size=(16, 16, 8),
                     num_modalities=3, attn_order={"0": "1", "1": "2", "2": "0"}),
        # config[4]: long sequence
        "long": dict(hidden_dim=768, mlp_dim=3072, num_heads=12, num_multi_blocks=2,
                     num_self_blocks=2, img_size=(128, 128, 128), patch_size=(8, 8, 8)),
    }
    base.update(table[name])
    base.update(over)
    return SimpleNamespace(**base)


def derived(cfg):
    D, H, W = cfg.img_size
    dp, hp, wp = cfg.patch_size
    P = (D // dp) * (H // hp) * (W // wp)
    return SimpleNamespace(P=P, N=P + 1, pd=dp * hp * wp, d=cfg.hidden_dim, f=cfg.mlp_dim,
                           H=cfg.num_heads, dh=cfg.hidden_dim // cfg.num_heads,
                           M=cfg.num_modalities, n_cross=len(cfg.attn_order))


def flops_per_sample(cfg):
    """Closed-form matmul FLOPs; returns (fwd, fwd+bwd)."""
    g = derived(cfg)
    PE = 2 * g.M * g.P * g.pd * g.d
    SAB = 2 * g.N * g.d * 3 * g.d + 2 * (2 * g.N * g.N * g.d) + 2 * g.N * g.d * g.d + 2 * (2 * g.N * g.d * g.f)
    CAB = 2 * (2 * g.N * g.d * g.d) + 2 * (2 * g.d * g.d) + 2 * (2 * g.N * g.d) + 2 * (2 * g.d * g.f)
    HEAD = g.M * (2 * g.d * g.f + 2 * g.f * cfg.num_classes)
    fwd = PE + cfg.num_multi_blocks * (g.M * cfg.num_self_blocks * SAB + g.n_cross * CAB) + HEAD
    return fwd, 3 * fwd - PE


# --------------------------------------------------------------------------------------
# seeded weights / inputs (the build's own generators; the reference only *loads* them)
# --------------------------------------------------------------------------------------


def _xavier(gen, out_f, in_f):
    bound = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen, dtype=torch.float32) * 2 - 1) * bound


def _linear(sd, gen, key, out_f, in_f, bias=True, bias_scale=0.02):
    sd[key + ".weight"] = _xavier(gen, out_f, in_f)
    if bias:
        # the reference zero-inits biases (model_cross.py:222-223); parity tests use small
        # random ones so that a dropped bias term cannot hide
        sd[key + ".bias"] = torch.randn(out_f, generator=gen) * bias_scale


def _norm(sd, gen, key, d):
    sd[key + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=gen)
    sd[key + ".bias"] = 0.05 * torch.randn(d, generator=gen)


def make_state_dict(cfg, seed: int = 0) -> dict:
    """ModelCross parameters under the reference's state_dict names."""
    g = derived(cfg)
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    sd["pos_embedding"] = 0.02 * torch.randn(1, g.N, g.d, generator=gen)
    sd["cls_token"] = 0.02 * torch.randn(1, 1, g.d, generator=gen)
    _linear(sd, gen, "patch_to_embedding", g.d, g.pd)
    for b in range(cfg.num_multi_blocks):
        for m in range(g.M):
            for s in range(cfg.num_self_blocks):
                p = f"transformer.{b}.blocks.{m}.{s}"
                _norm(sd, gen, p + ".attn.norm", g.d)
                _linear(sd, gen, p + ".attn.fn.to_qkv", 3 * g.d, g.d, bias=False)
                _linear(sd, gen, p + ".attn.fn.to_out.0", g.d, g.d)
                _norm(sd, gen, p + ".ffn.norm", g.d)
                _linear(sd, gen, p + ".ffn.fn.net.0", g.f, g.d)
                _linear(sd, gen, p + ".ffn.fn.net.3", g.d, g.f)
        for k in range(g.n_cross):
            p = f"transformer.{b}.fusion.{k}"
            _norm(sd, gen, p + ".attn.norm", g.d)
            for w in ("wq", "wk", "wv", "proj"):
                _linear(sd, gen, f"{p}.attn.fn.{w}", g.d, g.d)
            _norm(sd, gen, p + ".ffn.norm", g.d)
            _linear(sd, gen, p + ".ffn.fn.net.0", g.f, g.d)
            _linear(sd, gen, p + ".ffn.fn.net.3", g.d, g.f)
    for m in range(g.M):
        _norm(sd, gen, f"norm.{m}", g.d)
        _linear(sd, gen, f"mlp_head.{m}.0", g.f, g.d)
        _linear(sd, gen, f"mlp_head.{m}.3", cfg.num_classes, g.f)
    return sd


def make_encoder_state_dict(hidden, mlp, layers, seed: int = 0) -> dict:
    """model.py Encoder parameters (model.py:203-214 names)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for l in range(layers):
        p = f"layers.{l}"
        _norm(sd, gen, p + ".attention_norm", hidden)
        _norm(sd, gen, p + ".ffn_norm", hidden)
        for w in ("query", "key", "value", "out"):
            _linear(sd, gen, f"{p}.multi_head.{w}", hidden, hidden)
        _linear(sd, gen, p + ".ffn.fc1", mlp, hidden)
        _linear(sd, gen, p + ".ffn.fc2", hidden, mlp)
    _norm(sd, gen, "encoder_norm", hidden)
    return sd


def make_inputs(cfg, batch: int, seed: int = 0):
    gen = torch.Generator().manual_seed(1000 + seed)
    g = derived(cfg)
    img = torch.randn(batch, g.M, 1, *cfg.img_size, generator=gen)
    labels = torch.randint(0, cfg.num_classes, (batch,), generator=gen)
    return img, labels


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round-trip through bf16: the operand quantisation the HIP path applies."""
    return t.to(torch.bfloat16).to(t.dtype) if t.is_floating_point() else t


def sample_idx(numel: int, k: int, seed: int) -> torch.Tensor:
    """Deterministic sample positions shared by the fixture writer and the tests."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(k, numel),), generator=g)


def tensor_sha256(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


# --------------------------------------------------------------------------------------
# optional emulation of the HIP pipeline's storage precision
# --------------------------------------------------------------------------------------
# By default every function below is the exact fp32/fp64 restatement (the form pinned against the
# reference).  Inside `with emulate_bf16():` operands are additionally rounded to bf16 at exactly
# the points where the HIP path stores or feeds bf16: both operands of every GEMM over ALL tokens, the
# q/k/v projections' outputs, and the (un-normalised) softmax probabilities fed to the P.V product.
# Accumulation stays fp32, the residual stream stays fp32 — the arithmetic class of the kernels.
# The single-token CLS path (wq, proj and the FFN of the cross fusion, the heads: `exact=True` below)
# runs fp32 operands on the HIP side as well (xvit_linear_f32) and is therefore not rounded here.
# GPU-vs-emulation isolates kernel bugs from the (expected) cost of bf16 operands.

_QUANT = None


class emulate_bf16:
    def __enter__(self):
        global _QUANT
        self._prev, _QUANT = _QUANT, bf16_round
        return self

    def __exit__(self, *exc):
        global _QUANT
        _QUANT = self._prev
        return False


def _q(t):
    return t if _QUANT is None else _QUANT(t)


# --------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------


def patchify(vol: torch.Tensor, patch) -> torch.Tensor:
    """[B, D, H, W] -> [B, P, pd].  Token t = (h*Wn + w)*Dn + d (h-major), feature
    f = (p1*hp + p2)*wp + p3 — the index map of the einops pattern at model_cross.py:193
    with c == 1, restated as an explicit reshape/permute."""
    B, D, H, W = vol.shape
    dp, hp, wp = patch
    Dn, Hn, Wn = D // dp, H // hp, W // wp
    v = vol.reshape(B, Dn, dp, Hn, hp, Wn, wp)
    v = v.permute(0, 3, 5, 1, 2, 4, 6)  # b, h, w, d, p1, p2, p3
    return v.reshape(B, Hn * Wn * Dn, dp * hp * wp)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)  # biased, as nn.LayerNorm
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, w, b=None, store=False, exact=False):
    """y = x W^T + b.  Emulation mode rounds x and W (unless exact: the fp32 single-token path) and,
    when `store`, the result (outputs the HIP path keeps in bf16)."""
    if exact:
        y = x @ w.transpose(-1, -2)
        return y if b is None else y + b
    y = _q(x) @ _q(w).transpose(-1, -2)
    y = y if b is None else y + b
    return _q(y) if store else y


def _split_heads(t, H):
    B, N, d = t.shape
    return t.reshape(B, N, H, d // H).permute(0, 2, 1, 3)


def _merge_heads(t):
    B, H, N, dh = t.shape
    return t.permute(0, 2, 1, 3).reshape(B, N, H * dh)


def softmax_attention(q, k, v, scale):
    """q [B,H,Nq,dh], k/v [B,H,Nk,dh] -> ([B,H,Nq,dh], lse [B,H,Nq])."""
    s = (q @ k.transpose(-1, -2)) * scale
    m = s.amax(dim=-1, keepdim=True)
    e = torch.exp(s - m)
    z = e.sum(dim=-1, keepdim=True)
    if _QUANT is not None and q.shape[-2] > 1:   # flash kernel: bf16 P into the MFMA, fp32 row sum
        return (_q(e) @ v) / z, (m + torch.log(z)).squeeze(-1)
    return (e / z) @ v, (m + torch.log(z)).squeeze(-1)


# --------------------------------------------------------------------------------------
# model_cross.py blocks
# --------------------------------------------------------------------------------------


def feed_forward(sd, p, x, exact=False):
    """model_cross.py:19-31 (dropout p=0)."""
    h = gelu(linear(x, sd[p + ".net.0.weight"], sd[p + ".net.0.bias"], exact=exact))
    return linear(h, sd[p + ".net.3.weight"], sd[p + ".net.3.bias"], exact=exact)


def self_attention(sd, p, x, H):
    """model_cross.py:50-61: bias-free fused qkv, scale = dh**-0.5, biased out-proj."""
    d = x.shape[-1]
    qkv = linear(x, sd[p + ".to_qkv.weight"], store=True)
    q, k, v = (_split_heads(t, H) for t in qkv.split(d, dim=-1))
    o, _ = softmax_attention(q, k, v, (d // H) ** -0.5)
    return linear(_merge_heads(o), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def self_block(sd, p, x, H):
    """model_cross.py:69-72."""
    x = self_attention(sd, p + ".attn.fn", layer_norm(x, sd[p + ".attn.norm.weight"], sd[p + ".attn.norm.bias"]), H) + x
    x = feed_forward(sd, p + ".ffn.fn", layer_norm(x, sd[p + ".ffn.norm.weight"], sd[p + ".ffn.norm.bias"])) + x
    return x


def cls_cross_attention(sd, p, x, H):
    """model_cross.py:88-102: the query is row 0 only; keys/values are all N rows."""
    d = x.shape[-1]
    if _QUANT is not None and d == 64 * H and H <= 16:
        return _cls_cross_attention_lowrank_emulated(sd, p, x, H)
    q = _split_heads(linear(x[:, 0:1], sd[p + ".wq.weight"], sd[p + ".wq.bias"], exact=True), H)
    k = _split_heads(linear(x, sd[p + ".wk.weight"], sd[p + ".wk.bias"], store=True), H)
    v = _split_heads(linear(x, sd[p + ".wv.weight"], sd[p + ".wv.bias"], store=True), H)
    o, _ = softmax_attention(q, k, v, (d // H) ** -0.5)
    return linear(_merge_heads(o), sd[p + ".proj.weight"], sd[p + ".proj.bias"], exact=True)


def _cls_cross_attention_lowrank_emulated(sd, p, x, H):
    """Emulation mode only: the rounding points of the HIP path's low-rank form (csrc/head_linear.hip).  Same function as
    cls_cross_attention in exact arithmetic — scores[n] = q_h . (Wk_h x[n] + bk_h) = x[n] . (q_h Wk_h) + const, and
    sum_n p[n] (Wv_h x[n] + bv_h) = Wv_h (sum_n p[n] x[n]) + bv_h — but K and V are never formed (so never rounded):
    bf16 are the normed tokens x, the H vectors U_h = q_h Wk_h and the un-normalised softmax weights; wq, wk, wv, proj run fp32."""
    B, N, d = x.shape
    dh = d // H
    xq = _q(x)
    q = linear(x[:, 0], sd[p + ".wq.weight"], sd[p + ".wq.bias"], exact=True).reshape(B, H, dh)
    U = torch.einsum("bhe,hec->bhc", q, sd[p + ".wk.weight"].reshape(H, dh, d))
    sc = torch.einsum("bnc,bhc->bnh", xq, _q(U)) * dh ** -0.5
    e = _q(torch.exp(sc - sc.amax(dim=1, keepdim=True)))
    S = torch.einsum("bnh,bnc->bhc", e, xq) / e.sum(dim=1)[:, :, None]
    o = torch.einsum("bhc,hec->bhe", S, sd[p + ".wv.weight"].reshape(H, dh, d)).reshape(B, 1, d) + sd[p + ".wv.bias"]
    return linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"], exact=True)


def cross_block(sd, p, x, H):
    """model_cross.py:111-114: LN over all N rows, residual is the un-normed row 0."""
    y = cls_cross_attention(sd, p + ".attn.fn", layer_norm(x, sd[p + ".attn.norm.weight"], sd[p + ".attn.norm.bias"]), H) + x[:, 0:1]
    y = feed_forward(sd, p + ".ffn.fn", layer_norm(y, sd[p + ".ffn.norm.weight"], sd[p + ".ffn.norm.bias"]), exact=True) + y
    return y


def multi_scale_block(sd, p, xs, cfg):
    """model_cross.py:128-148."""
    H = cfg.num_heads
    ys = []
    for m, x in enumerate(xs):
        for s in range(cfg.num_self_blocks):
            x = self_block(sd, f"{p}.blocks.{m}.{s}", x, H)
        ys.append(x)
    outs, k = [], 0
    for i in range(len(ys)):
        if str(i) in cfg.attn_order:
            j = int(cfg.attn_order[str(i)])
            fused = cross_block(sd, f"{p}.fusion.{k}", torch.cat((ys[i][:, 0:1], ys[j][:, 1:]), dim=1), H)
            outs.append(torch.cat((fused, ys[i][:, 1:]), dim=1))
            k += 1
        else:
            outs.append(ys[i])
    return outs


def embed(sd, img, cfg):
    """model_cross.py:191-199 -> list of M tensors [B, N, d]."""
    toks = []
    for m in range(img.shape[1]):
        x = linear(patchify(img[:, m, 0], cfg.patch_size), sd["patch_to_embedding.weight"], sd["patch_to_embedding.bias"])
        x = torch.cat((sd["cls_token"].expand(img.shape[0], -1, -1), x), dim=1)
        toks.append(x + sd["pos_embedding"])
    return toks


def cross_entropy(logits, labels, smoothing=0.0):
    logp = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
    nll = -logp.gather(1, labels[:, None]).squeeze(1)
    if smoothing > 0.0:
        nll = (1.0 - smoothing) * nll + smoothing * (-logp.mean(dim=-1))
    return nll.mean()


def model_cross_forward(sd, img, labels, cfg, capture: dict | None = None):
    """model_cross.py:186-212 -> (logits [B,C], loss).  `capture` (optional) receives the
    intermediate tensors the parity tests probe."""
    xs = embed(sd, img, cfg)
    if capture is not None:
        capture["embed"] = [x for x in xs]
    for b in range(cfg.num_multi_blocks):
        xs = multi_scale_block(sd, f"transformer.{b}", xs, cfg)
        if capture is not None:
            capture[f"msb{b}"] = [x for x in xs]
    per_mod = []
    for m, x in enumerate(xs):
        c = layer_norm(x, sd[f"norm.{m}.weight"], sd[f"norm.{m}.bias"])[:, 0]
        h = gelu(linear(c, sd[f"mlp_head.{m}.0.weight"], sd[f"mlp_head.{m}.0.bias"], exact=True))
        per_mod.append(linear(h, sd[f"mlp_head.{m}.3.weight"], sd[f"mlp_head.{m}.3.bias"], exact=True))
    logits = torch.stack(per_mod).mean(dim=0)
    return logits, cross_entropy(logits, labels, cfg.label_smoothing)


def model_cross_loss_and_grads(sd, img, labels, cfg):
    """fwd + autograd bwd; returns (logits, loss, {name: grad})."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits, loss = model_cross_forward(leaf, img, labels, cfg)
    loss.backward()
    return logits.detach(), loss.detach(), {k: v.grad for k, v in leaf.items()}


# --------------------------------------------------------------------------------------
# input stage: MONAI ResizeWithPadOrCropd(spatial_size, constant_values) + float cast
# (dataset_ucsf.py:84-88,152-158).  PARITY UNPINNED: MONAI is neither installed nor version-pinned by the
# reference (requirements.txt:4); this restates its documented rule — SpatialPad(method="symmetric"): pad
# (deficit // 2) before and the rest after; CenterSpatialCrop: start = size // 2 - roi // 2.
# --------------------------------------------------------------------------------------


def resize_with_pad_or_crop(vol: torch.Tensor, spatial_size, constant_value=-1.0) -> torch.Tensor:
    """vol [..., Ds, Hs, Ws] (any dtype) -> float32 [..., D, H, W]."""
    out = vol.to(torch.float32)
    for ax, target in zip((-3, -2, -1), spatial_size):
        size = out.shape[ax]
        if size < target:
            before = (target - size) // 2
            pad = [0, 0] * 3
            pad[2 * (-1 - ax)] = before
            pad[2 * (-1 - ax) + 1] = target - size - before
            out = torch.nn.functional.pad(out, pad, value=float(constant_value))
        elif size > target:
            start = size // 2 - target // 2
            out = out.narrow(ax, start, target)
    return out.contiguous()


# --------------------------------------------------------------------------------------
# modelv3.ModelVIT (modelv3.py:90-147): all modalities' patch tokens in ONE sequence, same blocks
# --------------------------------------------------------------------------------------


def make_vit_state_dict(cfg, seed: int = 0) -> dict:
    g = derived(cfg)
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    sd["pos_embedding"] = 0.02 * torch.randn(1, g.M * g.P + 1, g.d, generator=gen)
    sd["cls_token"] = 0.02 * torch.randn(1, 1, g.d, generator=gen)
    _linear(sd, gen, "patch_to_embedding", g.d, g.pd)
    for l in range(cfg.num_layers):
        p = f"transformer.layers.{l}"
        _norm(sd, gen, p + ".0.norm", g.d)
        _linear(sd, gen, p + ".0.fn.to_qkv", 3 * g.d, g.d, bias=False)
        _linear(sd, gen, p + ".0.fn.to_out.0", g.d, g.d)
        _norm(sd, gen, p + ".2.norm", g.d)
        _linear(sd, gen, p + ".2.fn.net.0", g.f, g.d)
        _linear(sd, gen, p + ".2.fn.net.3", g.d, g.f)
    _norm(sd, gen, "mlp_head.0", g.d)
    _linear(sd, gen, "mlp_head.1", g.f, g.d)
    _linear(sd, gen, "mlp_head.4", cfg.num_classes, g.f)
    return sd


def model_vit_forward(sd, img, labels, cfg, capture: dict | None = None):
    """modelv3.py:123-147 -> (logits, loss)."""
    H = cfg.num_heads
    toks = [linear(patchify(img[:, m, 0], cfg.patch_size), sd["patch_to_embedding.weight"], sd["patch_to_embedding.bias"]) for m in range(img.shape[1])]
    x = torch.cat([sd["cls_token"].expand(img.shape[0], -1, -1)] + toks, dim=1) + sd["pos_embedding"]
    for l in range(cfg.num_layers):
        p = f"transformer.layers.{l}"
        x = self_attention(sd, p + ".0.fn", layer_norm(x, sd[p + ".0.norm.weight"], sd[p + ".0.norm.bias"]), H) + x
        x = feed_forward(sd, p + ".2.fn", layer_norm(x, sd[p + ".2.norm.weight"], sd[p + ".2.norm.bias"])) + x
        if capture is not None:
            capture[f"layer{l}"] = x
    c = layer_norm(x[:, 0], sd["mlp_head.0.weight"], sd["mlp_head.0.bias"])
    h = gelu(linear(c, sd["mlp_head.1.weight"], sd["mlp_head.1.bias"], exact=True))
    logits = linear(h, sd["mlp_head.4.weight"], sd["mlp_head.4.bias"], exact=True)
    return logits, cross_entropy(logits, labels)


# --------------------------------------------------------------------------------------
# model.py twin (model.py:107-214): separate biased q/k/v/out, /sqrt(dh), LN eps 1e-6
# --------------------------------------------------------------------------------------


def mha(sd, p, x, H):
    """model.py:156-178."""
    q = _split_heads(linear(x, sd[p + ".query.weight"], sd[p + ".query.bias"], store=True), H)
    k = _split_heads(linear(x, sd[p + ".key.weight"], sd[p + ".key.bias"], store=True), H)
    v = _split_heads(linear(x, sd[p + ".value.weight"], sd[p + ".value.bias"], store=True), H)
    o, _ = softmax_attention(q, k, v, 1.0 / math.sqrt(x.shape[-1] // H))
    return linear(_merge_heads(o), sd[p + ".out.weight"], sd[p + ".out.bias"])


def encoder_block(sd, p, x, H):
    """model.py:190-201."""
    x = x + mha(sd, p + ".multi_head", layer_norm(x, sd[p + ".attention_norm.weight"], sd[p + ".attention_norm.bias"], 1e-6), H)
    h = layer_norm(x, sd[p + ".ffn_norm.weight"], sd[p + ".ffn_norm.bias"], 1e-6)
    h = gelu(linear(h, sd[p + ".ffn.fc1.weight"], sd[p + ".ffn.fc1.bias"]))
    return x + linear(h, sd[p + ".ffn.fc2.weight"], sd[p + ".ffn.fc2.bias"])


def encoder_forward(sd, x, H, layers):
    """model.py:211-214."""
    for l in range(layers):
        x = encoder_block(sd, f"layers.{l}", x, H)
    return layer_norm(x, sd["encoder_norm.weight"], sd["encoder_norm.bias"], 1e-6)


# ------------------------------------------------------------------------------------------
# §8(f)-4: the per-step statistics of log_stats (model_cross.py:243-255 -> utils.py:18-62)
# ------------------------------------------------------------------------------------------
METRIC_KEYS = ("acc", "prec", "rec", "spec", "f1", "npv", "auc_roc")


def binary_step_metrics(logits: torch.Tensor, labels: torch.Tensor) -> dict:
    """One step of the reference's log_stats, restated without torchmetrics (absent from this image, so parity with
    torchmetrics itself is unpinned; tests/test_oracle.py pins every value against scikit-learn instead).
      pred = argmax(logits, 1)                                   model_cross.py:244
      accuracy, precision, recall, specificity, F1 of (pred, labels) with 0 for an empty denominator
        (torchmetrics' _safe_divide)                             utils.py:33-46
      NPV = tn / (tn + fn) if tn + fn > 0 else 0                 utils.py:49-53
      AUROC of softmax(logits)[:, 1] — the exact ROC area, i.e. P(p_pos > p_neg) + 0.5 P(p_pos == p_neg); 0 when one
        class is absent from the batch (torchmetrics returns zero there)      model_cross.py:253-254
    Returns the seven values plus the confusion counts (tn, fp, fn, tp)."""
    logits = logits.detach().float().cpu()
    y = (labels.detach().cpu() != 0)
    pred = logits[:, 1] > logits[:, 0]                       # ties -> class 0 (first maximum)
    tp = int((pred & y).sum()); tn = int((~pred & ~y).sum()); fp = int((pred & ~y).sum()); fn = int((~pred & y).sum())
    div = lambda a, b: a / b if b > 0 else 0.0               # noqa: E731
    prob = torch.softmax(logits, dim=1)[:, 1]
    pos, neg = prob[y], prob[~y]
    if len(pos) == 0 or len(neg) == 0:
        auc = 0.0
    else:
        gt = (pos[:, None] > neg[None, :]).sum().item()
        eq = (pos[:, None] == neg[None, :]).sum().item()
        auc = (gt + 0.5 * eq) / (len(pos) * len(neg))
    return {"acc": div(tp + tn, tp + tn + fp + fn), "prec": div(tp, tp + fp), "rec": div(tp, tp + fn), "spec": div(tn, tn + fp),
            "f1": div(2 * tp, 2 * tp + fp + fn), "npv": div(tn, tn + fn), "auc_roc": auc, "counts": (tn, fp, fn, tp)}


def epoch_metrics(steps) -> dict:
    """What Lightning logs for `self.log(value, on_epoch=True, on_step=False)` over an epoch: the batch-size-weighted
    mean of the per-step values.  steps: iterable of (logits, labels)."""
    tot = {k: 0.0 for k in METRIC_KEYS}
    n = 0
    for logits, labels in steps:
        m = binary_step_metrics(logits, labels)
        b = int(labels.shape[0])
        for k in METRIC_KEYS:
            tot[k] += m[k] * b
        n += b
    return {k: v / max(n, 1) for k, v in tot.items()}
