"""Pin the oracle: run the *reference's own code* on the build's seeded weights/inputs,
check ``oracle/ref_cpu.py`` against it, and write golden vectors to ``tests/golden/``.

Runs ONLY in the build container (needs /root/reference; never on the GPU box).
    python oracle/make_golden.py            # check + (re)write fixtures
    python oracle/make_golden.py --check    # check only

How the reference is executed:
  * ``model_cross.py`` imports three packages this image lacks (lightning, ml_collections,
    torchmetrics).  Before importing it we register three in-memory placeholder modules
    that provide only a base class (``LightningModule := nn.Module`` + no-op ``log``), an
    attribute-dict config type and six unused metric names.  No arithmetic on the hot
    path comes from them: every line that computes is the reference's, run on CPU torch.
  * ``modelv3.py`` imports torchvision (absent): its five classes on the path (PreNorm, FeedForward, Attention,
    Transformer, ModelVIT) are exec'd from the AST with ``StochasticDepth`` bound to an identity module — the
    reference constructs every instance with rate 0 (modelv3.py:73), for which torchvision's own forward
    returns its input unchanged.
  * ``model.py`` cannot be imported at all (module-level dataset construction), so its
    four encoder classes are taken as AST ``ClassDef`` nodes and exec'd in a namespace
    that provides torch/nn/math/copy — again the reference's own statements.
Nothing is written under /root/reference and no reference source is copied here: the
fixtures hold inputs' hashes and the reference's numeric outputs only.
"""
from __future__ import annotations

import argparse
import ast
import copy
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_cpu as R  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---------------------------------------------------------------------------- reference
def import_reference():
    sys.dont_write_bytecode = True
    L = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def log(self, *a, **k):
            pass

    L.LightningModule = LightningModule
    mc = types.ModuleType("ml_collections")

    class ConfigDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    mc.ConfigDict = ConfigDict
    tm = types.ModuleType("torchmetrics")
    tm.functional = types.ModuleType("torchmetrics.functional")
    tmc = types.ModuleType("torchmetrics.classification")
    for n in ("BinaryAccuracy", "BinaryPrecision", "BinaryRecall", "BinarySpecificity",
              "BinaryF1Score", "BinaryConfusionMatrix"):
        setattr(tmc, n, object)
    tm.classification = tmc
    sys.modules.update({"lightning": L, "ml_collections": mc, "torchmetrics": tm,
                        "torchmetrics.functional": tm.functional,
                        "torchmetrics.classification": tmc})
    sys.path.insert(0, REF)
    import model_cross  # the reference module

    ns = dict(torch=torch, nn=nn, math=math, copy=copy, Dropout=nn.Dropout,
              Softmax=nn.Softmax, Linear=nn.Linear, LayerNorm=nn.LayerNorm)
    with open(os.path.join(REF, "model.py")) as fh:
        tree = ast.parse(fh.read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in {"Mlp", "MultiHeadAttention", "Block", "Encoder"}:
            exec(compile(ast.Module([node], []), "model.py", "exec"), ns)
    from einops import rearrange, repeat

    class StochasticDepth(nn.Module):          # p = 0 everywhere in the reference -> identity
        def __init__(self, p, mode):
            super().__init__()
            assert p == 0.0
        def forward(self, x):
            return x

    ns3 = dict(torch=torch, nn=nn, F=torch.nn.functional, L=L, rearrange=rearrange, repeat=repeat, StochasticDepth=StochasticDepth)
    with open(os.path.join(REF, "modelv3.py")) as fh:
        tree3 = ast.parse(fh.read())
    for node in tree3.body:
        if isinstance(node, ast.ClassDef) and node.name in {"PreNorm", "FeedForward", "Attention", "Transformer", "ModelVIT"}:
            exec(compile(ast.Module([node], []), "modelv3.py", "exec"), ns3)
    ns["ModelVIT"] = ns3["ModelVIT"]
    return model_cross, ns, ConfigDict


def to_ref_config(cfg, ConfigDict):
    c = ConfigDict()
    for k, v in vars(cfg).items():
        c[k] = v
    return c


# ------------------------------------------------------------------------------ helpers
sample_idx = R.sample_idx


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def grad_summary(grads: dict, k=16):
    out = {}
    for i, (name, gr) in enumerate(sorted(grads.items())):
        flat = gr.reshape(-1)
        out[f"gnorm/{name}"] = np.float64(flat.double().norm().item())
        out[f"gsamp/{name}"] = flat[sample_idx(flat.numel(), k, 7919 + i)].numpy()
    return out


class Checker:
    def __init__(self):
        self.worst = 0.0
        self.rows = []

    def close(self, what, mine, ref, tol):
        if what.endswith("wk.bias"):
            # d(loss)/d(wk.bias) is identically zero (softmax is invariant to a constant added
            # to every key's score), so both sides hold round-off noise: compare absolutely
            assert float(mine.abs().max()) < 1e-5 and float(ref.abs().max()) < 1e-5, what
            return
        e = rel_err(mine, ref)
        self.worst = max(self.worst, e)
        self.rows.append((what, e))
        assert e <= tol, f"restatement != reference for {what}: rel {e:.3e} > {tol}"


# ------------------------------------------------------------------ ModelCross goldens
def golden_model_cross(Mref, ConfigDict, name, batch, chk: Checker, full: bool, **over):
    cfg = R.make_config(name, **over)
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, batch, seed=0)
    model = Mref.ModelCross(to_ref_config(cfg, ConfigDict))
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.train()  # dropout p = 0
    caps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, b=b: caps.__setitem__(f"msb{b}", [t.detach() for t in o]))
             for b, blk in enumerate(model.transformer)]
    logits, loss = model(img, labels)
    loss.backward()
    for h in hooks:
        h.remove()
    ref_grads = {k: p.grad.detach() for k, p in model.named_parameters()}

    # restatement vs reference (fp32, same op order up to BLAS blocking)
    mine_cap = {}
    my_logits, my_loss = R.model_cross_forward(sd, img, labels, cfg, capture=mine_cap)
    chk.close(f"{name}/logits", my_logits, logits.detach(), 2e-5)
    chk.close(f"{name}/loss", my_loss, loss.detach(), 2e-6)
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            chk.close(f"{name}/msb{b}/mod{m}", mine_cap[f"msb{b}"][m], caps[f"msb{b}"][m], 2e-5)
    _, _, my_grads = R.model_cross_loss_and_grads(sd, img, labels, cfg)
    assert set(my_grads) == set(ref_grads)
    for k in ref_grads:
        chk.close(f"{name}/grad/{k}", my_grads[k], ref_grads[k], 5e-4)

    out = dict(batch=np.int64(batch), logits=logits.detach().numpy(), loss=np.float64(loss.item()),
               labels=labels.numpy(), img_sha256=R.tensor_sha256(img),
               sd_sha256=R.tensor_sha256(torch.cat([v.reshape(-1) for _, v in sorted(sd.items())])))
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            t = caps[f"msb{b}"][m]
            out[f"msb{b}/mod{m}/cls"] = t[:, 0].numpy()
            out[f"msb{b}/mod{m}/rownorm"] = t.norm(dim=-1).numpy()
            if full:
                out[f"msb{b}/mod{m}/full"] = t.numpy()
            else:
                rows = sample_idx(t.shape[1], 8, 31 + 7 * b + m)
                out[f"msb{b}/mod{m}/rows_idx"] = rows.numpy()
                out[f"msb{b}/mod{m}/rows"] = t[:, rows].numpy()
    out.update(grad_summary(ref_grads))
    return out


# --------------------------------------------------------------------- block goldens
def golden_blocks(Mref, ConfigDict, chk: Checker):
    """Module-level vectors for every reference class on the path, at tail-heavy shapes."""
    out = {}
    cfg = R.make_config("small")  # d=256, H=4 (dh=64), f=512, 3 modalities
    rc = to_ref_config(cfg, ConfigDict)
    sd = R.make_state_dict(cfg, seed=3)
    gen = torch.Generator().manual_seed(77)
    d, H = cfg.hidden_dim, cfg.num_heads

    def sub(prefix):
        return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}

    for N in (17, 65, 130):
        x = torch.randn(2, N, d, generator=gen)
        # SelfAttentionBlock (model_cross.py:64-72)
        blk = Mref.SelfAttentionBlock(rc)
        blk.load_state_dict(sub("transformer.0.blocks.0.0"))
        xr = x.clone().requires_grad_(True)
        y = blk(xr)
        y.square().sum().backward()
        my = R.self_block(sd, "transformer.0.blocks.0.0", x, H)
        chk.close(f"SelfAttentionBlock/N{N}", my, y.detach(), 2e-5)
        out[f"sab/N{N}/x"] = x.numpy(); out[f"sab/N{N}/y"] = y.detach().numpy(); out[f"sab/N{N}/dx"] = xr.grad.numpy()
        # Attention alone (model_cross.py:33-61) on the normed input
        att = blk.attn.fn
        ya = att(x)
        chk.close(f"Attention/N{N}", R.self_attention(sd, "transformer.0.blocks.0.0.attn.fn", x, H), ya.detach(), 2e-5)
        out[f"attn/N{N}/y"] = ya.detach().numpy()
        # FeedForward (model_cross.py:19-31)
        yf = blk.ffn.fn(x)
        chk.close(f"FeedForward/N{N}", R.feed_forward(sd, "transformer.0.blocks.0.0.ffn.fn", x), yf.detach(), 2e-5)
        out[f"ffn/N{N}/y"] = yf.detach().numpy()
        # CrossAttentionBlock + CrossAttention (model_cross.py:74-114)
        cab = Mref.CrossAttentionBlock(rc)
        cab.load_state_dict(sub("transformer.0.fusion.0"))
        xr = x.clone().requires_grad_(True)
        yc = cab(xr)
        yc.square().sum().backward()
        chk.close(f"CrossAttentionBlock/N{N}", R.cross_block(sd, "transformer.0.fusion.0", x, H), yc.detach(), 2e-5)
        out[f"cab/N{N}/y"] = yc.detach().numpy(); out[f"cab/N{N}/dx"] = xr.grad.numpy()
        yx = cab.attn.fn(x)
        chk.close(f"CrossAttention/N{N}", R.cls_cross_attention(sd, "transformer.0.fusion.0.attn.fn", x, H), yx.detach(), 2e-5)
        out[f"xattn/N{N}/y"] = yx.detach().numpy()

    # MultiScaleBlock (model_cross.py:116-148) with the 3-ring
    msb = Mref.MultiScaleBlock(rc)
    msb.load_state_dict(sub("transformer.0"))
    xs = [torch.randn(2, 65, d, generator=gen) for _ in range(3)]
    ys = msb([t.clone() for t in xs])
    mine = R.multi_scale_block(sd, "transformer.0", xs, cfg)
    for m in range(3):
        chk.close(f"MultiScaleBlock/mod{m}", mine[m], ys[m].detach(), 2e-5)
        out[f"msb/x{m}"] = xs[m].numpy(); out[f"msb/y{m}"] = ys[m].detach().numpy()

    # patchify index map (model_cross.py:193) on a non-cubic volume, via the reference's einops call
    from einops import rearrange
    vol = torch.arange(2 * 1 * 8 * 12 * 6, dtype=torch.float32).reshape(2, 1, 8, 12, 6)
    pr = rearrange(vol, "b c (d p1) (h p2) (w p3) -> b (h w d) (p1 p2 p3 c)", p1=4, p2=3, p3=2)
    assert torch.equal(R.patchify(vol[:, 0], (4, 3, 2)), pr)
    out["patchify/out"] = pr.numpy()
    out["sd_sha256"] = R.tensor_sha256(torch.cat([v.reshape(-1) for _, v in sorted(sd.items())]))
    return out


def golden_encoder(ns, ConfigDict, chk: Checker):
    """model.py Encoder (model.py:203-214)."""
    hidden, mlp, H, layers = 256, 512, 4, 2
    sd = R.make_encoder_state_dict(hidden, mlp, layers, seed=5)
    enc = ns["Encoder"](ConfigDict(hidden_size=hidden, transformer=dict(
        num_heads=H, mlp_dim=mlp, dropout_rate=0.0, attention_dropout_rate=0.0, num_layers=layers)))
    enc.load_state_dict(sd, strict=True)
    out = {}
    gen = torch.Generator().manual_seed(99)
    for N in (65, 130):
        x = torch.randn(2, N, hidden, generator=gen)
        xr = x.clone().requires_grad_(True)
        y = enc(xr)
        enc.zero_grad()
        y.square().sum().backward()
        chk.close(f"Encoder/N{N}", R.encoder_forward(sd, x, H, layers), y.detach(), 2e-5)
        out[f"N{N}/x"] = x.numpy(); out[f"N{N}/y"] = y.detach().numpy(); out[f"N{N}/dx"] = xr.grad.numpy()
        if N == 65:
            out.update(grad_summary({k: p.grad.detach().clone() for k, p in enc.named_parameters()}))
    out["sd_sha256"] = R.tensor_sha256(torch.cat([v.reshape(-1) for _, v in sorted(sd.items())]))
    return out


def golden_model_vit(ns, ConfigDict, chk: Checker):
    """modelv3.ModelVIT at a tail-heavy small shape (3 modalities -> N = 3*16 + 1 = 49... with config "small": 3*16+1)."""
    out = {}
    cfg = R.make_config("small", num_layers=2)
    sd = R.make_vit_state_dict(cfg, seed=11)
    img, labels = R.make_inputs(cfg, 3, seed=4)
    model = ns["ModelVIT"](to_ref_config(cfg, ConfigDict))
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.train()
    logits, loss = model(img, labels)
    loss.backward()
    cap = {}
    my_logits, my_loss = R.model_vit_forward(sd, img, labels, cfg, capture=cap)
    chk.close("ModelVIT/logits", my_logits, logits.detach(), 2e-5)
    chk.close("ModelVIT/loss", my_loss, loss.detach(), 2e-6)
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    R.model_vit_forward(leaf, img, labels, cfg)[1].backward()
    ref_grads = {k: p.grad.detach() for k, p in model.named_parameters()}
    for k in ref_grads:
        chk.close(f"ModelVIT/grad/{k}", leaf[k].grad, ref_grads[k], 5e-4)
    out.update(logits=logits.detach().numpy(), loss=np.float64(loss.item()), labels=labels.numpy(), img_sha256=R.tensor_sha256(img),
               sd_sha256=R.tensor_sha256(torch.cat([v.reshape(-1) for _, v in sorted(sd.items())])))
    out.update(grad_summary(ref_grads))
    return out


PARTIAL = dict(num_modalities=3, attn_order={"0": "1", "1": "2"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="verify only; do not write fixtures")
    ap.add_argument("--skip-base", action="store_true", help="skip the 93M-parameter config[1] case and the 242M-parameter reference-default case")
    ap.add_argument("--only", action="append", default=[], help="make only this fixture file (repeatable), e.g. --only model_cross_mist.npz")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    Mref, ns, ConfigDict = import_reference()
    chk = Checker()
    makers = {
        "blocks.npz": lambda: golden_blocks(Mref, ConfigDict, chk),
        "encoder.npz": lambda: golden_encoder(ns, ConfigDict, chk),
        "model_vit_small.npz": lambda: golden_model_vit(ns, ConfigDict, chk),
        "model_cross_tiny.npz": lambda: golden_model_cross(Mref, ConfigDict, "tiny", 4, chk, full=True),
        "model_cross_small.npz": lambda: golden_model_cross(Mref, ConfigDict, "small", 2, chk, full=True),
    }
    # the reference's second run setting (main_mist.py:72): three modalities, only two of them fused (0 <- 1, 1 <- 2; modality 2 passes through)
    makers["model_cross_partial.npz"] = lambda: golden_model_cross(Mref, ConfigDict, "tiny", 3, chk, full=True, **PARTIAL)
    if not args.skip_base:
        makers["model_cross_base.npz"] = lambda: golden_model_cross(Mref, ConfigDict, "base", 2, chk, full=False)
        # the reference's own run shape (config2.py:5-22, main_mist.py:71): d = 1024, 16 heads, 3-ring, 16 x 16 x 8 patches
        makers["model_cross_mist.npz"] = lambda: golden_model_cross(Mref, ConfigDict, "mist", 2, chk, full=False)
    unknown = [fn for fn in args.only if fn not in makers]
    assert not unknown, f"unknown fixture(s) {unknown}; have {sorted(makers)}"
    files = {fn: make() for fn, make in makers.items() if not args.only or fn in args.only}
    print(f"restatement == reference on {len(chk.rows)} tensors; worst rel-L2 {chk.worst:.3e}")
    if not args.check:
        os.makedirs(GOLD, exist_ok=True)
        for fn, d in files.items():
            np.savez_compressed(os.path.join(GOLD, fn), **d)
            print("wrote", fn, f"{os.path.getsize(os.path.join(GOLD, fn)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
