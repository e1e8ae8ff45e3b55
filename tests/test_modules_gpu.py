"""GPU parity of the drop-in modules against the REFERENCE's outputs (tests/golden/*.npz, written
by oracle/make_golden.py from the reference's own code) and against the CPU oracle.

Two gates:

 (1) TIGHT — HIP path vs the CPU oracle in `emulate_bf16()` mode, i.e. the same fp32-accumulate
     arithmetic with operands rounded to bf16 at the same points (both operands of every all-token
     GEMM, stored q/k/v, P into P.V; the single-token CLS path is fp32 on both sides).  What remains
     is accumulation order and the rounding of stored activations: block outputs agree to 3e-3
     (measured 0.9-1.5e-3 at configs[1]), CLS rows to 5.5e-3 (3.8-4.6e-3), logits to 9e-3 (6.7e-3).
 (2) BUDGET — HIP path vs the REFERENCE's own fp32 outputs (goldens; un-rounded fp32 weights).
     bf16 operands alone move the result: the CPU emulation of this arithmetic measures 2.9-3.4e-3
     on block outputs, 5.2-6.3e-3 on CLS rows and 1.2e-2 on the (tiny, cancellation-heavy) logits against
     the same goldens at configs[1] — the kernels sit at exactly those distances.  Gates: full block
     outputs 6e-3, CLS rows 8e-3, logits 1.8e-2, loss 5e-3 absolute, gradients 2e-2 / norms 3 %.
     (tools/parity_probe.py prints every stage in isolation: each kernel is at 1e-5 .. 3e-5 of the
     oracle on its own inputs except flash attention, 1.3-1.8e-3 = the bf16 rounding of P.)
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import ref_cpu as R
from _util import note, dev, randn, rel

pytestmark = pytest.mark.gpu

BLOCK_TOL, GRAD_TOL = 6e-3, 2e-2


def _sub(sd, prefix):
    return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def small():
    import xvit
    cfg = R.make_config("small")
    return xvit, cfg, R.make_state_dict(cfg, seed=3)


@pytest.mark.parametrize("N", [17, 65, 130])
def test_blocks_vs_reference_golden(golden_dir, small, N):
    xvit, cfg, sd = small
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    x = _t(g[f"sab/N{N}/x"]).to(dev())

    sab = xvit.SelfAttentionBlock(cfg).to(dev())
    sab.load_state_dict(_sub(sd, "transformer.0.blocks.0.0"))
    xr = x.clone().requires_grad_()
    y = sab(xr)
    assert rel(y, _t(g[f"sab/N{N}/y"])) < BLOCK_TOL
    y.square().sum().backward()
    assert rel(xr.grad, _t(g[f"sab/N{N}/dx"])) < GRAD_TOL
    # the stand-alone sub-modules compose to the same block
    y2 = sab.attn(x) + x
    y2 = sab.ffn(y2) + y2
    assert rel(y2, _t(g[f"sab/N{N}/y"])) < BLOCK_TOL
    assert rel(sab.attn.fn(x), _t(g[f"attn/N{N}/y"])) < BLOCK_TOL
    assert rel(sab.ffn.fn(x), _t(g[f"ffn/N{N}/y"])) < BLOCK_TOL

    cab = xvit.CrossAttentionBlock(cfg).to(dev())
    cab.load_state_dict(_sub(sd, "transformer.0.fusion.0"))
    xr = x.clone().requires_grad_()
    yc = cab(xr)
    assert yc.shape == (2, 1, cfg.hidden_dim)
    assert rel(yc, _t(g[f"cab/N{N}/y"])) < BLOCK_TOL
    yc.square().sum().backward()
    assert rel(xr.grad, _t(g[f"cab/N{N}/dx"])) < GRAD_TOL
    assert rel(cab.attn.fn(x), _t(g[f"xattn/N{N}/y"])) < 1e-2   # raw attention output, no residual to lean on
    # unfused composition of the cross block == fused
    yu = cab.attn(x) + x[:, 0:1]
    yu = cab.ffn(yu) + yu
    assert rel(yu, _t(g[f"cab/N{N}/y"])) < BLOCK_TOL


def test_multiscale_block_vs_reference_golden(golden_dir, small):
    xvit, cfg, sd = small
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    msb = xvit.MultiScaleBlock(cfg).to(dev())
    msb.load_state_dict(_sub(sd, "transformer.0"))
    ys = msb([_t(g[f"msb/x{m}"]).to(dev()) for m in range(3)])
    for m in range(3):
        assert rel(ys[m], _t(g[f"msb/y{m}"])) < BLOCK_TOL


def test_encoder_vs_reference_golden(golden_dir):
    import xvit
    g = np.load(os.path.join(golden_dir, "encoder.npz"))
    cfg = SimpleNamespace(hidden_size=256, transformer=dict(num_heads=4, mlp_dim=512, dropout_rate=0.0, attention_dropout_rate=0.0, num_layers=2))
    enc = xvit.Encoder(cfg).to(dev())
    enc.load_state_dict(R.make_encoder_state_dict(256, 512, 2, seed=5))
    for N in (65, 130):
        xr = _t(g[f"N{N}/x"]).to(dev()).requires_grad_()
        y = enc(xr)
        assert rel(y, _t(g[f"N{N}/y"])) < BLOCK_TOL
        enc.zero_grad()
        y.square().sum().backward()
        assert rel(xr.grad, _t(g[f"N{N}/dx"])) < GRAD_TOL
        if N == 65:
            for k, p in enc.named_parameters():
                ref = float(g[f"gnorm/{k}"])
                if k.endswith("key.bias"):  # analytically zero (softmax shift invariance): bf16 round-off only
                    assert float(p.grad.double().norm()) < 0.05 * float(g["gnorm/" + k.replace("key", "query")]) + 1e-3
                    continue
                assert abs(float(p.grad.double().norm()) - ref) <= 0.02 * ref + 1e-6, k
    # stand-alone MultiHeadAttention + Mlp compose to Block
    blk = enc.layers[0]
    x = _t(g["N65/x"]).to(dev())
    h = x + blk.multi_head(xvit.functional.LayerNormFn.apply(x, blk.attention_norm.weight, blk.attention_norm.bias, 1e-6))
    h = h + blk.ffn(xvit.functional.LayerNormFn.apply(h, blk.ffn_norm.weight, blk.ffn_norm.bias, 1e-6))
    assert rel(h, blk(x)) < 3e-3


def _run_model(name, batch, **over):
    import xvit
    cfg = R.make_config(name, **over)
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, batch, seed=0)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(sd)
    model.train()
    caps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, b=b: caps.__setitem__(b, [t.detach() for t in o])) for b, blk in enumerate(model.transformer)]
    logits, loss = model(img.to(dev()), labels.to(dev()))
    loss.backward()
    for h in hooks:
        h.remove()
    return cfg, sd, img, labels, model, caps, logits, loss


@pytest.mark.parametrize("name,batch", [("tiny", 4), ("small", 2), ("base", 2), ("mist", 2)])   # mist: the reference's own run shape (main_mist.py:71)
def test_model_cross_vs_reference_golden(golden_dir, name, batch):
    _check_model_vs_golden(golden_dir, name, batch)


def test_model_cross_partial_fusion_map_vs_reference_golden(golden_dir):
    """The reference's second run setting (main_mist.py:72): three modalities, attn_order {0: 1, 1: 2} — modality 2 has no fusion of its own and
    the fusion modules are indexed by a running count (model_cross.py:131-147).  Fixture written by the imported reference."""
    _check_model_vs_golden(golden_dir, "tiny", 3, fixture="partial", num_modalities=3, attn_order={"0": "1", "1": "2"})


def test_model_cross_base_with_cls_peel_vs_reference_golden(golden_dir):
    """configs[1] at its bench batch runs the attention kernels in their CLS-peel form (N = 513 = 64 m + 1; include/xvit.h); the
    default heuristic keeps small grids on the tile-grid form, so force it here: same goldens, same gates."""
    from xvit import ops
    ops.set_option("attn_peel", 2)
    try:
        _check_model_vs_golden(golden_dir, "base", 2)
    finally:
        ops.set_option("attn_peel", 1)


def _check_model_vs_golden(golden_dir, name, batch, fixture=None, **over):
    g = np.load(os.path.join(golden_dir, f"model_cross_{fixture or name}.npz"))
    cfg, sd, img, labels, model, caps, logits, loss = _run_model(name, batch, **over)
    assert str(g["img_sha256"]) == R.tensor_sha256(img)  # same inputs the reference saw
    # logits: 2-class sums with heavy cancellation (|logit| ~ 0.01 .. 0.3 from a head whose terms are O(1)); measured 1.4e-2 at
    # configs[1], of which the bf16-emulating oracle shows 1.2e-2 itself (the CLS rows it feeds on differ by 5e-3 from fp32)
    # At the reference's own run shape (mist: three heads of width 4096 on 1024-wide CLS rows, |logit| 0.02 .. 0.15) the same 5.7e-3 on the
    # CLS rows becomes 2.9e-2 on the four logits; the bf16-emulating oracle sits at 3.0e-2 from this fixture itself and 1.6e-2 from the GPU
    # (tests/_probe_golden_distances.py mist 2 -> profiles/r03_parity_probe.txt).  Every other gate below is the one of configs[1].
    logit_tol = 4.5e-2 if name == "mist" else 1.8e-2
    assert rel(logits, _t(g["logits"])) < logit_tol, rel(logits, _t(g["logits"]))
    assert abs(float(loss.detach()) - float(g["loss"])) < 5e-3
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            t = caps[b][m]
            assert rel(t[:, 0], _t(g[f"msb{b}/mod{m}/cls"])) < 8e-3
            assert rel(t.norm(dim=-1), _t(g[f"msb{b}/mod{m}/rownorm"])) < 2e-3
            if f"msb{b}/mod{m}/full" in g:
                assert rel(t, _t(g[f"msb{b}/mod{m}/full"])) < BLOCK_TOL
            else:
                rows = _t(g[f"msb{b}/mod{m}/rows_idx"])
                assert rel(t[:, rows.to(dev())], _t(g[f"msb{b}/mod{m}/rows"])) < BLOCK_TOL
    # parameter gradients: every tensor's norm and 16 sampled entries
    sample_idx = R.sample_idx  # the fixture writer's own sampler
    for i, (k, p) in enumerate(sorted(model.named_parameters())):
        assert p.grad is not None, k
        ref_n = float(g[f"gnorm/{k}"])
        if k.endswith("wk.bias"):
            assert float(p.grad.abs().max()) < 1e-3  # analytically zero
            continue
        got_n = float(p.grad.double().norm())
        assert abs(got_n - ref_n) <= 0.03 * ref_n + 1e-7, (k, got_n, ref_n)
        idx = sample_idx(p.numel(), 16, 7919 + i)
        got = p.grad.reshape(-1)[idx.to(dev())].cpu().double()
        ref = _t(g[f"gsamp/{k}"]).double()
        assert float((got - ref).norm()) <= GRAD_TOL * float(ref.norm()) + 0.03 * ref_n / max(p.numel(), 1) ** 0.5 * 4, k


@pytest.mark.parametrize("name,batch", [("tiny", 4), ("small", 2), ("base", 2)])
def test_model_cross_vs_bf16_emulating_oracle(name, batch):
    """Gate (1): same arithmetic class on both sides."""
    cfg, sd, img, labels, model, caps, logits, loss = _run_model(name, batch)
    cap = {}
    with R.emulate_bf16():
        ref_logits, ref_loss = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            assert rel(caps[b][m], cap[f"msb{b}"][m]) < 3e-3, (b, m, rel(caps[b][m], cap[f"msb{b}"][m]))
            # the CLS row is the sum of bf16-operand attention / FFN updates with no large residual to dilute their rounding
            # (it starts at |cls_token + pos| ~ 0.03): measured 3.8e-3 .. 4.6e-3 at configs[1]
            assert rel(caps[b][m][:, 0], cap[f"msb{b}"][m][:, 0]) < 5.5e-3
    # everything behind the fused CLS token runs fp32 operands on both sides (xvit_linear_f32 / oracle `exact`): what is left is
    # the CLS rows' own deviation amplified by the cancellation in the 2-class head.  Four numbers (batch 2 x 2 classes) make a
    # noisy estimate of that amplification: measured at configs[1] 6.4e-3 (dense fusion form) .. 1.1e-2 (low-rank form) for the same
    # 4.5e-3 on the CLS rows and 1.45e-2 vs 1.46e-2 against the fp32 reference (tools/parity_forms_probe.py -> profiles/r03_parity_probe.txt);
    # at batch 6 the two forms read 9.9e-3.  The gate that means something is the CLS-row one above; this one guards against a gross error.
    assert rel(logits, ref_logits) < 1.5e-2, rel(logits, ref_logits)
    assert abs(float(loss) - float(ref_loss)) < 2e-3


def test_model_cross_base_vs_fp32_oracle_on_bf16_rounded_weights_and_inputs():
    """BASELINE.md section 5's end-to-end comparator, computed as written: the HIP path against the fp32 oracle run on the SAME bf16-rounded
    weights and inputs (so weight / input quantisation is on both sides and what is measured is the bf16 arithmetic of the path alone),
    at configs[1].  Budget = PyTorch's own bf16 autocast on the reference: <= 5.8e-3 on block outputs, 6.8e-3 on logits.
    Measured (profiles/r03_parity_probe.txt): all tokens of every MultiScaleBlock output 1.2e-3 .. 1.6e-3 (4x inside the budget), the
    CLS rows alone 3.4e-3 .. 3.9e-3 (inside), logits 9.2e-3: OUTSIDE the 6.8e-3 of autocast.  Reason: the CLS row is the sum of bf16-operand
    attention / FFN updates with no large residual to dilute their rounding (it starts at |cls + pos| ~ 0.03), and the 2-class head turns its
    3.5e-3 into ~9e-3 through cancellation (|logit| ~ 0.02 .. 0.3 from O(1) terms); the bf16-emulating oracle of the same arithmetic class
    sits at 7.0e-3 itself.  Gates: the section-5 budget for the blocks, measured x 1.5 for CLS rows and logits."""
    import xvit
    cfg = R.make_config("base")
    sd = {k: R.bf16_round(v) for k, v in R.make_state_dict(cfg, seed=0).items()}
    img, labels = R.make_inputs(cfg, 2, seed=0)
    img = R.bf16_round(img)
    cap = {}
    ref_logits, ref_loss = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(sd)
    model.train()
    caps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, b=b: caps.__setitem__(b, [t.detach() for t in o])) for b, blk in enumerate(model.transformer)]
    logits, loss = model(img.to(dev()), labels.to(dev()))
    for h in hooks:
        h.remove()
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            e_all, e_cls = rel(caps[b][m], cap[f"msb{b}"][m]), rel(caps[b][m][:, 0], cap[f"msb{b}"][m][:, 0])
            assert e_all < 2.4e-3, (b, m, e_all)           # section-5 budget 5.8e-3; measured 1.6e-3
            assert e_cls < 5.8e-3, (b, m, e_cls)           # measured 3.9e-3
    assert rel(logits, ref_logits) < 1.4e-2, rel(logits, ref_logits)     # measured 9.2e-3 (autocast: 6.8e-3, see the docstring)
    assert abs(float(loss.detach()) - float(ref_loss)) < 1e-3


@pytest.mark.parametrize("name,over", [("tiny", {}), ("small", {}),
                                       ("tiny", dict(num_modalities=4, attn_order={"0": "1", "1": "2", "2": "3", "3": "0"})),
                                       ("tiny", dict(num_modalities=3, attn_order={"0": "2"}))])   # modalities 1 and 2 have no fusion of their own
def test_unobserved_last_block_cls_only_path_matches_full_path(name, over):
    """ModelCross runs its last MultiScaleBlock CLS-only when nothing observes the block's output (the heads read
    x[m][:, 0] only, model_cross.py:203); with a forward hook it produces the reference's full token tensors.  Both
    paths must give the same logits, loss and parameter gradients."""
    import xvit
    cfg = R.make_config(name, **over)
    img, labels = R.make_inputs(cfg, 3, seed=5)
    img, labels = img.to(dev()), labels.to(dev())
    torch.manual_seed(2)
    model = xvit.ModelCross(cfg).to(dev())
    model.train()

    def run(observed):
        shapes = []
        hooks = [model.transformer[-1].register_forward_hook(lambda m, i, o: shapes.append([tuple(t.shape) for t in o]))] if observed else []
        model.zero_grad()
        logits, loss = model(img, labels)
        loss.backward()
        for h in hooks:
            h.remove()
        return logits.detach().clone(), float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}, shapes

    l_full, loss_full, g_full, shapes = run(True)
    assert all(s[1] == cfg.img_size[0] // cfg.patch_size[0] * (cfg.img_size[1] // cfg.patch_size[1]) * (cfg.img_size[2] // cfg.patch_size[2]) + 1 for s in shapes[0])
    l_cls, loss_cls, g_cls, _ = run(False)
    assert torch.equal(l_full, l_cls) and loss_full == loss_cls      # the CLS rows go through the very same kernels
    for k in g_full:
        assert rel(g_cls[k], g_full[k]) < 1e-5 or float(g_full[k].abs().max()) < 1e-6, k


def test_partial_fusion_map_vs_bf16_emulating_oracle():
    """attn_order need not cover every modality (model_cross.py:131-147: a modality without an entry passes through the
    MultiScaleBlock unchanged).  3 modalities, only 0 <- 2 fused: forward vs the bf16-emulating oracle, gradients vs
    the fp32 oracle's autograd (norms), with the unobserved (CLS-only last block) path."""
    import xvit
    cfg = R.make_config("tiny", num_modalities=3, attn_order={"0": "2"})
    sd = R.make_state_dict(cfg, seed=3)
    img, labels = R.make_inputs(cfg, 4, seed=8)
    model = xvit.ModelCross(cfg).to(dev())
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd, strict=True)
    model.train()
    logits, loss = model(img.to(dev()), labels.to(dev()))
    loss.backward()
    with R.emulate_bf16():
        ref_logits, ref_loss = R.model_cross_forward(sd, img, labels, cfg)
    # measured 1.9e-3 (profiles/r03_measured_gates.txt)
    assert note("partial_fusion_map.logits_vs_emu", rel(logits, ref_logits)) < 3e-3 and abs(float(loss.detach()) - float(ref_loss)) < 2e-3
    _, _, grads = R.model_cross_loss_and_grads(sd, img, labels, cfg)
    for k, p in model.named_parameters():
        if k.endswith("wk.bias"):
            assert float(p.grad.abs().max()) < 1e-3          # analytically zero (softmax shift invariance)
            continue
        ref_n = float(grads[k].double().norm())
        got_n = float(p.grad.double().norm())
        assert abs(got_n - ref_n) <= 0.05 * ref_n + 1e-6, (k, got_n, ref_n)


def test_accumulates_like_autograd_and_fails_loudly_off_gpu():
    import xvit
    cfg = R.make_config("tiny")
    model = xvit.ModelCross(cfg).to(dev())
    img, labels = R.make_inputs(cfg, 2, seed=1)
    _, loss = model(img.to(dev()), labels.to(dev()))
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    _, loss = model(img.to(dev()), labels.to(dev()))
    loss.backward()  # second backward accumulates into .grad
    for k, p in model.named_parameters():
        assert rel(p.grad, 2 * g1[k]) < 1e-3 or float(g1[k].abs().max()) < 1e-6, k
    cpu_model = xvit.ModelCross(cfg)
    with pytest.raises(RuntimeError):
        cpu_model(img, labels)  # no CPU fallback
    model.eval()
    model(img.to(dev()), labels.to(dev()))


@pytest.mark.parametrize("name,batch", [("long", 1), ("ucsf", 1), ("mist", 2)])
def test_large_configs_vs_bf16_emulating_oracle(name, batch):
    """BASELINE.json configs[4] (128^3, 8^3 patches -> N = 4097), configs[2] (4 modalities, 240^3 ->
    N = 3376, 4-ring) and the reference's own run shape (config2.py + main_mist.py:71: d = 1024, 16 heads, mlp 4096, three modalities of
    128 x 128 x 64 in a ring, 16 x 16 x 8 patches -> N = 513; 242 M parameters): forward parity against the emulating oracle at batch 1, EVERY parameter gradient's norm against the
    oracle's autograd (1 %), and a size-independent property of the path — each CLS-fused stream keeps its own patch tokens untouched by
    the fusion step (model_cross.py:142), checked through the backward: d loss / d img is non-zero for
    every modality."""
    import xvit
    cfg = R.make_config(name)
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, batch, seed=0)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(sd)
    model.train()
    caps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, b=b: caps.__setitem__(b, [t.detach() for t in o])) for b, blk in enumerate(model.transformer)]
    logits, loss = model(img.to(dev()), labels.to(dev()))
    loss.backward()
    for h in hooks:
        h.remove()
    # one oracle run, forward AND backward, in the emulation of the path's arithmetic class (autograd through the bf16 roundings:
    # gradients are rounded where the kernels store bf16 gradients); every parameter gradient's NORM is then compared, not its finiteness
    cap = {}
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with R.emulate_bf16():
        ref_logits, ref_loss = R.model_cross_forward(leaf, img, labels, cfg, capture=cap)
        ref_loss.backward()
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            assert rel(caps[b][m], cap[f"msb{b}"][m].detach()) < 3e-3, (b, m)
    assert abs(float(loss.detach()) - float(ref_loss)) < 5e-3
    worst = 0.0
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if k.endswith("wk.bias"):
            assert float(p.grad.abs().max()) < 1e-3          # analytically zero (softmax shift invariance)
            continue
        ref_n, got_n = float(leaf[k].grad.double().norm()), float(p.grad.double().norm())
        worst = max(worst, abs(got_n - ref_n) / (ref_n + 1e-12))
        assert abs(got_n - ref_n) <= 0.01 * ref_n + 1e-7, (k, got_n, ref_n)      # measured worst: 1.2e-3 (long), 2.8e-3 (ucsf); mist passes the same 1 % (3.5e-3 against the fp32 fixture)
    note(f"large_config.{name}.worst_grad_norm_dev", worst)


@pytest.mark.parametrize("N", [3376, 4097])
def test_self_attention_block_long_sequence_forward_and_gradients(N):
    """One SelfAttentionBlock at the sequence lengths of configs[2] (N = 3376) and configs[4] (N = 4097), d = 768, 12 heads:
    output, input gradient and EVERY parameter gradient against the oracle's autograd (not just finiteness)."""
    import xvit
    cfg = R.make_config("base")
    sd = R.make_state_dict(cfg, seed=0)
    pfx = "transformer.0.blocks.0.0"
    blk = xvit.SelfAttentionBlock(cfg).to(dev())
    blk.load_state_dict({k[len(pfx) + 1:]: v for k, v in sd.items() if k.startswith(pfx + ".")})
    blk.train()
    x = randn(1, N, cfg.hidden_dim, seed=N)
    w = randn(1, N, cfg.hidden_dim, seed=N + 1)                    # d loss / d y
    xr = x.to(dev()).requires_grad_()
    y = blk(xr)
    (y * w.to(dev())).sum().backward()
    leaf = {k: v.clone().requires_grad_() for k, v in sd.items() if k.startswith(pfx + ".")}
    xo = x.clone().requires_grad_()
    yo = R.self_block(leaf, pfx, xo, cfg.num_heads)
    (yo * w).sum().backward()
    assert rel(y, yo) < BLOCK_TOL, rel(y, yo)
    assert rel(xr.grad, xo.grad) < GRAD_TOL, rel(xr.grad, xo.grad)
    for k, p in blk.named_parameters():
        ref = leaf[pfx + "." + k].grad
        assert rel(p.grad, ref) < GRAD_TOL, (k, rel(p.grad, ref))


def test_model_vit_vs_reference_golden_and_emulation(golden_dir):
    """modelv3.ModelVIT drop-in: reference state_dict keys, forward vs the reference's outputs and vs the
    bf16-emulating oracle, every parameter gradient's norm vs the reference's."""
    import xvit
    g = np.load(os.path.join(golden_dir, "model_vit_small.npz"))
    cfg = R.make_config("small", num_layers=2)
    sd = R.make_vit_state_dict(cfg, seed=11)
    img, labels = R.make_inputs(cfg, 3, seed=4)
    model = xvit.ModelVIT(cfg).to(dev())
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd, strict=True)
    model.train()
    logits, loss = model(img.to(dev()), labels.to(dev()))
    loss.backward()
    # measured 4.1e-3 vs the reference's fp32 output, 4.0e-3 vs the emulation; the head runs fp32 operands (xvit_linear_f32) like ModelCross's
    assert note("model_vit.logits_vs_fp32_golden", rel(logits, _t(g["logits"]))) < 6.2e-3 and abs(float(loss.detach()) - float(g["loss"])) < 5e-3
    with R.emulate_bf16():
        emu_logits, emu_loss = R.model_vit_forward(sd, img, labels, cfg)
    assert note("model_vit.logits_vs_emu", rel(logits, emu_logits)) < 6e-3 and abs(float(loss.detach()) - float(emu_loss)) < 2e-3
    for k, p in model.named_parameters():
        ref_n = float(g[f"gnorm/{k}"])
        tol = 0.03
        if k == "mlp_head.4.bias":
            # the bias of the last Linear: dL/db = mean_b(softmax(logits_b) - target_b) is a closed form of OUR logits, so the
            # backward itself is checked exactly; against the reference's value this 2-element tensor inherits the logits'
            # own bf16 deviation (4e-3 measured, gate above) times |dp/dlogit| / |p - y| at batch 3, i.e. a few per cent
            eps = float(getattr(cfg, "label_smoothing", 0.0))
            target = torch.nn.functional.one_hot(labels.to(dev()), logits.shape[1]).float() * (1.0 - eps) + eps / logits.shape[1]
            want = (torch.softmax(logits.detach().float(), dim=1) - target).mean(0)
            assert rel(p.grad, want) < 1e-4, rel(p.grad, want)
            tol = 0.08
        assert abs(float(p.grad.double().norm()) - ref_n) <= tol * ref_n + 1e-7, k


def test_checkpoint_round_trip_in_lightning_layout(tmp_path):
    """main_mist.py:174-180 restores `torch.load(ckpt)["state_dict"]` into the model.  The flat fp32 / bf16 weight
    buffers (functional.FlatWeights) must be invisible to that: same keys and shapes as the reference's state dict, a
    saved file loads with weights_only=True into a fresh model AND into a model whose flat buffers are already live,
    and both then compute exactly what the source model computes."""
    import xvit
    cfg = R.make_config("tiny")
    img, labels = R.make_inputs(cfg, 3, seed=11)
    img, labels = img.to(dev()), labels.to(dev())
    torch.manual_seed(1)
    src = xvit.ModelCross(cfg).to(dev())
    opt = torch.optim.SGD(src.parameters(), lr=0.05)
    for _ in range(2):                                   # train a little so the weights differ from any initialisation
        opt.zero_grad()
        src(img, labels)[1].backward()
        opt.step()
    src.eval()
    ref_logits, ref_loss = src(img, labels)
    sd = src.state_dict()
    assert set(sd) == set(R.make_state_dict(cfg, seed=0))                       # the reference's key set (oracle state dict)
    assert all(v.is_contiguous() for v in sd.values())
    path = tmp_path / "epoch=0.ckpt"
    torch.save({"state_dict": {k: v.cpu() for k, v in sd.items()}, "epoch": 0}, path)
    loaded = torch.load(path, weights_only=True)["state_dict"]

    fresh = xvit.ModelCross(cfg).to(dev())
    fresh.load_state_dict(loaded, strict=True)
    fresh.eval()
    live = xvit.ModelCross(cfg).to(dev())
    live.eval()
    live(img, labels)                                    # flat buffers + bf16 operand copies now exist for the OLD weights
    live.load_state_dict(loaded, strict=True)            # in-place copy into the flat views: the copies must be re-cast
    for m in (fresh, live):
        logits, loss = m(img, labels)
        assert torch.equal(logits, ref_logits) and float(loss.detach()) == float(ref_loss.detach())


def test_inplace_cls_splice_equals_copying_form(monkeypatch):
    """CrossFusionFn writes the fused CLS rows into x_i in place (no copy of the token tensor, reference model_cross.py:142
    torch.cat): logits and EVERY gradient must equal the copying form bit for bit (XVIT_CLS_INPLACE=0), also when the
    backward runs twice on a retained graph."""
    import xvit
    cfg = R.make_config("small")
    sd = R.make_state_dict(cfg, seed=3)
    img, labels = R.make_inputs(cfg, 3, seed=3)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("XVIT_CLS_INPLACE", mode)
        xvit.ops.set_deterministic(True)
        try:
            model = xvit.ModelCross(cfg).to(dev())
            model.load_state_dict(sd)
            model.train()
            logits, loss = model(img.to(dev()), labels.to(dev()))
            loss.backward(retain_graph=True)
            g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
            model.zero_grad()
            loss.backward()
            g2 = {k: p.grad.clone() for k, p in model.named_parameters()}
        finally:
            xvit.ops.set_deterministic(False)
        assert all(torch.equal(g1[k], g2[k]) for k in g1), "second backward over the retained graph differs"
        res[mode] = (logits.detach().clone(), g1)
    assert torch.equal(res["1"][0], res["0"][0])
    for k in res["1"][1]:
        assert torch.equal(res["1"][1][k], res["0"][1][k]), k


def test_multiscale_block_standalone_keeps_callers_tensors_intact():
    """The in-place CLS splice is reserved for ModelCross (which knows every tensor has one owner).  Driven directly,
    MultiScaleBlock behaves like the reference's torch.cat (model_cross.py:140-142): a forward hook on a branch keeps the
    branch's own CLS rows, the caller's grad_outputs is not written to, and a gradient shared by two nodes stays whole."""
    import xvit
    cfg = R.make_config("small")
    torch.manual_seed(0)
    blk = xvit.MultiScaleBlock(cfg).to(dev())
    blk.train()
    d = R.derived(cfg)
    B, N, dm = 2, d.N, cfg.hidden_dim
    xs = [torch.randn(B, N, dm, device=dev(), requires_grad=True) for _ in range(cfg.num_modalities)]
    seen = {}
    h = blk.blocks[0].register_forward_hook(lambda m, i, o: seen.__setitem__("branch0", o))   # kept WITHOUT a clone
    outs = blk(xs)
    h.remove()
    ref = blk.blocks[0](xs[0]).detach()
    assert torch.equal(seen["branch0"].detach(), ref), "the hooked branch output was modified after the hook saw it"
    assert not torch.equal(outs[0][:, 0].detach(), ref[:, 0]), "the fusion did not replace the CLS rows of its output"
    assert torch.equal(outs[0][:, 1:].detach(), ref[:, 1:])
    g = [torch.randn_like(o) for o in outs]
    keep = [t.clone() for t in g]
    grads = torch.autograd.grad(outs, xs, grad_outputs=g, retain_graph=True)
    assert all(torch.equal(a, b) for a, b in zip(g, keep)), "backward wrote into the caller's grad_outputs"
    # the same gradient object handed to two consumers: out0 + aux, with aux a second leaf
    aux = torch.zeros_like(outs[0], requires_grad=True)
    (g_x0, g_aux) = torch.autograd.grad((outs[0] + aux), (xs[0], aux), grad_outputs=g[0])
    assert torch.equal(g_aux, keep[0]), "the gradient shared with the other addend was modified"
    assert all(torch.isfinite(t).all() for t in grads) and torch.isfinite(g_x0).all()


def test_model_cross_hook_on_a_branch_sees_the_branch_output():
    """A forward hook anywhere inside a MultiScaleBlock's branch switches that fusion to the copying form: the hooked tensor
    keeps the branch's CLS rows while the model's logits stay what they are without the hook."""
    import xvit
    cfg = R.make_config("small")
    sd = R.make_state_dict(cfg, seed=1)
    img, labels = R.make_inputs(cfg, 2, seed=1)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        base, _ = model(img.to(dev()), labels.to(dev()))
    seen = {}
    h = model.transformer[0].blocks[1][-1].register_forward_hook(lambda m, i, o: seen.__setitem__("o", o))
    with torch.no_grad():
        hooked, _ = model(img.to(dev()), labels.to(dev()))
    h.remove()
    assert torch.equal(base, hooked)
    # recompute the branch alone: its CLS rows are what the hook must have kept
    with torch.no_grad():
        toks = xvit.functional.PatchEmbedFn.apply(img.to(dev()), model.patch_to_embedding.weight, model.patch_to_embedding.bias, model.cls_token,
                                                  model.pos_embedding, model.patch_size, 0.0)
        ref = model.transformer[0].blocks[1](toks[1])
    assert torch.equal(seen["o"], ref)


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_fusion_kv_path_low_rank_vs_dense(monkeypatch, p):
    """The fusion's K/V path in its two forms: "lowrank" (default: wk / wv never meet the N tokens, csrc/head_linear.hip) and the reference's
    literal order (XVIT_XATTN_FORM=dense: kv and dkv tensors, GEMMs, column sums; also what runs when H > 16), without dropout and with the
    reference's rate 0.25 on every site (model_cross.py:97 puts one on the probabilities: both forms draw the SAME masks from the same
    seeds, so they compute the same function).  The low-rank forward differs by where bf16 rounding happens (K and V are no longer rounded
    at all); gradients agree to the bf16 rounding of the tensors only one form has."""
    import xvit
    import xvit.functional as XF
    cfg = R.make_config("small", dropout=p)
    sd = R.make_state_dict(cfg, seed=6)
    img, labels = R.make_inputs(cfg, 3, seed=6)
    res = {}
    torch.manual_seed(1234)                                    # the dropout seeds derive from torch's seed and a call counter: fixed, so the
    for form in ("lowrank", "dense"):                          # masks (and the measured distances) do not depend on which tests ran before
        monkeypatch.setattr(XF, "XATTN_FORM", form)
        monkeypatch.setattr(XF, "_DROP_CALLS", 1000)           # the same seeds for every dropout site in both runs
        model = xvit.ModelCross(cfg).to(dev())
        model.load_state_dict(sd)
        model.train()
        logits, loss = model(img.to(dev()), labels.to(dev()))
        loss.backward()
        res[form] = (logits.detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters()})
    full, dense = res["lowrank"], res["dense"]
    assert note(f"fusion_forms.p{p}.logits", rel(full[0], dense[0])) < 8e-3, rel(full[0], dense[0])
    worst = (0.0, "")
    for k, g in full[1].items():
        if k.endswith("wk.bias"):
            continue                      # analytically zero (softmax gradients sum to zero over the keys): rounding noise or exact 0
        if float(dense[1][k].abs().max()) >= 1e-6:
            worst = max(worst, (rel(g, dense[1][k]), k))
    # the most sensitive gradient is wq's (through the softmax Jacobian, whose ds the two forms round at different places); with masks the
    # sums run over fewer, larger terms: measured 1.3e-2 (p = 0) and 2.0e-2 (p = 0.25)
    note(f"fusion_forms.p{p}.worst_grad", worst[0])
    assert worst[0] < (3e-2 if p > 0 else 2e-2), worst


def test_fusion_form_is_chosen_by_where_the_bound_is(monkeypatch):
    """XVIT_XATTN_FORM=auto (the default): the literal order on small eager batches (host-bound: fewer launches), the low-rank form from
    8192 token rows on and inside a HIP-graph capture; shapes the low-rank kernels are not built for always take the literal order."""
    import xvit.functional as XF
    monkeypatch.setattr(XF, "XATTN_FORM", "auto")
    assert not XF._xattn_lowrank_ok(12, 768, 8 * 513) and XF._xattn_lowrank_ok(12, 768, 16 * 513) and XF._xattn_lowrank_ok(16, 1024, 126 * 513)
    assert not XF._xattn_lowrank_ok(24, 1536, 126 * 513) and not XF._xattn_lowrank_ok(3, 96, 126 * 513)
    g = torch.cuda.CUDAGraph()
    seen = []
    with torch.cuda.graph(g):
        seen.append(XF._xattn_lowrank_ok(12, 768, 8 * 513))
    assert seen == [True]
    monkeypatch.setattr(XF, "XATTN_FORM", "dense")
    assert not XF._xattn_lowrank_ok(12, 768, 126 * 513)
    monkeypatch.setattr(XF, "XATTN_FORM", "lowrank")
    assert XF._xattn_lowrank_ok(12, 768, 513)


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_gradient_handoff_in_bf16_changes_no_bit_and_saves_the_casts(monkeypatch, name):
    """A block's backward starts by casting its incoming gradient to bf16; the node that produced the gradient (the next block's LayerNorm
    backward, a cls-only fusion's, the fan-out sum) can write that copy in the pass it makes anyway (functional.attach_b16).  Same values
    rounded once either way: logits and every gradient are bit-identical with the hand-off on and off (fixed-order reductions), and the
    number of stand-alone cast launches drops."""
    import xvit
    import xvit.functional as XF
    from xvit import ops
    cfg = R.make_config(name)
    sd = R.make_state_dict(cfg, seed=2)
    img, labels = R.make_inputs(cfg, 8, seed=2)      # d-column counts and rows that keep numel % 8 == 0
    res = {}
    ops.set_deterministic(True)
    try:
        for on in (True, False):
            monkeypatch.setattr(XF, "B16_HANDOFF", on)
            model = xvit.ModelCross(cfg).to(dev())
            model.load_state_dict(sd)
            model.train()
            monkeypatch.setattr(ops, "PROFILE", [])
            logits, loss = model(img.to(dev()), labels.to(dev()))
            loss.backward()
            torch.cuda.synchronize()
            fam = [e[0] for e in ops.PROFILE]
            monkeypatch.setattr(ops, "PROFILE", None)
            res[on] = (logits.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}, fam.count("cast_f32_bf16"), fam.count("add_cast"))
    finally:
        ops.set_deterministic(False)
    assert torch.equal(res[True][0], res[False][0])
    for k, g in res[True][1].items():
        assert torch.equal(g, res[False][1][k]), k
    assert res[True][2] < res[False][2] and res[False][3] == 0, (res[True][2:], res[False][2:])
