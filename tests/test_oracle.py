"""CPU: the oracle restatement (oracle/ref_cpu.py) against the golden vectors the
reference itself produced (oracle/make_golden.py).  No GPU, no /root/reference."""
import os

import numpy as np
import pytest
import torch

import ref_cpu as R

TOL = 2e-5  # fp32 round-off between two CPU evaluation orders


def rel(a, b):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _sd_hash(sd):
    return R.tensor_sha256(torch.cat([v.reshape(-1) for _, v in sorted(sd.items())]))


PARTIAL = dict(num_modalities=3, attn_order={"0": "1", "1": "2"})   # main_mist.py:72: modality 2 has no fusion of its own


@pytest.mark.parametrize("name,batch,fixture,over", [("tiny", 4, "tiny", {}), ("small", 2, "small", {}), ("tiny", 3, "partial", PARTIAL)])
def test_model_cross_matches_reference(golden_dir, name, batch, fixture, over):
    g = np.load(os.path.join(golden_dir, f"model_cross_{fixture}.npz"))
    cfg = R.make_config(name, **over)
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, batch, seed=0)
    # the generators must reproduce the tensors the reference was run on
    assert str(g["img_sha256"]) == R.tensor_sha256(img)
    assert str(g["sd_sha256"]) == _sd_hash(sd)
    assert np.array_equal(g["labels"], labels.numpy())
    cap = {}
    logits, loss = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
    assert rel(logits, g["logits"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            assert rel(cap[f"msb{b}"][m], g[f"msb{b}/mod{m}/full"]) < TOL
    _, _, grads = R.model_cross_loss_and_grads(sd, img, labels, cfg)
    for k, gr in grads.items():
        if k.endswith("wk.bias"):  # identically-zero gradient: round-off only
            assert float(gr.abs().max()) < 1e-5
            continue
        assert abs(float(gr.double().norm()) - float(g[f"gnorm/{k}"])) <= 5e-4 * float(g[f"gnorm/{k}"]) + 1e-9, k


def test_blocks_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    cfg = R.make_config("small")
    sd = R.make_state_dict(cfg, seed=3)
    assert str(g["sd_sha256"]) == _sd_hash(sd)
    H = cfg.num_heads
    for N in (17, 65, 130):
        x = torch.from_numpy(g[f"sab/N{N}/x"])
        assert rel(R.self_block(sd, "transformer.0.blocks.0.0", x, H), g[f"sab/N{N}/y"]) < TOL
        assert rel(R.self_attention(sd, "transformer.0.blocks.0.0.attn.fn", x, H), g[f"attn/N{N}/y"]) < TOL
        assert rel(R.feed_forward(sd, "transformer.0.blocks.0.0.ffn.fn", x), g[f"ffn/N{N}/y"]) < TOL
        assert rel(R.cross_block(sd, "transformer.0.fusion.0", x, H), g[f"cab/N{N}/y"]) < TOL
        assert rel(R.cls_cross_attention(sd, "transformer.0.fusion.0.attn.fn", x, H), g[f"xattn/N{N}/y"]) < TOL
        xr = x.clone().requires_grad_(True)
        R.self_block(sd, "transformer.0.blocks.0.0", xr, H).square().sum().backward()
        assert rel(xr.grad, g[f"sab/N{N}/dx"]) < 1e-4
        xr = x.clone().requires_grad_(True)
        R.cross_block(sd, "transformer.0.fusion.0", xr, H).square().sum().backward()
        assert rel(xr.grad, g[f"cab/N{N}/dx"]) < 1e-4
    xs = [torch.from_numpy(g[f"msb/x{m}"]) for m in range(3)]
    ys = R.multi_scale_block(sd, "transformer.0", xs, cfg)
    for m in range(3):
        assert rel(ys[m], g[f"msb/y{m}"]) < TOL


def test_patchify_index_map(golden_dir):
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    vol = torch.arange(2 * 8 * 12 * 6, dtype=torch.float32).reshape(2, 8, 12, 6)
    assert np.array_equal(R.patchify(vol, (4, 3, 2)).numpy(), g["patchify/out"])
    # closed form: token t=(h*Wn+w)*Dn+d, feature f=(p1*hp+p2)*wp+p3
    Dn, Hn, Wn, dp, hp, wp = 2, 4, 3, 4, 3, 2
    out = R.patchify(vol, (dp, hp, wp))
    for (b, h, w, d, p1, p2, p3) in [(0, 0, 0, 0, 0, 0, 0), (1, 3, 2, 1, 3, 2, 1), (0, 2, 1, 1, 2, 0, 1)]:
        assert out[b, (h * Wn + w) * Dn + d, (p1 * hp + p2) * wp + p3] == vol[b, d * dp + p1, h * hp + p2, w * wp + p3]


def test_encoder_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder.npz"))
    sd = R.make_encoder_state_dict(256, 512, 2, seed=5)
    assert str(g["sd_sha256"]) == _sd_hash(sd)
    for N in (65, 130):
        x = torch.from_numpy(g[f"N{N}/x"])
        assert rel(R.encoder_forward(sd, x, 4, 2), g[f"N{N}/y"]) < TOL
        xr = x.clone().requires_grad_(True)
        R.encoder_forward(sd, xr, 4, 2).square().sum().backward()
        assert rel(xr.grad, g[f"N{N}/dx"]) < 1e-4


def test_flop_model_matches_baseline():
    # BASELINE.md §3 table
    fwd, both = R.flops_per_sample(R.make_config("base"))
    assert abs(fwd / 1e9 - 75.910) < 0.01 and abs(both / 1e9 - 221.286) < 0.01
    fwd, both = R.flops_per_sample(R.make_config("tiny"))
    assert abs(fwd / 1e9 - 0.137) < 0.001  # 2x2 blocks, 5.66 M params
    fwd, both = R.flops_per_sample(R.make_config("long"))
    assert abs(fwd / 1e9 - 921.7) < 0.1


def test_base_logits_golden_present(golden_dir):
    g = np.load(os.path.join(golden_dir, "model_cross_base.npz"))
    assert g["logits"].shape == (2, 2) and np.isfinite(g["logits"]).all()


def test_reference_run_shape_matches_reference(golden_dir):
    """The reference's own run shape (config2.py:5-22 + main_mist.py:71: d = 1024, 16 heads, mlp 4096, three 128 x 128 x 64 modalities in a
    ring, 16 x 16 x 8 patches; 241.95 M parameters, SURVEY.md section 8 shape ladder): the restatement's forward against the fixture
    oracle/make_golden.py wrote from the imported reference."""
    cfg = R.make_config("mist")
    fwd, both = R.flops_per_sample(cfg)
    assert abs(fwd / 1e9 - 187.4) < 0.1 and abs(both / 1e9 - 555.7) < 0.1
    g = np.load(os.path.join(golden_dir, "model_cross_mist.npz"))
    sd = R.make_state_dict(cfg, seed=0)
    assert sum(v.numel() for v in sd.values()) == 241_945_606
    img, labels = R.make_inputs(cfg, int(g["batch"]), seed=0)
    assert str(g["img_sha256"]) == R.tensor_sha256(img)
    cap = {}
    with torch.no_grad():
        logits, loss = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
    assert rel(logits, torch.from_numpy(g["logits"])) < 2e-5 and abs(float(loss) - float(g["loss"])) < 2e-6
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            assert rel(cap[f"msb{b}"][m][:, 0], torch.from_numpy(g[f"msb{b}/mod{m}/cls"])) < 2e-5


def test_model_vit_matches_reference(golden_dir):
    """modelv3.ModelVIT (the reference's concatenated-token comparison arm)."""
    g = np.load(os.path.join(golden_dir, "model_vit_small.npz"))
    cfg = R.make_config("small", num_layers=2)
    sd = R.make_vit_state_dict(cfg, seed=11)
    img, labels = R.make_inputs(cfg, 3, seed=4)
    assert str(g["img_sha256"]) == R.tensor_sha256(img) and str(g["sd_sha256"]) == _sd_hash(sd)
    logits, loss = R.model_vit_forward(sd, img, labels, cfg)
    assert rel(logits, g["logits"]) < TOL and abs(float(loss) - float(g["loss"])) < 1e-5


def test_resize_with_pad_or_crop_rule():
    """Input-stage restatement (parity unpinned: MONAI absent).  Hand-checked cases of the documented rule."""
    v = torch.arange(5 * 4 * 7, dtype=torch.int16).reshape(5, 4, 7)
    out = R.resize_with_pad_or_crop(v, (3, 6, 7), -1)          # crop D 5->3 (start 5//2 - 3//2 = 1), pad H 4->6 (1 before, 1 after)
    assert out.shape == (3, 6, 7)
    assert torch.equal(out[:, 1:5, :], v[1:4].float()) and bool((out[:, 0] == -1).all()) and bool((out[:, 5] == -1).all())
    out = R.resize_with_pad_or_crop(v, (8, 4, 4), -1)          # pad D 5->8 (1 before, 2 after), crop W 7->4 (start 3 - 2 = 1)
    assert torch.equal(out[1:6], v[:, :, 1:5].float()) and bool((out[0] == -1).all()) and bool((out[6:] == -1).all())
    # the UCSF-PDGM shape of the reference's sample data: 240x240x155 -> 128^3 is a pure centre crop
    big = torch.zeros(240, 240, 155, dtype=torch.int16); big[56, 56, 13] = 7; big[183, 183, 140] = 9
    out = R.resize_with_pad_or_crop(big, (128, 128, 128), -1)
    assert out[0, 0, 0] == 7 and out[127, 127, 127] == 9


def test_step_metrics_restatement_vs_scikit_learn():
    """oracle.binary_step_metrics (log_stats, model_cross.py:243-255) against scikit-learn's definitions of the same
    quantities (torchmetrics is not installed here): random batches, ties, single-class batches."""
    import numpy as np
    from sklearn import metrics as skm
    g = torch.Generator().manual_seed(3)
    cases = []
    for B in (1, 2, 7, 64, 126):
        logits = torch.randn(B, 2, generator=g)
        labels = torch.randint(0, 2, (B,), generator=g)
        cases.append((logits, labels))
    tied = torch.tensor([[0.5, 0.5], [1.0, -1.0], [1.0, -1.0], [-2.0, 2.0], [-2.0, 2.0], [0.0, 0.0]])
    cases.append((tied, torch.tensor([1, 0, 1, 1, 0, 0])))
    cases.append((torch.randn(9, 2, generator=g), torch.ones(9, dtype=torch.int64)))      # no negatives
    cases.append((torch.randn(9, 2, generator=g), torch.zeros(9, dtype=torch.int64)))     # no positives
    for logits, labels in cases:
        m = R.binary_step_metrics(logits, labels)
        y = labels.numpy()
        pred = torch.argmax(logits, dim=1).numpy()
        assert m["counts"] == tuple(int(v) for v in skm.confusion_matrix(y, pred, labels=[0, 1]).ravel())
        assert abs(m["acc"] - skm.accuracy_score(y, pred)) < 1e-12
        assert abs(m["prec"] - skm.precision_score(y, pred, zero_division=0)) < 1e-12
        assert abs(m["rec"] - skm.recall_score(y, pred, zero_division=0)) < 1e-12
        assert abs(m["spec"] - skm.recall_score(1 - y, 1 - pred, zero_division=0)) < 1e-12
        assert abs(m["f1"] - skm.f1_score(y, pred, zero_division=0)) < 1e-12
        assert abs(m["npv"] - skm.precision_score(1 - y, 1 - pred, zero_division=0)) < 1e-12
        if 0 < y.sum() < len(y):
            prob = torch.softmax(logits.float(), dim=1)[:, 1].numpy()
            assert abs(m["auc_roc"] - skm.roc_auc_score(y, prob)) < 1e-12
        else:
            assert m["auc_roc"] == 0.0
    ep = R.epoch_metrics(cases)
    n = sum(len(l) for _, l in cases)
    assert abs(ep["acc"] - sum(R.binary_step_metrics(a, b)["acc"] * len(b) for a, b in cases) / n) < 1e-12


def test_low_rank_emulation_is_the_same_function_as_the_literal_cross_attention():
    """oracle/ref_cpu.py restates the fusion's attention twice: the reference's literal order (pinned against the reference) and,
    for emulate_bf16(), the order of the HIP path's low-rank form.  With rounding switched off they must agree to fp32 round-off."""
    import ref_cpu as R
    cfg = R.make_config("tiny")
    sd = R.make_state_dict(cfg, seed=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 17, cfg.hidden_dim, generator=g)
    p = "transformer.0.fusion.0.attn.fn"
    ref = R.cls_cross_attention(sd, p, x, cfg.num_heads)
    prev, R._QUANT = R._QUANT, (lambda t: t)
    try:
        low = R.cls_cross_attention(sd, p, x, cfg.num_heads)
    finally:
        R._QUANT = prev
    assert float((low - ref).norm() / ref.norm()) < 2e-6
