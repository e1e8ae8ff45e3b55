"""GPU: training-mode dropout.  The masks are a pure function of (seed, element index), so the tests
extract the very masks the kernels used (xvit_dropout on a tensor of ones with the same seed) and feed
them to the CPU oracle: forward values AND gradients must then agree like in the p = 0 tests."""
import pytest
import torch

import ref_cpu as R
from _util import note, assert_close, dev, randn, rel, rt

pytestmark = pytest.mark.gpu


def _mask(shape, p, seed):
    from xvit import ops
    return ops.dropout(torch.ones(*shape, device=dev()), p, seed).cpu()


@pytest.mark.parametrize("M,N,K,split", [(1026, 768, 256, 1), (130, 192, 128, 1), (32, 768, 1024, 4)])
def test_gemm_epilogue_dropout_uses_the_shared_mask(M, N, K, split):
    from xvit import ops
    p, seed = 0.3, 123456789
    a, w, b, r = rt(randn(M, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5)), randn(N, seed=3), randn(M, N, seed=4)
    C = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, bias=b.to(dev()), residual=r.to(dev()), dropout=(p, seed), split_k=split)
    mask = _mask((M, N), p, seed)
    assert abs(float((mask != 0).float().mean()) - (1 - p)) < 0.02
    assert_close(C, (a @ w.T + b) * mask + r, "dropout before the residual")


def test_cls_xattn_probability_dropout():
    from xvit import ops
    B, H, N, d = 2, 3, 65, 192
    p, seed, scale = 0.25, 42, 0.125
    qv, kv = rt(randn(B, d, seed=1)), rt(randn(B, N, 2 * d, seed=2))
    o, probs = ops.cls_xattn_fwd(qv.to(dev(), torch.bfloat16), kv.to(dev(), torch.bfloat16).reshape(B * N, 2 * d), B, N, H, scale, dropout=(p, seed))
    mask = _mask((B, H, N), p, seed)
    q = qv.reshape(B, 1, H, 64).permute(0, 2, 1, 3).clone().requires_grad_()
    k, v = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3).clone().requires_grad_() for t in kv.split(d, dim=-1))
    pr = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)           # [B,H,1,N]
    o_ref = (pr * mask[:, :, None, :]) @ v
    assert_close(o, o_ref.permute(0, 2, 1, 3).reshape(B, d), "cls xattn fwd with attn_drop")
    do = rt(randn(B, d, seed=3))
    o_ref.backward(do.reshape(B, 1, H, 64).permute(0, 2, 1, 3))
    dq, dkv = ops.cls_xattn_bwd(qv.to(dev(), torch.bfloat16), kv.to(dev(), torch.bfloat16).reshape(B * N, 2 * d), probs, do.to(dev(), torch.bfloat16), B, N, H, scale, dropout=(p, seed))
    assert_close(dq, q.grad.permute(0, 2, 1, 3).reshape(B, d), "dq")
    dk, dv = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in dkv.float().cpu().reshape(B, N, 2 * d).split(d, dim=-1))
    assert rel(dk, k.grad) < 4e-3 and rel(dv, v.grad) < 4e-3


def test_self_attention_block_training_dropout_matches_oracle_with_same_masks():
    import xvit
    import xvit.functional as XF
    cfg = R.make_config("small", dropout=0.2)
    sd = R.make_state_dict(cfg, seed=3)
    pfx = "transformer.0.blocks.0.0"
    blk = xvit.SelfAttentionBlock(cfg).to(dev())
    blk.load_state_dict({k[len(pfx) + 1:]: v for k, v in sd.items() if k.startswith(pfx + ".")})
    blk.train()
    B, N, d, f, H = 2, 65, cfg.hidden_dim, cfg.mlp_dim, cfg.num_heads
    x = randn(B, N, d, seed=9)
    XF._DROP_CALLS = 1000
    xr = x.to(dev()).requires_grad_()
    y = blk(xr)
    y.square().sum().backward()
    XF._DROP_CALLS = 1000
    s_o, s_a, s_f = XF.drop_seeds(3)
    m_o, m_a, m_f = _mask((B * N, d), 0.2, s_o).reshape(B, N, d), _mask((B * N, f), 0.2, s_a).reshape(B, N, f), _mask((B * N, d), 0.2, s_f).reshape(B, N, d)
    # oracle with the same masks (model_cross.py:69-72 with Dropout at :47, :25, :27)
    xo = x.clone().requires_grad_()
    a1 = R.self_attention(sd, pfx + ".attn.fn", R.layer_norm(xo, sd[pfx + ".attn.norm.weight"], sd[pfx + ".attn.norm.bias"]), H)
    x1 = xo + a1 * m_o
    h2 = R.layer_norm(x1, sd[pfx + ".ffn.norm.weight"], sd[pfx + ".ffn.norm.bias"])
    act = R.gelu(R.linear(h2, sd[pfx + ".ffn.fn.net.0.weight"], sd[pfx + ".ffn.fn.net.0.bias"])) * m_a
    x2 = x1 + R.linear(act, sd[pfx + ".ffn.fn.net.3.weight"], sd[pfx + ".ffn.fn.net.3.bias"]) * m_f
    x2.square().sum().backward()
    assert rel(y, x2) < 6e-3, rel(y, x2)
    assert rel(xr.grad, xo.grad) < 2e-2, rel(xr.grad, xo.grad)
    # eval mode: dropout off, and equal to the p = 0 block
    blk.eval()
    y_eval = blk(x.to(dev()))
    cfg0 = R.make_config("small")
    assert rel(y_eval, R.self_block(sd, pfx, x, H)) < 6e-3 and cfg0.dropout == 0.0


def test_model_cross_trains_with_reference_dropout_rates():
    """The reference trains with dropout 0.1-0.25 (main_mist.py:71-77): the drop-in must run there."""
    import xvit
    import xvit.functional as XF
    cfg = R.make_config("tiny", dropout=0.25)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    img, labels = R.make_inputs(cfg, 4, seed=0)
    model.train()
    XF._DROP_CALLS = 77
    l1, loss1 = model(img.to(dev()), labels.to(dev()))
    loss1.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g1.values())
    l2, _ = model(img.to(dev()), labels.to(dev()))
    assert not torch.equal(l1, l2)                                   # a new mask every call
    model.zero_grad()
    XF._DROP_CALLS = 77
    l3, loss3 = model(img.to(dev()), labels.to(dev()))               # same seed state -> same masks -> same result
    loss3.backward()
    assert torch.equal(l1, l3)
    for k, p in model.named_parameters():
        assert rel(p.grad, g1[k]) < 1e-3 or float(g1[k].abs().max()) < 1e-6, k
    model.eval()
    e1, _ = model(img.to(dev()), labels.to(dev()))
    e2, _ = model(img.to(dev()), labels.to(dev()))
    assert torch.equal(e1, e2)
    ref_logits, _ = R.model_cross_forward(R.make_state_dict(cfg, seed=0), img, labels, R.make_config("tiny"))
    assert note("dropout.eval_logits_vs_fp32", rel(e1, ref_logits)) < 1.1e-2      # measured 7.4e-3 (tiny config, eval mode, vs the fp32 oracle)


@pytest.mark.parametrize("B,H,N", [(2, 3, 65), (1, 2, 200), (1, 12, 513)])
def test_flash_attention_probability_dropout_exact_mask(B, H, N):
    """model.py:169 drops attention PROBABILITIES.  The fused kernels apply a counter-hash mask keyed by (seed, b, h, q, k): the
    very mask is extracted (xvit_dropout on ones [B, H, N, N]) and fed to the oracle; forward and all three gradients agree."""
    from xvit import ops
    d, p, seed, scale = H * 64, 0.3, 987654321, 0.125
    qkv = rt(randn(B, N, 3 * d, seed=N))
    do = rt(randn(B, N, d, seed=N + 1))
    qd = qkv.to(dev(), torch.bfloat16).reshape(B * N, 3 * d)
    o, lse = ops.attn_fwd(qd, B, N, H, scale, dropout=(p, seed))
    dqkv = ops.attn_bwd(qd, o, do.to(dev(), torch.bfloat16).reshape(B * N, d), lse, B, N, H, scale, dropout=(p, seed))
    mask = _mask((B, H, N, N), p, seed)                      # 0 or 1/(1-p)
    assert abs(float((mask != 0).float().mean()) - (1 - p)) < 0.02
    heads = lambda t: t.reshape(B, N, H, 64).permute(0, 2, 1, 3)   # noqa: E731
    q, k, v = (heads(t).clone().requires_grad_() for t in qkv.split(d, dim=-1))
    pr = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    o_ref = (pr * mask) @ v
    assert_close(heads(o.reshape(B, N, d)), o_ref, "flash fwd with probability dropout")
    assert_close(lse, torch.logsumexp((q @ k.transpose(-1, -2)) * scale, dim=-1), "lse is that of the un-dropped softmax")
    o_ref.backward(heads(do))
    dq, dk, dv = (heads(t) for t in dqkv.float().cpu().reshape(B, N, 3 * d).split(d, dim=-1))
    assert rel(dv, v.grad) < 4e-3 and rel(dk, k.grad) < 6e-3 and rel(dq, q.grad) < 6e-3, (rel(dv, v.grad), rel(dk, k.grad), rel(dq, q.grad))
    # p = 0 launches the plain kernels: bit-identical to a call without the argument
    o0, _ = ops.attn_fwd(qd, B, N, H, scale)
    o1, _ = ops.attn_fwd(qd, B, N, H, scale, dropout=(0.0, seed))
    assert torch.equal(o0, o1)


def test_encoder_trains_with_attention_probability_dropout():
    """The model.py twin with attention_dropout_rate > 0 (model.py:169, :177) in training mode: runs, is reproducible under the
    same seed state, differs between calls, and eval mode equals the p = 0 encoder."""
    import xvit
    import xvit.functional as XF
    from types import SimpleNamespace
    cfg = SimpleNamespace(hidden_size=256, transformer=dict(num_heads=4, mlp_dim=512, dropout_rate=0.1, attention_dropout_rate=0.1, num_layers=2))
    torch.manual_seed(0)
    enc = xvit.Encoder(cfg).to(dev()).train()
    x = randn(2, 17, 256, seed=1).to(dev())
    XF._DROP_CALLS = 500
    y1 = enc(x)
    y1.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())
    y2 = enc(x)
    assert not torch.equal(y1, y2)
    XF._DROP_CALLS = 500
    assert torch.equal(enc(x), y1)
    cfg0 = SimpleNamespace(hidden_size=256, transformer=dict(num_heads=4, mlp_dim=512, dropout_rate=0.0, attention_dropout_rate=0.0, num_layers=2))
    enc0 = xvit.Encoder(cfg0).to(dev()).train()
    enc0.load_state_dict(enc.state_dict())
    enc.eval()
    assert torch.equal(enc(x), enc0(x))
    # stand-alone MultiHeadAttention takes the same path
    mha = xvit.MultiHeadAttention(cfg).to(dev()).train()
    mha(x).sum().backward()
