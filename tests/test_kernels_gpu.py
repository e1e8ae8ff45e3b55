"""GPU parity for the non-GEMM kernels: LayerNorm, fused attention fwd/bwd, CLS cross-attention,
patchify and the small helpers — each against the CPU oracle on the same bf16-rounded inputs."""
import math

import pytest
import torch

import ref_cpu as R
from _util import assert_close, dev, randn, rel, rt

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _peel_on_any_grid():
    """The kernel tests run tiny batches: take the CLS-peel form of the attention kernels whenever the SHAPE qualifies (N = 64 m + 1),
    not only on the large grids the default heuristic picks it for."""
    from xvit import ops
    ops.set_option("attn_peel", 2)
    yield
    ops.set_option("attn_peel", 1)


def _ops():
    from xvit import ops
    return ops


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,d", [(7, 192), (1026, 768), (130, 256), (65, 1024), (9, 2048)])
def test_layernorm_fwd_bwd(rows, d):
    ops = _ops()
    x, g, b = randn(rows, d, seed=1) * 2 + 0.3, 1 + 0.1 * randn(d, seed=2), 0.1 * randn(d, seed=3)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev()), g.to(dev()), b.to(dev()), 1e-5)
    ref = R.layer_norm(x, g, b, 1e-5)
    assert_close(y, ref, "ln fwd")
    assert_close(mean, x.mean(-1), "mean")
    # backward against autograd (dy is bf16 on the device: feed the oracle the same rounded values)
    dy, dres = rt(randn(rows, d, seed=4)), randn(rows, d, seed=5)
    xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    R.layer_norm(xr, gr, br, 1e-5).backward(dy)
    dg = torch.zeros(d, device=dev()); db = torch.zeros(d, device=dev())
    sx = torch.zeros(d, device=dev()); sr = torch.zeros(d, device=dev())
    dx, dxb = ops.layernorm_bwd(dy.to(dev(), torch.bfloat16), x.to(dev()), mean, rstd, g.to(dev()), dg, db, dres=dres.to(dev()), want_bf16=True,
                                dxsum=sx, dressum=sr)
    assert_close(dx, xr.grad + dres, "ln dx (+residual)")
    assert_close(sx, (xr.grad + dres).sum(0), "column sums of dx")
    assert_close(sr, dres.sum(0), "column sums of dres")
    assert_close(dxb, xr.grad + dres, "ln dx bf16 copy")
    assert_close(dg, gr.grad, "dgamma")
    assert_close(db, br.grad, "dbeta")


def test_layernorm_row0_from_other_tensor():
    """rows with n == 0 come from x_alt: the cat(cls_i, patches_j) of model_cross.py:140."""
    ops = _ops()
    B, N, d = 3, 17, 192
    xi, xj = randn(B, N, d, seed=1), randn(B, N, d, seed=2)
    g, b = 1 + 0.1 * randn(d, seed=3), 0.1 * randn(d, seed=4)
    y, _, _ = ops.layernorm_fwd(xj.to(dev()).reshape(B * N, d), g.to(dev()), b.to(dev()), 1e-5, x_alt=xi.to(dev()).reshape(B * N, d), seq_len=N)
    cat = torch.cat((xi[:, 0:1], xj[:, 1:]), dim=1)
    assert_close(y.reshape(B, N, d), R.layer_norm(cat, g, b), "ln with x_alt")
    # the same rows from a packed [B, d] copy of the CLS rows (the in-place CLS splice keeps only that copy)
    packed = xi[:, 0].contiguous().to(dev())
    y2, mean2, rstd2 = ops.layernorm_fwd(xj.to(dev()).reshape(B * N, d), g.to(dev()), b.to(dev()), 1e-5, x_alt=packed, seq_len=N)
    assert torch.equal(y2, y)
    dy = randn(B * N, d, seed=9).to(dev(), torch.bfloat16)
    outs = []
    for alt in (xi.to(dev()).reshape(B * N, d), packed):
        dg, db = torch.zeros(d, device=dev()), torch.zeros(d, device=dev())
        dx, _ = ops.layernorm_bwd(dy, xj.to(dev()).reshape(B * N, d), mean2, rstd2, g.to(dev()), dg, db, x_alt=alt, seq_len=N)
        outs.append(dx)
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,N", [(2, 3, 17), (2, 2, 65), (1, 4, 130), (2, 12, 513), (1, 2, 64), (1, 1, 128), (1, 2, 1),
                                   (3, 2, 129), (2, 3, 193), (1, 2, 1025)])   # N = 64 m + 1: the CLS-peel form (include/xvit.h)
def test_attention_fwd_bwd(B, H, N):
    ops = _ops()
    d = H * 64
    qkv = rt(randn(B, N, 3 * d, seed=N))
    scale = 64 ** -0.5
    o, lse = ops.attn_fwd(qkv.to(dev(), torch.bfloat16).reshape(B * N, 3 * d), B, N, H, scale)
    q, k, v = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in qkv.split(d, dim=-1))
    qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
    o_ref, lse_ref = R.softmax_attention(qr, kr, vr, scale)
    assert_close(o.reshape(B, N, H, 64).permute(0, 2, 1, 3), o_ref, f"attn fwd N={N}")
    assert_close(lse, lse_ref, "lse")
    do = rt(randn(B, N, d, seed=N + 1))
    o_ref.backward(do.reshape(B, N, H, 64).permute(0, 2, 1, 3))
    # the kernel recomputes P from ITS bf16 O; use the device O for delta like the real pipeline does
    dqkv = ops.attn_bwd(qkv.to(dev(), torch.bfloat16).reshape(B * N, 3 * d), o, do.to(dev(), torch.bfloat16).reshape(B * N, d), lse, B, N, H, scale)
    dq, dk, dv = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in dqkv.float().cpu().reshape(B, N, 3 * d).split(d, dim=-1))
    assert rel(dv, vr.grad) < 4e-3, rel(dv, vr.grad)
    if N == 1:  # a single key: softmax == 1, so dk = dq = 0 exactly; only round-off remains
        assert float(dk.abs().max()) < 1e-5 and float(dq.abs().max()) < 1e-5
    else:
        assert rel(dk, kr.grad) < 6e-3, rel(dk, kr.grad)
        assert rel(dq, qr.grad) < 6e-3, rel(dq, qr.grad)


def _attn_bwd_emulated(q, k, v, o_dev, do, lse, scale):
    """The backward kernels' arithmetic on the CPU: P recomputed from the forward's lse, delta from the DEVICE o (bf16), and
    P / dS rounded to bf16 where they feed the second MFMA of their product (attention.hip: dV^T += dO^T P, dK^T += Q^T dS,
    dQ += dS K).  fp32 everywhere else."""
    s = (q @ k.transpose(-1, -2)) * scale
    p = torch.exp(s - lse[..., None])
    dp = do @ v.transpose(-1, -2)
    delta = (do * o_dev).sum(-1, keepdim=True)
    ds = p * (dp - delta)
    dv = rt(p).transpose(-1, -2) @ do
    dk = rt(ds).transpose(-1, -2) @ q * scale
    dq = rt(ds) @ k * scale
    return dq, dk, dv


@pytest.mark.parametrize("B,H,N", [(2, 12, 513), (1, 2, 3376), (1, 2, 4097), (2, 3, 130), (2, 2, 193)])
def test_attention_bwd_vs_bf16_emulating_oracle(B, H, N):
    """Gate with the kernels' own rounding points emulated (bf16 P and dS into the second product): 3e-3 = 1e-3 + one bf16
    output rounding, also at the long-sequence tile counts of configs[2] (N = 3376: 53 key tiles) and configs[4]
    (N = 4097: 65 key tiles, several rounds of the XCD block remap)."""
    ops = _ops()
    d = H * 64
    scale = 64 ** -0.5
    qkv = rt(randn(B, N, 3 * d, seed=N + 7))
    do = rt(randn(B, N, d, seed=N + 8))
    qd = qkv.to(dev(), torch.bfloat16).reshape(B * N, 3 * d)
    o, lse = ops.attn_fwd(qd, B, N, H, scale)
    dqkv = ops.attn_bwd(qd, o, do.to(dev(), torch.bfloat16).reshape(B * N, d), lse, B, N, H, scale)
    heads = lambda t: t.reshape(B, N, H, 64).permute(0, 2, 1, 3)   # noqa: E731
    q, k, v = (heads(t) for t in qkv.split(d, dim=-1))
    rq, rk, rv = _attn_bwd_emulated(q, k, v, heads(o.float().cpu().reshape(B, N, d)), heads(do), lse.cpu(), scale)
    dq, dk, dv = (heads(t) for t in dqkv.float().cpu().reshape(B, N, 3 * d).split(d, dim=-1))
    for name, got, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        assert rel(got, ref) < 3e-3, (name, rel(got, ref))


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump at a late key tile (guide rule 26): one key aligned with one
    query and scaled up, placed in the last tile."""
    ops = _ops()
    B, H, N = 1, 1, 200
    qkv = rt(randn(B, N, 192, seed=5))
    qkv[0, 190, 64:128] = rt(qkv[0, 3, 0:64] * 6.0)   # k[190] = 6 * q[3]
    o, lse = ops.attn_fwd(qkv.to(dev(), torch.bfloat16).reshape(N, 192), B, N, H, 0.125)
    q, k, v = (t.reshape(B, N, 1, 64).permute(0, 2, 1, 3) for t in qkv.split(64, dim=-1))
    o_ref, lse_ref = R.softmax_attention(q, k, v, 0.125)
    assert_close(o.reshape(B, N, 1, 64).permute(0, 2, 1, 3), o_ref, "rescale branch")
    assert_close(lse, lse_ref, "rescale lse")


@pytest.mark.parametrize("B,H,N", [(2, 3, 193), (2, 12, 513)])
def test_attention_cls_peel_matches_tile_grid_form(B, H, N):
    """N = 64 m + 1 runs with token 0 off the tile grid (xvit_attn_fwd_workspace_bytes > 0); xvit_set_option("attn_peel", 0)
    keeps it on the grid.  Same function, different summation order / rounding points for token 0's row and column: outputs
    agree to bf16 round-off, every row and column included (row 0 and key 0 are checked on their own)."""
    ops = _ops()
    d = H * 64
    scale = 0.125
    qd = rt(randn(B, N, 3 * d, seed=N + 3)).to(dev(), torch.bfloat16).reshape(B * N, 3 * d)
    dod = rt(randn(B, N, d, seed=N + 4)).to(dev(), torch.bfloat16).reshape(B * N, d)
    outs = []
    for peel in (2, 0):
        ops.set_option("attn_peel", peel)
        try:
            o, lse = ops.attn_fwd(qd, B, N, H, scale)
            dqkv = ops.attn_bwd(qd, o, dod, lse, B, N, H, scale)
        finally:
            ops.set_option("attn_peel", 2)
        outs.append((o.float().reshape(B, N, d), lse, dqkv.float().reshape(B, N, 3 * d)))
    (o1, l1, g1), (o0, l0, g0) = outs
    assert rel(o1, o0) < 3e-3 and rel(o1[:, 0], o0[:, 0]) < 3e-3, (rel(o1, o0), rel(o1[:, 0], o0[:, 0]))
    assert float((l1 - l0).abs().max()) < 1e-4
    assert rel(g1, g0) < 4e-3 and rel(g1[:, 0], g0[:, 0]) < 4e-3, (rel(g1, g0), rel(g1[:, 0], g0[:, 0]))
    # bit-reproducible: the partials are merged in slot order
    o2, lse2 = ops.attn_fwd(qd, B, N, H, scale)
    assert torch.equal(o2.float().reshape(B, N, d), o1) and torch.equal(lse2, l1)
    assert torch.equal(ops.attn_bwd(qd, o2, dod, lse2, B, N, H, scale).float().reshape(B, N, 3 * d), g1)


def test_attention_cls_peel_extreme_scores():
    """Rule 26 for the peeled form: (a) the CLS key dominates some rows (the initial state IS the maximum and every later tile
    rescales nothing), (b) a late patch key dominates others (the initial state is rescaled away), (c) the CLS query's maximum
    sits in one wave's key block (the merge weights of all other partials underflow towards 0)."""
    ops = _ops()
    B, H, N = 1, 1, 193
    qkv = rt(randn(B, N, 192, seed=11))
    qkv[0, 0, 64:128] = rt(qkv[0, 7, 0:64] * 6.0)      # (a) k[0] = 6 q[7]
    qkv[0, 180, 64:128] = rt(qkv[0, 40, 0:64] * 6.0)   # (b) k[180] = 6 q[40]
    qkv[0, 100, 64:128] = rt(qkv[0, 0, 0:64] * 6.0)    # (c) k[100] = 6 q[0]
    qd = qkv.to(dev(), torch.bfloat16).reshape(N, 192)
    o, lse = ops.attn_fwd(qd, B, N, H, 0.125)
    q, k, v = (t.reshape(B, N, 1, 64).permute(0, 2, 1, 3).clone().requires_grad_() for t in qkv.split(64, dim=-1))
    o_ref, lse_ref = R.softmax_attention(q, k, v, 0.125)
    assert_close(o.reshape(B, N, 1, 64).permute(0, 2, 1, 3), o_ref, "peel, extreme scores")
    assert_close(lse, lse_ref, "peel, extreme scores: lse")
    do = rt(randn(B, N, 64, seed=12))
    o_ref.backward(do.reshape(B, N, 1, 64).permute(0, 2, 1, 3))
    dqkv = ops.attn_bwd(qd, o, do.to(dev(), torch.bfloat16).reshape(N, 64), lse, B, N, H, 0.125).float().cpu().reshape(B, N, 192)
    dq, dk, dv = (t.reshape(B, N, 1, 64).permute(0, 2, 1, 3) for t in dqkv.split(64, dim=-1))
    assert rel(dq, q.grad) < 6e-3 and rel(dk, k.grad) < 6e-3 and rel(dv, v.grad) < 4e-3, (rel(dq, q.grad), rel(dk, k.grad), rel(dv, v.grad))
    # the rows the spikes touch, one by one (a saturated softmax row has dq ~ 0: measure against the typical row norm)
    nq, nk = float(q.grad.norm()) / N ** 0.5, float(k.grad.norm()) / N ** 0.5
    for row in (0, 7, 40, 100, 180):
        eq = float((dq[0, 0, row] - q.grad[0, 0, row]).norm()) / (float(q.grad[0, 0, row].norm()) + 0.05 * nq)
        ek = float((dk[0, 0, row] - k.grad[0, 0, row]).norm()) / (float(k.grad[0, 0, row].norm()) + 0.05 * nk)
        assert eq < 2e-2 and ek < 2e-2, (row, eq, ek)


# --------------------------------------------------------------------------- CLS cross-attention
@pytest.mark.parametrize("B,H,N", [(2, 3, 17), (3, 4, 65), (2, 12, 513), (1, 2, 700)])
def test_cls_xattn_fwd_bwd(B, H, N):
    ops = _ops()
    d = H * 64
    scale = 0.125
    qv, kv = rt(randn(B, d, seed=1)), rt(randn(B, N, 2 * d, seed=2))
    o, p = ops.cls_xattn_fwd(qv.to(dev(), torch.bfloat16), kv.to(dev(), torch.bfloat16).reshape(B * N, 2 * d), B, N, H, scale)
    q = qv.reshape(B, 1, H, 64).permute(0, 2, 1, 3).clone().requires_grad_()
    k, v = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3).clone().requires_grad_() for t in kv.split(d, dim=-1))
    o_ref, _ = R.softmax_attention(q, k, v, scale)
    assert_close(o, o_ref.permute(0, 2, 1, 3).reshape(B, d), "cls xattn fwd")
    p_ref = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1).squeeze(2)
    assert_close(p, p_ref, "cls xattn probs")
    do = rt(randn(B, d, seed=3))
    o_ref.backward(do.reshape(B, 1, H, 64).permute(0, 2, 1, 3))
    dq, dkv = ops.cls_xattn_bwd(qv.to(dev(), torch.bfloat16), kv.to(dev(), torch.bfloat16).reshape(B * N, 2 * d), p, do.to(dev(), torch.bfloat16), B, N, H, scale)
    assert_close(dq, q.grad.permute(0, 2, 1, 3).reshape(B, d), "dq")
    dk, dv = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in dkv.float().cpu().reshape(B, N, 2 * d).split(d, dim=-1))
    assert rel(dk, k.grad) < 3e-3 and rel(dv, v.grad) < 3e-3


# ------------------------------------------------------------------------------------ patchify
@pytest.mark.parametrize("img_size,patch,dtype", [((32, 32, 2), (8, 8, 2), torch.float32), ((16, 24, 32), (8, 8, 16), torch.float32),
                                                   ((32, 32, 16), (8, 8, 8), torch.bfloat16), ((32, 16, 32), (16, 16, 16), torch.float32)])
def test_patchify_bit_exact(img_size, patch, dtype):
    ops = _ops()
    B, M = 2, 3
    img = randn(B, M, 1, *img_size, seed=3).to(dtype)
    out = ops.patchify(img.to(dev()), patch)
    for m in range(M):
        ref = R.patchify(img[:, m, 0].float(), patch).to(torch.bfloat16)  # pure permutation + one rounding
        assert torch.equal(out[m].cpu().reshape(ref.shape), ref), f"patchify modality {m}"


# ------------------------------------------------------------------------------------- helpers
def test_colsum_cast_embed_small_linear_ce_dropout():
    ops = _ops()
    x = rt(randn(1026, 768, seed=1))
    assert_close(ops.colsum(x.to(dev(), torch.bfloat16)), x.sum(0), "colsum bf16")
    acc = torch.ones(768, device=dev())
    ops.colsum(x.to(dev()), out=acc, accumulate=True)
    assert_close(acc, x.sum(0) + 1, "colsum fp32 accumulate")
    w = randn(1000, 37, seed=2)
    assert torch.equal(ops.cast_bf16(w.to(dev())).cpu(), w.to(torch.bfloat16))
    # embed: cls row + pos/cls grads
    MB, N, d = 6, 17, 192
    cls, pos = randn(d, seed=3), randn(N, d, seed=4)
    xx = torch.zeros(MB, N, d, device=dev())
    ops.cls_row_fwd(cls.to(dev()), pos.to(dev()), xx, MB, N, d)
    assert_close(xx[:, 0].cpu(), (cls + pos[0]).expand(MB, d), "cls row")
    assert float(xx[:, 1:].abs().max()) == 0.0
    dx = randn(MB, N, d, seed=5)
    dpos, dcls = torch.zeros(N, d, device=dev()), torch.zeros(d, device=dev())
    ops.embed_bwd(dx.to(dev()), dpos, dcls, MB, N, d)
    assert_close(dpos, dx.sum(0), "dpos"); assert_close(dcls, dx[:, 0].sum(0), "dcls")
    # tiny head linear
    xh, W, b = rt(randn(5, 768, seed=6)), randn(2, 768, seed=7), randn(2, seed=8)
    y = ops.small_linear_fwd(xh.to(dev(), torch.bfloat16), W.to(dev()), b.to(dev()))
    assert_close(y, xh @ W.T + b, "small linear fwd")
    dy = randn(5, 2, seed=9)
    dW, dbb = torch.zeros(2, 768, device=dev()), torch.zeros(2, device=dev())
    dxh = ops.small_linear_bwd(dy.to(dev()), xh.to(dev(), torch.bfloat16), W.to(dev()), dW, dbb)
    assert_close(dxh, dy @ W, "small linear dx"); assert_close(dW, dy.T @ xh, "small linear dW"); assert_close(dbb, dy.sum(0), "small linear db")
    # mean + cross-entropy (label smoothing on and off)
    for eps in (0.0, 0.1):
        lm = randn(3, 7, 2, seed=10).requires_grad_()
        labels = torch.tensor([0, 1, 1, 0, 1, 0, 0])
        logits, loss, dl = ops.mean_ce(lm.detach().to(dev()), labels.to(dev()), eps)
        ref_logits = lm.mean(0)
        ref_loss = R.cross_entropy(ref_logits, labels, eps)
        ref_loss.backward()
        assert_close(logits, ref_logits, "mean logits"); assert abs(float(loss) - float(ref_loss)) < 1e-5
        assert_close(dl, lm.grad, "dlogits")
    # dropout: deterministic in (seed, index), right keep rate and scaling
    xd = torch.ones(1 << 20, device=dev())
    y1, y2, y3 = ops.dropout(xd, 0.25, 1234), ops.dropout(xd, 0.25, 1234), ops.dropout(xd, 0.25, 99)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = float((y1 != 0).float().mean())
    assert abs(keep - 0.75) < 5e-3 and abs(float(y1.max()) - 1 / 0.75) < 1e-6


@pytest.mark.parametrize("src,dst", [((240, 240, 155), (128, 128, 128)), ((20, 31, 17), (32, 32, 16)), ((9, 8, 8), (8, 8, 12))])
def test_input_stage_resize_pad_crop_int16(src, dst):
    """int16 volumes -> bf16 model input, against the oracle's restatement of the MONAI rule (bit-exact up to the one
    bf16 rounding of integer intensities)."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    vol = torch.randint(-50, 3000, (2, 2, *src), generator=g, dtype=torch.int16)
    out = ops.resize_pad_crop_i16(vol.to(dev()), dst, -1.0)
    ref = R.resize_with_pad_or_crop(vol, dst, -1.0).to(torch.bfloat16).reshape(2, 2, 1, *dst)
    assert out.shape == ref.shape and torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("B,N,H", [(3, 17, 3), (5, 513, 12), (2, 130, 4), (1, 1000, 16)])
def test_xattn_kv_dgrad_from_rank_one_coefficients(B, N, H):
    """dK and dV of the CLS-query attention are rank one per (sample, head): xvit_cls_xattn_bwd's coefficients reproduce the dense dk / dv,
    and xvit_xattn_kv_dgrad (with R from xvit_head_rows on the fp32 weights) gives dhn = dkv Wkv (reference model_cross.py:92-99) — checked
    against the dense chain in fp64."""
    ops = _ops()
    d = 64 * H
    scale = 0.125
    qv, kv = rt(randn(B, d, seed=1)), rt(randn(B, N, 2 * d, seed=2))
    do = rt(randn(B, d, seed=3))
    wkv = randn(2 * d, d, seed=4, scale=d ** -0.5)
    gq, gkv = qv.to(dev(), torch.bfloat16), kv.to(dev(), torch.bfloat16).reshape(B * N, 2 * d)
    p = ops.cls_xattn_fwd(gq, gkv, B, N, H, scale)[1]
    dq1, dkv = ops.cls_xattn_bwd(gq, gkv, p, do.to(dev(), torch.bfloat16), B, N, H, scale)
    dq2, coef = ops.cls_xattn_bwd(gq, gkv, p, do.to(dev(), torch.bfloat16), B, N, H, scale, low_rank=True)
    assert torch.equal(dq1, dq2)
    c = coef.double().cpu()
    dk = (c[:, :, :H, None] * qv.double().view(B, 1, H, 64)).reshape(B * N, d)
    dv = (c[:, :, H:, None] * do.double().view(B, 1, H, 64)).reshape(B * N, d)
    dense = torch.cat((dk, dv), dim=1)
    assert_close(dkv, dense.float(), "dkv from the coefficients")
    Rm = torch.empty(2 * H, B, d, dtype=torch.float32, device=dev())
    gw = wkv.to(dev())
    ops.head_rows(qv.to(dev()), gw[:d].contiguous(), Rm[:H].transpose(0, 1), H)
    ops.head_rows(do.to(dev()), gw[d:].contiguous(), Rm[H:].transpose(0, 1), H)
    dhn = ops.xattn_kv_dgrad(coef, Rm, B, N, H, d)
    assert_close(dhn, (dense @ wkv.double()).float(), "dhn = dkv Wkv")


def test_rows_combine_all_dtype_and_stride_combinations():
    """xvit_rows_combine: the CLS-row bookkeeping of the fusions (copy out / write back / add / zero the B CLS rows of a [B, N, d] tensor,
    fp32 or bf16, optionally into two destinations) against torch indexing — bit-exact."""
    ops = _ops()
    B, N, d = 5, 7, 200
    g = torch.Generator().manual_seed(0)
    tok = torch.randn(B, N, d, generator=g).to(dev())
    tokb = torch.randn(B, N, d, generator=g).to(dev()).bfloat16()
    vec = torch.randn(B, d, generator=g).to(dev())
    vecb = torch.randn(B, d, generator=g).to(dev()).bfloat16()
    # copy the CLS rows out (strided source, packed destination)
    out = ops.rows_combine(torch.empty(B, d, device=dev()), a=tok[:, 0])
    assert torch.equal(out, tok[:, 0])
    # write rows back in place of the CLS rows; the rest of the tensor is untouched
    t2 = tok.clone()
    ops.rows_combine(t2.reshape(B, N * d)[:, :d], a=vec)
    assert torch.equal(t2[:, 0], vec) and torch.equal(t2[:, 1:], tok[:, 1:])
    # strided = strided + packed, second destination in bf16
    t3, t3b = tok.clone(), tokb.clone()
    ops.rows_combine(t3[:, 0], a=t3[:, 0], b=vec, dst2=t3b[:, 0])
    assert torch.equal(t3[:, 0], tok[:, 0] + vec) and torch.equal(t3b[:, 0], (tok[:, 0] + vec).bfloat16()) and torch.equal(t3b[:, 1:], tokb[:, 1:])
    # bf16 += bf16 through fp32, one rounding
    t4 = tokb.clone()
    ops.rows_combine(t4.reshape(B, N * d)[:, :d], a=t4.reshape(B, N * d)[:, :d], b=vecb)
    assert torch.equal(t4[:, 0], (tokb[:, 0].float() + vecb.float()).bfloat16())
    # zero rows in both dtypes
    t5, t5b = tok.clone(), tokb.clone()
    ops.rows_combine(t5[:, 0], dst2=t5b[:, 0])
    assert float(t5[:, 0].abs().max()) == 0.0 and float(t5b[:, 0].float().abs().max()) == 0.0 and torch.equal(t5[:, 1:], tok[:, 1:])
    with pytest.raises(AssertionError):
        ops.rows_combine(tok[:, :, 0], a=vec[:, :N])      # last stride must be one
