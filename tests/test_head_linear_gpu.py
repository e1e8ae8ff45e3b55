"""GPU parity: the per-head fp32 products and the token-softmax kernels of the fusion's low-rank key / value form
(csrc/head_linear.hip; reference model_cross.py:88-99) against float64 einsums, and the whole low-rank attention
against the literal order (k = x Wk^T + bk, v = x Wv^T + bv, softmax(q k^T) v) it replaces."""
import pytest
import torch

from _util import dev, randn, rel, rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H", [(126, 12), (3, 3), (33, 4), (1, 16)])
def test_head_rows_cols_wgrad(B, H):
    from xvit import ops
    d = 64 * H
    x, W = randn(B, d, seed=1), randn(d, d, seed=2, scale=d ** -0.5)
    # rows: out[b, h, :] = x[b, h-slice] @ W[h-slice, :]
    R = torch.zeros(2 * H, B, d, device=dev())
    ob = torch.full((B, 16, d), 7.0, dtype=torch.bfloat16, device=dev())          # the kernel zeroes the padding heads itself
    ops.head_rows(x.to(dev()), W.to(dev()), R[H:].transpose(0, 1), H, out_bf16=ob)
    ref = torch.einsum("bhe,hec->bhc", x.double().view(B, H, 64), W.double().view(H, 64, d))
    assert rel(R[H:].transpose(0, 1), ref) < 2e-6 and float(R[:H].abs().max()) == 0.0
    assert torch.equal(ob[:, :H], R[H:].transpose(0, 1).to(torch.bfloat16)) and (H == 16 or float(ob[:, H:].abs().max()) == 0.0)
    # cols: out[b, 64h+e] = rs[b, h] * t[b, h, :] . W[64h+e, :] + bias
    t, rs, bias = randn(B, 16, d, seed=3), randn(B, H, seed=4).abs() + 0.5, randn(d, seed=5)
    out, outb = ops.head_cols(t.to(dev()), W.to(dev()), H, row_scale=rs.to(dev()), bias=bias.to(dev()), want_bf16=True)
    ref = (torch.einsum("bhc,hec->bhe", t[:, :H].double(), W.double().view(H, 64, d)) * rs.double()[:, :, None]).reshape(B, d) + bias.double()
    assert rel(out, ref) < 2e-6 and torch.equal(outb, out.to(torch.bfloat16))
    out2, _ = ops.head_cols(t.to(dev()), W.to(dev()), H)
    assert rel(out2, torch.einsum("bhc,hec->bhe", t[:, :H].double(), W.double().view(H, 64, d)).reshape(B, d)) < 2e-6
    assert torch.equal(ops.head_cols(t.to(dev()), W.to(dev()), H)[0], out2)            # fixed summation order
    # wgrad: dW[64h+e, :] = sum_b x[b, 64h+e] rs[b, h] t[b, h, :]
    dW = ops.head_wgrad(x.to(dev()), t.to(dev()), H, row_scale=rs.to(dev()))
    ref = torch.einsum("bhe,bh,bhc->hec", x.double().view(B, H, 64), rs.double(), t[:, :H].double()).reshape(d, d)
    assert rel(dW, ref) < 2e-6
    assert rel(ops.head_wgrad(x.to(dev()), t.to(dev()), H), torch.einsum("bhe,bhc->hec", x.double().view(B, H, 64), t[:, :H].double()).reshape(d, d)) < 2e-6


@pytest.mark.parametrize("B,H,N", [(5, 12, 513), (2, 3, 17), (3, 16, 130)])
def test_cls_softmax_fwd_bwd(B, H, N):
    from xvit import ops
    s = torch.zeros(B, N, 16)
    s[:, :, :H] = randn(B, N, H, seed=1) * 4
    e, rz = ops.cls_softmax_fwd(s.to(dev()), H, 0.125)
    ref_e = torch.exp(0.125 * (s[:, :, :H] - s[:, :, :H].amax(dim=1, keepdim=True))).to(torch.bfloat16)
    assert H == 16 or float(e[:, :, H:].float().abs().max()) == 0.0
    assert rel(e[:, :, :H].float(), ref_e.float()) < 3e-3                       # v_exp vs libm, then one bf16 rounding
    assert rel(rz, 1.0 / e[:, :, :H].float().sum(dim=1)) < 1e-6                  # the sum of the ROUNDED weights
    dp = torch.zeros(B, N, 16)
    dp[:, :, :H] = randn(B, N, H, seed=2)
    coef, dsb = ops.cls_softmax_bwd(e, rz, dp.to(dev()), H, 0.125)
    p = e[:, :, :H].double().cpu() * rz.double().cpu()[:, None, :]
    ds = 0.125 * p * (dp[:, :, :H].double() - (p * dp[:, :, :H].double()).sum(dim=1, keepdim=True))
    assert rel(coef[:, :, :H], ds) < 1e-5 and rel(coef[:, :, H:], p) < 1e-6
    assert torch.equal(dsb[:, :, :H], coef[:, :, :H].to(torch.bfloat16)) and (H == 16 or float(dsb[:, :, H:].float().abs().max()) == 0.0)


@pytest.mark.parametrize("B,H,N", [(4, 12, 513), (3, 3, 17), (2, 4, 65)])
def test_low_rank_attention_equals_the_literal_order(B, H, N):
    """softmax(q (x Wk^T + bk)^T scale) (x Wv^T + bv) per head, computed without k and v (fp64 reference with them)."""
    from xvit import ops
    d = 64 * H
    scale = 0.125
    x = rt(randn(B, N, d, seed=1))
    q = randn(B, d, seed=2)
    Wk, Wv = randn(d, d, seed=3, scale=d ** -0.5), randn(d, d, seed=4, scale=d ** -0.5)
    bk, bv = randn(d, seed=5), randn(d, seed=6)
    xd = x.to(dev(), torch.bfloat16)
    R = torch.empty(2 * H, B, d, device=dev())
    Ub = torch.empty(B, 16, d, dtype=torch.bfloat16, device=dev())
    ops.head_rows(q.to(dev()), Wk.to(dev()), R[:H].transpose(0, 1), H, out_bf16=Ub)
    sc = torch.empty(B, N, 16, device=dev())
    ops.gemm(ops.NT, xd, Ub, sc)
    e, rz = ops.cls_softmax_fwd(sc, H, scale)
    S = torch.empty(B, 16, d, device=dev())
    ops.gemm(ops.TN, e, xd, S)
    o, _ = ops.head_cols(S, Wv.to(dev()), H, row_scale=rz, bias=bv.to(dev()))
    k = (x.double() @ Wk.double().T + bk.double()).view(B, N, H, 64)
    v = (x.double() @ Wv.double().T + bv.double()).view(B, N, H, 64)
    p = torch.softmax(torch.einsum("bhe,bnhe->bhn", q.double().view(B, H, 64), k) * scale, dim=-1)
    ref = torch.einsum("bhn,bnhe->bhe", p, v).reshape(B, d)
    assert rel(o, ref) < 3e-3, rel(o, ref)       # bf16 U and bf16 softmax weights; everything else fp32


@pytest.mark.parametrize("B,H,N", [(5, 12, 513), (2, 3, 17), (3, 16, 130)])
def test_cls_softmax_with_probability_dropout(B, H, N):
    """attn_drop on the probabilities (model_cross.py:97) inside the low-rank form: the kept weights, the three stat rows, and the backward
    with the regenerated mask — against fp64 with the very mask xvit_dropout draws on a contiguous [B, H, N] tensor (the one
    xvit_cls_xattn_fwd applies in the literal order)."""
    from xvit import ops
    pr, seed, scale = 0.25, 20240607, 0.125
    s = torch.zeros(B, N, 16)
    s[:, :, :H] = randn(B, N, H, seed=1) * 4
    e0, rz0 = ops.cls_softmax_fwd(s.to(dev()), H, scale)
    e, stat, ek = ops.cls_softmax_fwd(s.to(dev()), H, scale, dropout=(pr, seed))
    mask = (ops.dropout(torch.ones(B, H, N, device=dev()), pr, seed) != 0).permute(0, 2, 1)              # [B, N, H] keep flags
    assert 0.6 < float(mask.float().mean()) < 0.9
    assert torch.equal(e, e0) and torch.equal(stat[0], rz0)                                               # the undropped softmax is untouched
    assert torch.equal(ek[:, :, :H], torch.where(mask, e[:, :, :H], torch.zeros_like(e[:, :, :H])))
    assert H == 16 or float(ek[:, :, H:].float().abs().max()) == 0.0
    inv = 1.0 / (1.0 - pr)
    assert rel(stat[1], rz0 * inv) < 1e-6
    assert rel(stat[2], rz0 * inv * ek[:, :, :H].float().sum(dim=1)) < 1e-5
    # head_cols with the bias weighted by stat[2]; head_bias_grad is its transpose
    d = 64 * H
    t, W, bias = randn(B, 16, d, seed=3), randn(d, d, seed=2, scale=d ** -0.5), randn(d, seed=5)
    out, _ = ops.head_cols(t.to(dev()), W.to(dev()), H, row_scale=stat[1], bias=bias.to(dev()), bias_scale=stat[2])
    ref = (torch.einsum("bhc,hec->bhe", t[:, :H].double(), W.double().view(H, 64, d)) * stat[1].double().cpu()[:, :, None]
           + bias.double().view(1, H, 64) * stat[2].double().cpu()[:, :, None]).reshape(B, d)
    assert rel(out, ref) < 2e-6
    x = randn(B, d, seed=7)
    gb = ops.head_bias_grad(x.to(dev()), stat[2], H)
    assert rel(gb, (x.double().view(B, H, 64) * stat[2].double().cpu()[:, :, None]).sum(0).reshape(d)) < 2e-6
    # backward: dp is the gradient of the DROPPED probabilities
    dp = torch.zeros(B, N, 16)
    dp[:, :, :H] = randn(B, N, H, seed=2)
    coef, dsb = ops.cls_softmax_bwd(e, stat[0], dp.to(dev()), H, scale, dropout=(pr, seed))
    p = e[:, :, :H].double().cpu() * rz0.double().cpu()[:, None, :]
    m = mask.double().cpu() * inv
    dpt = m * dp[:, :, :H].double()
    ds = scale * p * (dpt - (p * dpt).sum(dim=1, keepdim=True))
    assert rel(coef[:, :, :H], ds) < 1e-5 and rel(coef[:, :, H:], p * m) < 1e-6
    assert torch.equal(dsb[:, :, :H], coef[:, :, :H].to(torch.bfloat16))
