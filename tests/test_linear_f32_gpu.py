"""GPU parity: the fp32 Linear of the single-token CLS path (xvit_linear_f32, f32-input MFMA) against float64 matmul.
All-fp32 operands: the gate is fp32 round-off, not a bf16 budget."""
import math

import pytest
import torch

from _util import dev, randn, rel

pytestmark = pytest.mark.gpu


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


@pytest.mark.parametrize("M,N,K", [(126, 768, 768), (126, 3072, 768), (126, 768, 3072), (2, 192, 192), (33, 2, 3072), (4, 768, 192), (257, 96, 64), (1, 768, 768)])
def test_linear_f32_plain_bias_residual(M, N, K):
    from xvit import ops
    x, w, b, r = randn(M, K, seed=1), randn(N, K, seed=2, scale=K ** -0.5), randn(N, seed=3), randn(M, N, seed=4)
    ref = x.double() @ w.double().T
    y, yb, zb = ops.linear_f32(x.to(dev()), w.to(dev()))
    assert yb is None and zb is None and rel(y, ref) < 2e-6, rel(y, ref)
    y, yb, _ = ops.linear_f32(x.to(dev()), w.to(dev()), b.to(dev()), residual=r.to(dev()), want_bf16=True)
    ref2 = ref + b.double() + r.double()
    assert rel(y, ref2) < 2e-6
    assert torch.equal(yb, y.to(torch.bfloat16))                  # the bf16 copy is the rounding of the stored fp32 value
    # bit-reproducible: the K-chunks are summed in a fixed order
    y2, _, _ = ops.linear_f32(x.to(dev()), w.to(dev()), b.to(dev()), residual=r.to(dev()))
    assert torch.equal(y, y2)


def test_linear_f32_gelu_and_strided_rows():
    from xvit import ops
    B, N_tok, d, f = 6, 17, 192, 768
    tokens = randn(B, N_tok * d, seed=5)                          # the CLS rows of a [B, N, d] tensor: row stride N*d
    w, b = randn(f, d, seed=6, scale=d ** -0.5), randn(f, seed=7, scale=0.1)
    t = tokens.to(dev())
    a, ab, zb = ops.linear_f32(t[:, :d], w.to(dev()), b.to(dev()), act=ops.ACT_GELU, want_z=True, want_bf16=True)
    z = tokens[:, :d].double() @ w.double().T + b.double()
    assert rel(a, _gelu(z)) < 3e-6                                # erf by a 1.5e-7-accurate rational form
    assert rel(zb.float(), z) < 3e-3 and torch.equal(ab, a.to(torch.bfloat16))


def test_linear_f32_dropout_mask_matches_xvit_dropout():
    from xvit import ops
    M, N, K, p, seed = 8, 192, 192, 0.25, 1234567
    x, w = randn(M, K, seed=1), randn(N, K, seed=2, scale=K ** -0.5)
    y, _, _ = ops.linear_f32(x.to(dev()), w.to(dev()), dropout=(p, seed))
    plain, _, _ = ops.linear_f32(x.to(dev()), w.to(dev()))
    assert torch.equal(y, ops.dropout(plain.contiguous(), p, seed))


def test_linear_f32_any_width_and_offset_views():
    """The reference accepts any hidden_dim / mlp_dim and any view; the kernel needs K % 16 == 0 and 16-byte-aligned rows, so
    ops.linear_f32 pads odd widths / misaligned views into aligned scratch copies (exact).  The raw C entry point still
    refuses them loudly."""
    from xvit import _lib, ops
    for M, N, K in ((4, 8, 24), (5, 100, 200), (3, 7, 50)):
        x, w, b = randn(M, K, seed=1), randn(N, K, seed=2, scale=K ** -0.5), randn(N, seed=3)
        y, _, _ = ops.linear_f32(x.to(dev()), w.to(dev()), b.to(dev()))
        assert rel(y, x.double() @ w.double().T + b.double()) < 2e-6
    big = randn(9, 4 * 64 + 1, seed=4).to(dev())
    xv = big[:, 1:1 + 64]                                           # rows start 4 bytes off a 16-byte boundary
    w = randn(32, 64, seed=5).to(dev())
    y, _, _ = ops.linear_f32(xv, w)
    assert rel(y, xv.double().cpu() @ w.double().cpu().T) < 2e-6
    x24, w24, y24 = torch.zeros(4, 24, device=dev()), torch.zeros(8, 24, device=dev()), torch.zeros(4, 8, device=dev())
    rc = _lib.load().xvit_linear_f32(x24.data_ptr(), 24, w24.data_ptr(), 24, None, y24.data_ptr(), 8, 4, 8, 24, 0, None, 0, None, 0, None, 0, 0.0, 0, None, 0,
                                     torch.cuda.current_stream().cuda_stream)
    assert rc < 0 and b"multiple of 16" in _lib.load().xvit_last_error_string()


def test_layernorm_and_cls_attention_fp32_io():
    """fp32 outputs of LayerNorm and fp32 query / output of the CLS attention: same values as the bf16 path before rounding."""
    import ref_cpu as R
    from xvit import ops
    rows, d = 10, 192
    x, g, b = randn(rows, d, seed=1), randn(d, seed=2).abs() + 0.5, randn(d, seed=3)
    yf, yb, mu, rs = ops.layernorm_fwd_f32(x.to(dev()), g.to(dev()), b.to(dev()), 1e-5)
    ref = R.layer_norm(x.double(), g.double(), b.double())
    assert rel(yf, ref) < 2e-6 and torch.equal(yb, yf.to(torch.bfloat16))
    B, N, H, dd = 3, 65, 3, 192
    q = randn(B, dd, seed=4)
    kv = randn(B * N, 2 * dd, seed=5).to(torch.bfloat16)
    o_b, p_b, o_f = ops.cls_xattn_fwd(q.to(dev()), kv.to(dev()), B, N, H, (dd // H) ** -0.5, want_f32=True)
    k, v = (R._split_heads(t.float().reshape(B, N, dd), H) for t in kv.split(dd, dim=-1))
    o_ref, _ = R.softmax_attention(R._split_heads(q.reshape(B, 1, dd), H).double(), k.double(), v.double(), (dd // H) ** -0.5)
    assert rel(o_f, R._merge_heads(o_ref).reshape(B, dd)) < 5e-6
    assert torch.equal(o_b, o_f.to(torch.bfloat16))
