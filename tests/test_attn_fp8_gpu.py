"""GPU: the MX-fp8 forward attention (xvit_attn_fwd_fp8; SURVEY.md 8 / BASELINE.json configs[4]) against the CPU oracle's
softmax attention (reference model_cross.py:55-59) and against the bf16 kernel.

Parity budget.  e4m3 carries 3 mantissa bits: an operand element is off by up to 6.25 % (3.6 % rms) of its block's scale, so
this path CANNOT meet the 1e-3 of the bf16 / fp32 paths.  What is stated and gated here, on N(0, 1) q / k / v (the scale of
LayerNorm-ed tokens through Xavier weights), rel-L2 of the output against the fp32 oracle on the same bf16 inputs:
  measured 5.0e-2 .. 5.6e-2 (N = 64 .. 4097)  -> gate 8e-2;   lse: measured 2.5e-2 .. 4.7e-2 absolute -> gate 8e-2
(q, k, v each carry 2.6e-2 of block-scaled e4m3 rounding, P another 3.6e-2).  The bf16 kernel sits at 2.1e-3 .. 2.3e-3 on the
same inputs (gate 3e-3).  The kernel is opt-in (XVIT_ATTN_FP8=1) and forward-only."""
import pytest
import torch

import ref_cpu as R
from _util import dev, rel

pytestmark = pytest.mark.gpu


def _qkv(B, N, H, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B * N, 3 * H * 64, generator=g).bfloat16()


@pytest.mark.parametrize("B,H,N", [(2, 3, 513), (1, 2, 4097), (2, 2, 130), (1, 1, 64), (3, 1, 65), (1, 2, 1000)])
def test_fp8_forward_vs_oracle(B, H, N):
    from xvit import ops
    d = H * 64
    qkv = _qkv(B, N, H, seed=N)
    o8, lse8 = ops.attn_fwd_fp8(qkv.to(dev()), B, N, H, 0.125)
    ob, lseb = ops.attn_fwd(qkv.to(dev()), B, N, H, 0.125)
    q, k, v = (R._split_heads(t.float().reshape(B, N, d), H) for t in qkv.split(d, dim=-1))
    o_ref, lse_ref = R.softmax_attention(q, k, v, 0.125)
    o_ref = R._merge_heads(o_ref).reshape(B * N, d)
    e8, eb = rel(o8, o_ref), rel(ob, o_ref)
    dl = float((lse8.cpu() - lse_ref).abs().max())
    print(f"\nB={B} H={H} N={N}: fp8 rel-L2 {e8:.3e} (bf16 kernel {eb:.3e}); max |lse - ref| {dl:.3e}")
    assert torch.isfinite(o8.float()).all() and torch.isfinite(lse8).all()
    assert e8 < 8e-2, f"fp8 attention output off by {e8:.3e}"
    assert dl < 8e-2, f"fp8 lse off by {dl:.3e}"
    assert eb < 3e-3


def test_fp8_handles_outliers_and_zero_blocks():
    """Block scales must follow the data: one 50x outlier row, an all-zero value block and tiny keys."""
    from xvit import ops
    B, H, N, d = 1, 2, 257, 128
    qkv = _qkv(B, N, H, seed=7).float()
    qkv[5, :64] *= 50.0                     # a query with a large norm: sharp softmax
    qkv[64:96, 2 * d:2 * d + 64] = 0.0      # 32 keys x head 0 of V exactly zero (scale byte 0)
    qkv[:, d + 64:2 * d] *= 1e-3            # head 1 keys tiny: near-uniform attention
    qkv = qkv.bfloat16()
    o8, lse8 = ops.attn_fwd_fp8(qkv.to(dev()), B, N, H, 0.125)
    q, k, v = (R._split_heads(t.float().reshape(B, N, d), H) for t in qkv.split(d, dim=-1))
    o_ref, lse_ref = R.softmax_attention(q, k, v, 0.125)
    o_ref = R._merge_heads(o_ref).reshape(B * N, d)
    assert torch.isfinite(o8.float()).all()
    assert rel(o8, o_ref) < 1.2e-1
    assert float((lse8.cpu() - lse_ref).abs().max()) < 0.35     # the outlier row's scores reach +-400: 3 mantissa bits of k


def test_fp8_workspace_is_checked():
    from xvit import _lib
    qkv = _qkv(1, 64, 1, seed=1).to(dev())
    o = torch.empty(64, 64, dtype=torch.bfloat16, device=dev())
    lse = torch.empty(64, device=dev())
    ws = torch.empty(256, dtype=torch.uint8, device=dev())
    p = qkv.data_ptr()
    rc = _lib.load().xvit_attn_fwd_fp8(p, p + 128, p + 256, 64 * 192, 192, o.data_ptr(), 64 * 64, 64, lse.data_ptr(), 1, 1, 64, 64, 0.125, ws.data_ptr(), 256, None)
    assert rc != 0 and b"workspace" in _lib.load().xvit_last_error_string()


def test_model_runs_with_fp8_attention_flag(monkeypatch):
    """XVIT_ATTN_FP8=1 switches the SelfAttentionBlocks' forward attention to the fp8 kernel: the model trains (finite loss and
    gradients; backward on the bf16 kernels) and its logits stay within the fp8 budget of the bf16 path."""
    import xvit
    cfg = R.make_config("small")
    sd = R.make_state_dict(cfg, seed=1)
    img, labels = R.make_inputs(cfg, 2, seed=1)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("XVIT_ATTN_FP8", flag)
        model = xvit.ModelCross(cfg).to(dev())
        model.load_state_dict(sd)
        model.train()
        logits, loss = model(img.to(dev()), labels.to(dev()))
        loss.backward()
        assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        out[flag] = logits.detach().clone()
    assert not torch.equal(out["0"], out["1"])                 # the flag did switch kernels
    assert rel(out["1"], out["0"]) < 1.5e-1
