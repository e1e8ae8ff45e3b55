"""GPU parity of the fused patch embedding (xvit_patch_embed_fwd / _wgrad: the GEMM's loaders gather the patch rows from the
volume, reference model_cross.py:193-197) against (a) the stored-patch-matrix path (xvit_patchify + xvit_gemm: same kernel
arithmetic, so the forward must match BIT FOR BIT) and (b) the CPU oracle's patchify + linear."""
import os

import pytest
import torch

import ref_cpu as R
from _util import dev, rel

pytestmark = pytest.mark.gpu

# (B, M, (D, H, W), patch, d)
GEOMS = [
    pytest.param(2, 2, (128, 128, 128), (16, 16, 16), 768, id="configs1-16cube"),      # configs[1]: 512 patches of 16^3
    pytest.param(1, 1, (128, 128, 128), (8, 8, 8), 256, id="configs4-8cube"),          # configs[4]: 4096 patches of 8^3
    pytest.param(5, 1, (64, 64, 64), (8, 8, 8), 256, id="64cube-8cube"),
    pytest.param(3, 2, (128, 64, 128), (16, 8, 16), 512, id="non-cubic"),
    pytest.param(6, 3, (128, 32, 256), (16, 16, 32), 256, id="wide-runs"),
    # patch grids where 64 consecutive tokens are NOT whole d-columns of one h-row: the weight gradient places its k-rows per K-step
    pytest.param(1, 2, (240, 240, 240), (16, 16, 16), 768, id="configs2-240cube-15-per-axis"),   # configs[2]: 3375 patches (dataset_ucsf.py:84-88)
    pytest.param(20, 1, (80, 48, 112), (16, 16, 16), 256, id="odd-grid-5x3x7"),                  # 105 patches per sample: K-steps straddle samples
    pytest.param(2, 3, (96, 160, 48), (8, 16, 16), 512, id="odd-grid-12x10x3"),
]


def _inputs(B, M, vol, patch, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, M, 1, *vol, generator=g).bfloat16()
    pd = patch[0] * patch[1] * patch[2]
    P = (vol[0] // patch[0]) * (vol[1] // patch[1]) * (vol[2] // patch[2])
    w = (torch.randn(d, pd, generator=g) / pd ** 0.5).bfloat16()
    bias = torch.randn(d, generator=g)
    pos = torch.randn(1 + P, d, generator=g)
    return img, w, bias, pos, P, pd


@pytest.mark.parametrize("B,M,vol,patch,d", GEOMS)
def test_fused_forward_bit_exact_and_vs_oracle(B, M, vol, patch, d):
    from xvit import ops
    img, w, bias, pos, P, pd = _inputs(B, M, vol, patch, d)
    gi, gw, gb, gp = img.to(dev()), w.to(dev()), bias.to(dev()), pos.to(dev())
    assert ops.patch_embed_supported(gi, patch, d)
    x = ops.patch_embed_fwd(gi, patch, gw, gb, gp)
    # the stored-patch-matrix path
    patches = ops.patchify(gi, patch, pad_cls_row=True).reshape(-1, pd)
    x_ref = torch.empty_like(x)
    ops.set_option("gemm_tile", 2)       # the same 256x256 kernel (and K order) whatever the grid size
    try:
        ops.gemm(ops.NT, patches, gw, x_ref, bias=gb, residual=gp, res_row_mod=1 + P, res_row_off=0)
    finally:
        ops.set_option("gemm_tile", 0)
    assert torch.equal(x, x_ref), f"fused forward differs from patchify + GEMM: rel {rel(x, x_ref):.3e}"
    # the oracle (fp32 accumulate of the same bf16 operands), patch rows only: CLS rows are overwritten by xvit_cls_row_fwd
    xo = x.reshape(M, B, 1 + P, d)[:, :, 1:].cpu()
    for m in range(M):
        ref = R.patchify(img[:, m, 0].float(), patch) @ w.float().T + bias + pos[1:]
        assert rel(xo[m], ref) < 1e-5, f"modality {m}: {rel(xo[m], ref):.3e}"


@pytest.mark.parametrize("B,M,vol,patch,d", GEOMS)
def test_fused_wgrad(B, M, vol, patch, d):
    from xvit import ops
    img, w, bias, pos, P, pd = _inputs(B, M, vol, patch, d, seed=1)
    gi = img.to(dev())
    rows = M * B * (1 + P)
    dx = torch.randn(rows, d, generator=torch.Generator().manual_seed(2)).bfloat16().to(dev())
    dW = ops.patch_embed_wgrad(gi, patch, dx)
    patches = ops.patchify(gi, patch, pad_cls_row=True).reshape(-1, pd)          # zero CLS rows
    ref = (dx.double().T @ patches.double()).float()
    assert rel(dW, ref) < 2e-6, f"fused wgrad vs fp64 reference: {rel(dW, ref):.3e}"
    dW2 = ops.patch_embed_wgrad(gi, patch, dx)
    assert torch.equal(dW, dW2), "fused wgrad is not reproducible"


def test_unsupported_inputs_take_the_patchify_path():
    from xvit import _lib, ops
    import ctypes as C
    img = torch.randn(2, 2, 1, 128, 128, 128).bfloat16().to(dev())
    assert ops.patch_embed_supported(img, (16, 16, 16), 768)
    assert not ops.patch_embed_supported(img.float(), (16, 16, 16), 768)               # fp32 volumes are converted by patchify
    assert not ops.patch_embed_supported(img, (16, 16, 16), 192)                       # d not a multiple of 256
    assert not ops.patch_embed_supported(img[:1, :1], (16, 16, 16), 768)               # 513 rows: small-tile kernels
    assert not ops.patch_embed_supported(img, (32, 32, 4), 768)                        # 8-byte runs
    small = torch.randn(8, 1, 1, 48, 48, 48).bfloat16().to(dev())
    assert not ops.patch_embed_supported(small, (16, 16, 16), 768)                     # 3 patches per axis
    # the C entry point refuses what the predicate refuses (no launch)
    g = ops._patch_geom(small, (16, 16, 16), 1)
    x = torch.empty(8 * 28, 768, device=dev())
    wb = torch.empty(768, 4096, dtype=torch.bfloat16, device=dev())
    rc = _lib.load().xvit_patch_embed_fwd(small.data_ptr(), C.byref(g), wb.data_ptr(), 4096, None, None, 0, x.data_ptr(), 768, 768, None)
    assert rc != 0 and b"not supported" in _lib.load().xvit_last_error_string()


def test_model_cross_fused_equals_unfused(monkeypatch):
    """ModelCross on a bf16 volume: fused patch embedding vs XVIT_PATCH_EMBED=unfused — same logits bit for bit, same
    patch-embedding weight gradient up to fp32 summation order."""
    import xvit
    cfg = R.make_config("base")
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, 4, seed=0)
    img = img.bfloat16().to(dev())
    labels = labels.to(dev())
    out = {}
    for mode in ("fused", "unfused"):
        monkeypatch.setenv("XVIT_PATCH_EMBED", mode)
        model = xvit.ModelCross(cfg).to(dev())
        model.load_state_dict(sd)
        model.train()
        logits, loss = model(img, labels)
        loss.backward()
        out[mode] = (logits.detach().clone(), model.patch_to_embedding.weight.grad.clone(), model.pos_embedding.grad.clone())
    assert torch.equal(out["fused"][0], out["unfused"][0])
    assert torch.equal(out["fused"][2], out["unfused"][2])
    assert rel(out["fused"][1], out["unfused"][1]) < 1e-5
