"""GPU: size-independent properties of the path at BASELINE.json's full configs[1] shape
(2 modalities, 128^3 volumes, 16^3 patches -> N = 513, d = 768, 12 heads, 2x2 blocks), where running
the CPU oracle for every case would take minutes.  Each property holds for the reference by
construction (model_cross.py:186-212 has no cross-sample operation and LayerNorm, not BatchNorm)."""
import pytest
import torch

import ref_cpu as R
from _util import dev, randn, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def base_model():
    import xvit
    cfg = R.make_config("base")
    torch.manual_seed(0)
    model = xvit.ModelCross(cfg).to(dev())
    model.eval()
    return cfg, model


def _inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, cfg.num_modalities, 1, *cfg.img_size, generator=g).to(dev(), torch.bfloat16)
    labels = torch.randint(0, 2, (B,), generator=g).to(dev())
    return img, labels


def test_samples_are_independent_and_batch_order_is_irrelevant(base_model):
    """logits[b] depends on sample b only: permuting the batch permutes the logits, and a sample run alone
    gives the logits it gets inside a batch.  Every per-token reduction (GEMM K loop, softmax over keys, LN
    over d) has a fixed order regardless of which tile a row lands in, so this holds to fp32 round-off of the
    (differently split) 1-token GEMMs."""
    cfg, model = base_model
    B = 6
    img, labels = _inputs(cfg, B, seed=3)
    with torch.no_grad():
        full, _ = model(img, labels)
        perm = torch.tensor([4, 0, 5, 2, 1, 3], device=dev())
        permuted, _ = model(img[perm].contiguous(), labels[perm])
        alone, _ = model(img[2:3].contiguous(), labels[2:3])
    scale = float(full.abs().max())
    assert float((permuted - full[perm]).abs().max()) <= 2e-3 * scale + 1e-5
    assert float((alone[0] - full[2]).abs().max()) <= 2e-3 * scale + 1e-5


def test_loss_is_the_batch_mean_and_gradients_add_over_samples(base_model):
    """CE is a mean over the batch (model_cross.py:211): grad(batch of 4) == mean of the 4 single-sample grads."""
    cfg, model = base_model
    img, labels = _inputs(cfg, 4, seed=5)
    names = ["patch_to_embedding.weight", "transformer.1.blocks.0.1.ffn.fn.net.3.weight", "transformer.0.fusion.1.attn.fn.wv.weight", "mlp_head.1.0.bias"]
    params = dict(model.named_parameters())
    model.zero_grad()
    _, loss = model(img, labels)
    loss.backward()
    whole = {n: params[n].grad.clone() for n in names}
    model.zero_grad()
    losses = []
    for b in range(4):
        _, lb = model(img[b:b + 1].contiguous(), labels[b:b + 1])
        (lb / 4).backward()          # accumulates into .grad like autograd does
        losses.append(float(lb))
    assert abs(float(loss) - sum(losses) / 4) < 2e-3
    for n in names:
        assert rel(params[n].grad, whole[n]) < 2e-2, (n, rel(params[n].grad, whole[n]))


def test_attention_rows_are_convex_combinations_at_full_size():
    """softmax rows sum to one: with V == const every output equals that constant, for every (b, h, query),
    at B*H = 504 heads x N = 513 — exercises every tile/tail path of the fused kernel at the bench shape."""
    from xvit import ops
    B, H, N, d = 42, 12, 513, 768
    qkv = (torch.randn(B * N, 3 * d, device=dev()) * 2).bfloat16()
    qkv[:, 2 * d:] = 0.75
    o, lse = ops.attn_fwd(qkv, B, N, H, 0.125)
    assert float((o.float() - 0.75).abs().max()) <= 0.75 * 2 ** -8
    assert torch.isfinite(lse).all()
    # and d(loss)/dq = d(loss)/dk = 0 when V is constant (dP = do . v is the same for every key)
    do = torch.randn(B * N, d, device=dev()).bfloat16()
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, 0.125).float()
    scale = float(dqkv[:, 2 * d:].abs().max())
    assert float(dqkv[:, :2 * d].abs().max()) <= 2e-2 * scale


def test_patchify_is_a_bijection_at_full_size():
    """Every voxel lands in exactly one (token, feature) slot: multiset of values preserved per volume."""
    from xvit import ops
    img = torch.randn(4, 2, 1, 128, 128, 128, device=dev()).bfloat16()
    pt = ops.patchify(img, (16, 16, 16))                                   # [M, B*P, pd]
    for m in range(2):
        for b in range(4):
            a = img[b, m, 0].reshape(-1).float().sort().values
            p = pt[m, b * 512:(b + 1) * 512].reshape(-1).float().sort().values
            assert torch.equal(a, p)
    # spot-check the index map against the closed form (model_cross.py:193): t = (h*8 + w)*8 + d, f = (p1*16 + p2)*16 + p3
    for (b, m, dd, hh, ww) in [(0, 0, 0, 0, 0), (3, 1, 127, 64, 33), (1, 0, 17, 127, 5)]:
        t = ((hh // 16) * 8 + ww // 16) * 8 + dd // 16
        f = ((dd % 16) * 16 + hh % 16) * 16 + ww % 16
        assert pt[m, b * 512 + t, f] == img[b, m, 0, dd, hh, ww]


def test_layernorm_output_statistics_at_full_size():
    from xvit import ops
    rows, d = 42 * 513, 768
    x = torch.randn(rows, d, device=dev()) * 3 + 1.5
    y, mean, rstd = ops.layernorm_fwd(x, torch.ones(d, device=dev()), torch.zeros(d, device=dev()), 1e-5)
    yf = y.float()
    assert float(yf.mean(dim=1).abs().max()) < 5e-3 and float((yf.var(dim=1, unbiased=False) - 1).abs().max()) < 2e-2


def test_workgroups_are_dealt_round_robin_to_the_xcds_and_cu_masks_restrict_streams():
    """The tile / block remaps in gemm.hip and attention.hip assume that the workgroups of a launch are dealt to the 8 XCDs round-robin in
    dispatch order (each XCD with its own L2).  xvit_cu_trace records where every workgroup actually ran; a CU-masked stream
    (xvit/cu_mask.py) must confine a launch to its share of the CUs on every XCD."""
    from xvit import _lib, cu_mask
    n = 2048
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev())
    st = torch.cuda.current_stream()
    _lib.check(_lib.load().xvit_cu_trace(out.data_ptr(), n, 20, st.cuda_stream), "xvit_cu_trace")
    torch.cuda.synchronize()
    xcc = (out.view(n, 2)[:, 0] & 0xF).cpu()
    n_xcd = int(xcc.max()) + 1
    assert n_xcd == 8
    # dispatch order -> XCD is round-robin; the starting XCD carries over from the previous launch, so what holds (and
    # all the remaps need) is that workgroups i and j share an XCD exactly when i == j (mod 8)
    assert torch.equal((xcc[1:256] - xcc[:255]) % n_xcd, torch.ones(255, dtype=xcc.dtype))

    def places(stream):
        o = torch.zeros(2 * n, dtype=torch.int32, device=dev())
        with torch.cuda.stream(stream):
            _lib.check(_lib.load().xvit_cu_trace(o.data_ptr(), n, 30, stream.cuda_stream), "xvit_cu_trace")
        stream.synchronize()
        o = o.view(n, 2).cpu()
        return {(int(x) & 0xF, (int(h) >> 8) & 0xFF) for x, h in o.tolist()}     # (XCD, se/sh/cu bits of HW_ID)

    n_cu = torch.cuda.get_device_properties(dev()).multi_processor_count
    assert len(places(torch.cuda.Stream(dev()))) == n_cu
    halves = [places(cu_mask.masked_stream(dev(), bits)) for bits in cu_mask.split_masks(n_cu, 2)]
    assert len(halves[0]) == n_cu // 2 and len(halves[1]) == n_cu // 2 and not (halves[0] & halves[1])
    assert {x for x, _ in halves[0]} == set(range(n_xcd))            # a contiguous bit range = the same CU slice on every XCD
