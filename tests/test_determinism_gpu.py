"""GPU: deterministic-gradient mode (SURVEY.md 8(b) "Determinism": forward bit-reproducible; backward offers a deterministic
mode).  With xvit.ops.set_deterministic(True) the kernels that normally meet in fp32 atomics (LayerNorm dgamma / dbeta and the
bias column sums, xvit_colsum, the class-head wgrad, the GEMM epilogue's fused column sums) add per-block partial sums in a fixed
order: two backward passes give bit-identical .grad for EVERY parameter, and the values agree with the default (atomic) mode."""
import pytest
import torch

import ref_cpu as R
from _util import dev, randn, rel

pytestmark = pytest.mark.gpu


@pytest.fixture
def deterministic():
    from xvit import ops
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


def _grads(model, img, labels):
    for p in model.parameters():
        p.grad = None
    logits, loss = model(img, labels)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("name,batch", [("tiny", 6), ("small", 5)])
def test_two_backward_passes_are_bit_identical(deterministic, name, batch):
    import xvit
    cfg = R.make_config(name)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img, labels = R.make_inputs(cfg, batch, seed=1)
    img, labels = img.to(dev()), labels.to(dev())
    l1, g1 = _grads(model, img, labels)
    for _ in range(3):
        l2, g2 = _grads(model, img, labels)
        assert torch.equal(l1, l2)
        for k in g1:
            assert torch.equal(g1[k], g2[k]), k


def test_deterministic_mode_matches_default_mode():
    import xvit
    from xvit import ops
    cfg = R.make_config("small")
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img, labels = R.make_inputs(cfg, 4, seed=2)
    img, labels = img.to(dev()), labels.to(dev())
    _, g_atomic = _grads(model, img, labels)
    ops.set_deterministic(True)
    try:
        _, g_det = _grads(model, img, labels)
    finally:
        ops.set_deterministic(False)
    for k in g_atomic:
        # bias gradients taken from the GEMM epilogue sum fp32 values, the two-pass form the stored bf16 ones: up to half a bf16 ulp
        # (2^-9 = 2e-3) per value, and the fusion's single-token FFN sums only B = 4 of them — nothing averages; the rest 1e-5
        tol = 4e-3 if k.endswith("net.0.bias") else 1e-5
        assert rel(g_det[k], g_atomic[k]) < tol or float(g_atomic[k].abs().max()) < 1e-6, (k, rel(g_det[k], g_atomic[k]))


def test_kernel_level_fixed_order_reductions(deterministic):
    """colsum and LayerNorm backward with the partial-sum workspace: bit-identical across calls, equal to the float64 sums."""
    from xvit import ops
    rows, d = 5000, 768
    x = randn(rows, d, seed=3)
    xd = x.to(dev())
    a, b = ops.colsum(xd), ops.colsum(xd)
    assert torch.equal(a, b) and rel(a, x.double().sum(0)) < 1e-6
    acc = torch.full((d,), 2.0, device=dev())
    ops.colsum(xd, out=acc, accumulate=True)
    assert rel(acc - 2.0, x.double().sum(0)) < 1e-5
    g, dy = (1 + 0.1 * randn(d, seed=4)).to(dev()), randn(rows, d, seed=5).to(dev(), torch.bfloat16)
    _, mu, rs = ops.layernorm_fwd(xd, g, torch.zeros(d, device=dev()), 1e-5)
    outs = []
    for _ in range(2):
        dg, db, sx = (torch.zeros(d, device=dev()) for _ in range(3))
        dx, _ = ops.layernorm_bwd(dy, xd, mu, rs, g, dg, db, dxsum=sx)
        outs.append((dx, dg, db, sx))
    for t0, t1 in zip(*outs):
        assert torch.equal(t0, t1)
    xh = (x.double() - x.double().mean(-1, keepdim=True)) * torch.rsqrt(x.double().var(-1, unbiased=False, keepdim=True) + 1e-5)
    assert rel(outs[0][1], (dy.double().cpu() * xh).sum(0)) < 1e-5 and rel(outs[0][2], dy.double().cpu().sum(0)) < 1e-5
    assert rel(outs[0][3], outs[0][0].double().cpu().sum(0)) < 1e-5
