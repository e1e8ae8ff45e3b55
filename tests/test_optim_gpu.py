"""GPU: xvit.optim.FusedAdam against torch.optim.Adam (the reference's optimizer, model_cross.py:277) on the same
parameters and gradients, including L2 weight decay, several steps, odd sizes, and the bf16 operand copy it maintains."""
import pytest
import torch

import ref_cpu as R
from _util import dev, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 0.05])
def test_fused_adam_matches_torch_adam(wd):
    from xvit.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(768, 768), (3072,), (5,), (2, 3072), (1, 513, 768), (16385,), (1,)]
    ps_a = [torch.nn.Parameter(torch.randn(*s, device=dev())) for s in shapes]
    ps_b = [torch.nn.Parameter(p.detach().clone()) for p in ps_a]
    a = FusedAdam(ps_a, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
    b = torch.optim.Adam(ps_b, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
    for step in range(5):
        for pa, pb in zip(ps_a, ps_b):
            g = torch.randn_like(pa)
            pa.grad, pb.grad = g.clone(), g.clone()
        if step == 3:
            ps_a[2].grad = None; ps_b[2].grad = None      # a parameter without gradient is skipped, like torch does
        a.step(); b.step()
    for pa, pb in zip(ps_a, ps_b):
        assert rel(pa, pb) < 2e-6, rel(pa, pb)
    for pa, pb in zip(ps_a, ps_b):
        assert rel(a.state[pa]["exp_avg"], b.state[pb]["exp_avg"]) < 5e-6
        assert rel(a.state[pa]["exp_avg_sq"], b.state[pb]["exp_avg_sq"]) < 5e-6


def test_fused_adam_trains_the_model_and_keeps_operand_copies_fresh():
    """Three optimizer steps on ModelCross (tiny): same trajectory as torch.optim.Adam within bf16 noise, and the flat
    bf16 weight copy equals bf16(weights) after each step without any cast launch."""
    import xvit
    import xvit.functional as XF
    from xvit.optim import FusedAdam
    cfg = R.make_config("tiny")
    sd = R.make_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, 4, seed=0)
    img, labels = img.to(dev()), labels.to(dev())
    losses = {}
    for kind in ("fused", "torch"):
        model = xvit.ModelCross(cfg).to(dev())
        model.load_state_dict(sd)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-3) if kind == "fused" else torch.optim.Adam(model.parameters(), lr=1e-3)
        hist = []
        for _ in range(3):
            opt.zero_grad()
            _, loss = model(img, labels)
            loss.backward()
            opt.step()
            hist.append(float(loss.detach()))
            if kind == "fused":
                grp = model._flat
                casts_before = XF.SHADOWS.casts
                for p, v16 in zip(grp.params, grp.view16):
                    assert torch.equal(v16, p.detach().to(torch.bfloat16))
                assert grp.stamp == sum(q._version for q in grp.params) and XF.SHADOWS.casts == casts_before
        losses[kind] = hist
    assert losses["fused"][0] == pytest.approx(losses["torch"][0], abs=1e-6)
    assert losses["fused"][2] < losses["fused"][0]                       # it learns
    assert abs(losses["fused"][2] - losses["torch"][2]) < 2e-2           # same trajectory up to bf16 rounding flips


def test_fused_adam_refreshes_per_parameter_operand_copies():
    """Models without a flat weight group (ModelVIT here, also the model.py Encoder, stand-alone modules, XVIT_FLAT_WEIGHTS=0)
    cache one bf16 copy per weight keyed by the parameter's version counter.  FusedAdam writes parameters through raw
    pointers (no version bump), so it must drop those copies itself — otherwise the GEMMs train against frozen weights while
    the fp32 masters move."""
    import xvit
    import xvit.functional as XF
    from xvit.optim import FusedAdam
    cfg = R.make_config("tiny", num_layers=2)
    sd = R.make_vit_state_dict(cfg, seed=0)
    img, labels = R.make_inputs(cfg, 4, seed=0)
    img, labels = img.to(dev()), labels.to(dev())
    losses = {}
    for kind in ("fused", "torch"):
        model = xvit.ModelVIT(cfg).to(dev())
        model.load_state_dict(sd)
        model.train()
        opt = FusedAdam(model.parameters(), lr=2e-3) if kind == "fused" else torch.optim.Adam(model.parameters(), lr=2e-3)
        hist = []
        for _ in range(4):
            opt.zero_grad()
            _, loss = model(img, labels)
            loss.backward()
            opt.step()
            hist.append(float(loss.detach()))
            if kind == "fused":    # the copy the NEXT forward will use is bf16(current weight)
                for p in model.parameters():
                    if p.dim() == 2 and p.shape[0] > 2:
                        assert torch.equal(XF.SHADOWS.get(p), p.detach().to(torch.bfloat16))
        losses[kind] = hist
    assert losses["fused"][0] == pytest.approx(losses["torch"][0], abs=1e-6)
    assert losses["fused"][3] < losses["fused"][0]
    for a, b in zip(losses["fused"], losses["torch"]):
        assert abs(a - b) < 2e-2, (losses["fused"], losses["torch"])


def test_discarded_model_releases_its_flat_weight_buffers():
    """main_mist.py builds a new model per seed x arm x fold in one process: a dropped model must take its flat fp32 / bf16
    weight buffers with it (the registry holds them weakly)."""
    import gc
    import xvit
    import xvit.functional as XF
    cfg = R.make_config("tiny")
    img, labels = R.make_inputs(cfg, 2, seed=0)
    img, labels = img.to(dev()), labels.to(dev())
    gc.collect(); torch.cuda.synchronize()
    n_groups, mem0 = len(XF.SHADOWS.groups), torch.cuda.memory_allocated()
    for _ in range(3):
        model = xvit.ModelCross(cfg).to(dev())
        _, loss = model(img, labels)
        loss.backward()
        assert len(XF.SHADOWS.groups) == n_groups + 1
        del model, loss, _
        gc.collect()
        assert len(XF.SHADOWS.groups) == n_groups
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= mem0 + (1 << 20)
