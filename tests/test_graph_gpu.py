"""GPU: the HIP-graph captured training step reproduces the eager step (same logits, loss, gradients) and can be
replayed with new inputs."""
import pytest
import torch

import ref_cpu as R
from _util import dev, rel

pytestmark = pytest.mark.gpu


def test_graphed_step_matches_eager_and_follows_new_inputs():
    import xvit
    from xvit.graph import GraphedStep
    cfg = R.make_config("tiny")
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img1, lab1 = R.make_inputs(cfg, 4, seed=0)
    img2, lab2 = R.make_inputs(cfg, 4, seed=9)
    img1, lab1, img2, lab2 = img1.to(dev()), lab1.to(dev()), img2.to(dev()), lab2.to(dev())

    def eager(img, lab):
        for p in model.parameters():
            p.grad = None
        logits, loss = model(img, lab)
        loss.backward()
        return logits.detach().clone(), float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}

    e1, e2 = eager(img1, lab1), eager(img2, lab2)
    step = GraphedStep(model, img1, lab1)
    for (img, lab), (el, eloss, eg) in (((img1, lab1), e1), ((img2, lab2), e2), ((img1, lab1), e1)):
        logits, loss = step(img, lab)
        torch.cuda.synchronize()
        assert torch.equal(logits, el) and float(loss.detach()) == eloss  # same kernels, same order: bit-identical
        for k, p in model.named_parameters():   # LN gamma/beta and bias grads are fp32 atomics across blocks: order-dependent last bits
            assert rel(p.grad, eg[k]) < 1e-5 or float(eg[k].abs().max()) < 1e-6, k
    # an optimizer step between replays is picked up (weights are re-cast inside the graph)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    opt.step()
    logits, loss = step(img1, lab1)
    torch.cuda.synchronize()
    assert float(loss) != e1[1]
    ref = eager(img1, lab1)
    assert rel(logits, ref[0]) < 1e-6 and abs(float(loss) - ref[1]) < 1e-6
    # the reference's loop drops the gradients between steps (zero_grad(set_to_none=True)): a replay hands the static buffers back
    opt.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in model.parameters())
    step(img1, lab1)
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        assert p.grad is not None and (rel(p.grad, ref[2][k]) < 1e-5 or float(ref[2][k].abs().max()) < 1e-6), k


def test_graphed_step_survives_a_stale_autograd_graph():
    """A kept `loss` pins the parameters' AccumulateGrad nodes to the default stream; a captured loss.backward() then crashed
    inside hipStreamEndCapture (bench.py hit it).  GraphedStep takes gradients with autograd.grad, which never runs those
    nodes: capture works with the stale graph alive and gives the eager gradients."""
    import xvit
    from xvit.graph import GraphedStep
    cfg = R.make_config("tiny")
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img, lab = R.make_inputs(cfg, 4, seed=0)
    img, lab = img.to(dev()), lab.to(dev())
    _, loss = model(img, lab)
    loss.backward()                      # `loss` (and with it the graph) stays alive on purpose
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    step = GraphedStep(model, img, lab)
    _, loss_g = step()
    torch.cuda.synchronize()
    assert float(loss_g) == float(loss.detach())
    for k, p in model.named_parameters():
        assert rel(p.grad, ref[k]) < 1e-5 or float(ref[k].abs().max()) < 1e-6, k


@pytest.mark.parametrize("name,batch", [("small", 3), ("base", 2)])
def test_graphed_step_replays_with_new_inputs_on_forked_fusions(name, batch):
    """Three modalities in a ring / the configs[1] shape, fusions forked on side streams inside the capture: a replay with NEW inputs must
    give that input's eager gradients.  (Round 3 found a latent reuse race here: a branch output read by another modality's fusion was
    freed by its owner stream's pool while that fusion's backward, a parallel graph branch, still read it — functional.keep.)"""
    import xvit
    from xvit.graph import GraphedStep
    cfg = R.make_config(name)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    ins = [tuple(t.to(dev()) for t in R.make_inputs(cfg, batch, seed=s)) for s in (0, 5)]

    def eager(img, lab):
        for p in model.parameters():
            p.grad = None
        logits, loss = model(img, lab)
        loss.backward()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}

    refs = [eager(*i) for i in ins]
    step = GraphedStep(model, *ins[0])
    for which in (0, 1, 1, 0):
        logits, _ = step(*ins[which])
        torch.cuda.synchronize()
        assert torch.equal(logits, refs[which][0])
        for k, p in model.named_parameters():
            ref = refs[which][1][k]
            assert rel(p.grad, ref) < 1e-5 or float(ref.abs().max()) < 1e-6, (which, k, rel(p.grad, ref))


def test_graphed_step_with_dropout_draws_new_masks_per_replay_and_matches_eager_at_the_same_epoch():
    """The reference trains with dropout 0.1 .. 0.25 (main_mist.py:71-77).  A captured step freezes the host-side seeds; the device-side
    epoch (xvit_set_dropout_epoch) makes each replay draw new masks.  Checked: (1) two replays of the same input differ; (2) a replay is
    exactly the eager step issued with the same host seeds and the same epoch value — logits, loss and every gradient; (3) eval mode and
    eager training afterwards are untouched (the registration is gone)."""
    import xvit
    import xvit.functional as XF
    from xvit import ops
    from xvit.graph import GraphedStep
    cfg = R.make_config("tiny", dropout=0.25)
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img, lab = R.make_inputs(cfg, 4, seed=0)
    img, lab = img.to(dev()), lab.to(dev())
    torch.manual_seed(123)
    calls0 = XF._DROP_CALLS
    step = GraphedStep(model, img, lab, warmup=2)
    per_step = (XF._DROP_CALLS - calls0) // 3            # 2 warm-up steps + the capture drew seeds
    assert per_step > 0 and ops._DROP_EPOCH is None
    capture_calls = XF._DROP_CALLS - per_step            # the call counter the captured step started from
    assert int(step._epoch) == 2                         # the capture itself executes nothing
    l1, loss1 = step()
    l1, loss1 = l1.clone(), float(loss1)
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    l2, loss2 = step()
    torch.cuda.synchronize()
    assert int(step._epoch) == 4 and not torch.equal(l1, l2) and float(loss2) != loss1       # (1) fresh masks
    # (2) the eager step with the capture's host seeds at epoch 3 (= the first replay)
    epoch = torch.full((1,), 3, dtype=torch.int64, device=dev())
    XF._DROP_CALLS = capture_calls
    ops.set_dropout_epoch(epoch)
    try:
        for p in model.parameters():
            p.grad = None
        le, losse = model(img, lab)
        losse.backward()
    finally:
        ops.set_dropout_epoch(None)
    assert torch.equal(le.detach(), l1) and float(losse.detach()) == loss1
    for k, p in model.named_parameters():
        assert rel(p.grad, g1[k]) < 1e-5 or float(g1[k].abs().max()) < 1e-6, k
    # the masks are real: the same seeds WITHOUT the epoch give another result
    XF._DROP_CALLS = capture_calls
    l0, _ = model(img, lab)
    assert not torch.equal(l0.detach(), l1)
    # (3) eval mode has no dropout: deterministic, equal to the p = 0 model
    model.eval()
    a, _ = model(img, lab)
    b, _ = model(img, lab)
    assert torch.equal(a, b)
