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


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_graphed_step_of_model_vit(p):
    """The reference's second model (modelv3.ModelVIT, main_mist.py:87 with params_list2: dropout 0.1) as a captured step: without dropout a
    replay reproduces the eager gradients; with the reference's rate the replays draw different masks and every gradient stays finite."""
    import xvit
    from xvit.graph import GraphedStep
    cfg = R.make_config("small", num_layers=2, dropout=p)
    sd = R.make_vit_state_dict(cfg, seed=11)
    model = xvit.ModelVIT(cfg).to(dev())
    model.load_state_dict(sd, strict=True)
    model.train()
    ins = [tuple(t.to(dev()) for t in R.make_inputs(cfg, 3, seed=s)) for s in (4, 6)]
    torch.manual_seed(7)
    step = GraphedStep(model, *ins[0])
    if p == 0.0:
        for which in (0, 1, 0):
            for q in model.parameters():
                q.grad = None
            logits_e, loss_e = model(*ins[which])
            loss_e.backward()
            ref = {k: q.grad.clone() for k, q in model.named_parameters()}
            logits, loss = step(*ins[which])
            torch.cuda.synchronize()
            assert torch.equal(logits, logits_e.detach()) and float(loss) == float(loss_e.detach())
            for k, q in model.named_parameters():
                assert rel(q.grad, ref[k]) < 1e-5 or float(ref[k].abs().max()) < 1e-6, (which, k)
    else:
        l1 = step(*ins[0])[0].clone()
        l2 = step(*ins[0])[0].clone()
        torch.cuda.synchronize()
        assert not torch.equal(l1, l2)
        assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in model.parameters())


def test_capture_of_a_model_that_opens_no_zero_arena():
    """The backward chains accumulate bias / LayerNorm gradients into zeroed vectors.  ModelCross and ModelVIT open a zero arena per step
    (functional.arena_begin, one fill); any other composition of the module classes — here a bare nn.Sequential of SelfAttentionBlocks under
    a loss — does not, and a capture must then zero per request INSIDE the graph instead of slicing an arena an earlier eager forward
    filled once (replays would accumulate into it: the second replay's LayerNorm gradients came out doubled)."""
    import xvit
    cfg = R.make_config("small")
    # leave a stale eager arena behind, as an earlier ModelCross step in the same process does
    warm = xvit.ModelCross(cfg).to(dev())
    warm(*[t.to(dev()) for t in R.make_inputs(cfg, 2, seed=1)])[1].backward()
    torch.manual_seed(3)
    blocks = torch.nn.Sequential(xvit.SelfAttentionBlock(cfg), xvit.SelfAttentionBlock(cfg)).to(dev())
    x = torch.randn(2, 65, cfg.hidden_dim, device=dev())

    def run():
        for q in blocks.parameters():
            q.grad = None
        blocks(x).square().mean().backward()

    run()
    torch.cuda.synchronize()
    ref = {k: q.grad.clone() for k, q in blocks.named_parameters()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    for q in blocks.parameters():
        q.grad = None
    with torch.cuda.graph(g):
        blocks(x).square().mean().backward()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for k, q in blocks.named_parameters():
        assert rel(q.grad, ref[k]) < 1e-5 or float(ref[k].abs().max()) < 1e-6, k


def test_two_captured_models_and_eager_steps_do_not_share_state():
    """Process-wide pieces (zero arena, capture keep-list, flat weight buffers, stream pools) must not leak between two captured steps of
    different models, nor into eager steps run afterwards: replay A, B, A, B, then eager A and B — every gradient as in the eager reference."""
    import xvit
    from xvit.graph import GraphedStep
    cfg_a, cfg_b = R.make_config("tiny"), R.make_config("small", num_layers=2)
    a = xvit.ModelCross(cfg_a).to(dev())
    a.load_state_dict(R.make_state_dict(cfg_a, seed=0))
    b = xvit.ModelVIT(cfg_b).to(dev())
    b.load_state_dict(R.make_vit_state_dict(cfg_b, seed=11))
    a.train(); b.train()
    in_a = tuple(t.to(dev()) for t in R.make_inputs(cfg_a, 4, seed=0))
    in_b = tuple(t.to(dev()) for t in R.make_inputs(cfg_b, 3, seed=4))

    def eager(m, ins):
        for q in m.parameters():
            q.grad = None
        logits, loss = m(*ins)
        loss.backward()
        return logits.detach().clone(), {k: q.grad.clone() for k, q in m.named_parameters()}

    ref_a, ref_b = eager(a, in_a), eager(b, in_b)
    step_a = GraphedStep(a, *in_a)
    step_b = GraphedStep(b, *in_b)

    def check(m, ref, logits):
        torch.cuda.synchronize()
        assert torch.equal(logits, ref[0])
        for k, q in m.named_parameters():
            assert rel(q.grad, ref[1][k]) < 1e-5 or float(ref[1][k].abs().max()) < 1e-6, k

    for _ in range(2):
        check(a, ref_a, step_a(*in_a)[0])
        check(b, ref_b, step_b(*in_b)[0])
    la, ga = eager(a, in_a)
    check(a, ref_a, la)
    lb, gb = eager(b, in_b)
    check(b, ref_b, lb)
    check(a, ref_a, step_a(*in_a)[0])
