"""GPU: the bucketed gradient reducer (xvit/ddp.py) under the REAL model and its multi-stream backward.

ModelCross runs its modality branches and fusions on side streams; autograd replays backward on them, so the gradients
of one bucket are written on several streams.  The reducer must order its pack + all-reduce (on its own comm stream) behind
ALL of them.  (a) 1-rank RCCL group on this box: with tiny buckets that each span both branches, the reduced gradients
must equal the gradients of the same step without a reducer.  (b) where >= 2 GPUs are visible: two RCCL ranks in fresh
processes, reduced gradients == single-process global-batch gradients, parameters bit-identical across ranks after 3
Adam steps."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import ref_cpu as R
from _util import dev, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grads(model, img, labels, reducer=None):
    for p in model.parameters():
        p.grad = None
    _, loss = model(img, labels)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def test_reducer_orders_behind_every_branch_stream_one_rank_rccl(monkeypatch):
    import torch.distributed as dist
    import xvit
    from xvit.ddp import BucketedGradReducer
    monkeypatch.setenv("XVIT_STREAMS", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    cfg = R.make_config("small")                       # 3 modalities, 3-ring: three branch streams, three fusions
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img, labels = R.make_inputs(cfg, 6, seed=2)
    img, labels = img.to(dev()), labels.to(dev())
    ref = _grads(model, img, labels)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        red = BucketedGradReducer(list(model.parameters()), bucket_bytes=128 << 10)
        assert len(red.buckets) >= 8
        names = {id(p): k for k, p in model.named_parameters()}
        spans = [{names[id(p)].split(".blocks.")[1].split(".")[0] for p in b.params if ".blocks." in names[id(p)]} for b in red.buckets]
        assert any(len(s) >= 2 for s in spans), "a bucket must mix gradients of different branch streams for this test to bite"
        for _ in range(3):                             # repeated: a race would show up as run-to-run differences
            got = _grads(model, img, labels, red)
            assert all(len(b.events) >= 1 for b in red.buckets)
            for k, g in got.items():
                if g.dim() == 2:                       # weight gradients: deterministic kernels (fixed-order split-K) -> bit-identical
                    assert torch.equal(g, ref[k]), k
                else:                                  # 1-D gradients are fp32 atomic sums: order-dependent last bits
                    assert rel(g, ref[k]) < 1e-5 or float(ref[k].abs().max()) < 1e-6, k
        assert max(len(b.events) for b in red.buckets) >= 2     # some bucket really waited on more than one stream
        red.remove()
    finally:
        dist.destroy_process_group()


def test_graphed_step_with_reducer_one_rank_rccl(monkeypatch):
    """The captured step and the reducer together (the reference's DDP setting at 8 volume pairs per GPU is host-bound without the
    graph): weight gradients are written straight into the bucket views, the bucket all-reduces are nodes of the captured graph.
    1-rank RCCL group on this box: replayed + reduced gradients == eager gradients without a reducer (bit for bit for the weight
    matrices), p.grad ARE the bucket views, new inputs are followed, and the eager reducer path with the gradient sink agrees too."""
    import torch.distributed as dist
    import xvit
    import xvit.functional as XF
    from xvit.ddp import BucketedGradReducer
    from xvit.graph import GraphedStep
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    cfg = R.make_config("small")
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    ins = [tuple(t.to(dev()) for t in R.make_inputs(cfg, 6, seed=s)) for s in (2, 7)]
    refs = [_grads(model, *i) for i in ins]
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        red = BucketedGradReducer(list(model.parameters()), bucket_bytes=128 << 10)
        # eager + gradient sink: no pack copy for the weight matrices
        XF.GRAD_SINK = red.grad_sink(model)
        try:
            got = _grads(model, *ins[0], red)
        finally:
            XF.GRAD_SINK = None
        sunk = 0
        for k, p in model.named_parameters():
            assert p.grad.data_ptr() == red._view_of[id(p)].data_ptr(), k
            if p.dim() == 2:
                assert torch.equal(got[k], refs[0][k]), k
                sunk += 1
            else:
                assert rel(got[k], refs[0][k]) < 1e-5 or float(refs[0][k].abs().max()) < 1e-6, k
        assert sunk > 20
        # captured step + captured collectives
        step = GraphedStep(model, *ins[0], reducer=red)
        for which in (0, 1, 0):
            step(*ins[which])
            torch.cuda.synchronize()
            for k, p in model.named_parameters():
                assert p.grad.data_ptr() == red._view_of[id(p)].data_ptr(), k
                ref = refs[which][k]
                if p.dim() == 2:
                    assert torch.equal(p.grad, ref), (which, k)
                else:
                    assert rel(p.grad, ref) < 1e-5 or float(ref.abs().max()) < 1e-6, (which, k)
        red.remove()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (the gpurun box has one; the driver's 8-GPU node runs it)")
def test_model_cross_two_rccl_ranks(tmp_path):
    import xvit
    per_rank = 4
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_ddp_rccl_worker.py"), str(tmp_path), str(per_rank)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "rccl ranks: 2" in r.stdout
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    # single-process reference on the global batch
    cfg = R.make_config("tiny")
    model = xvit.ModelCross(cfg).to(dev())
    model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    img, labels = R.make_inputs(cfg, 2 * per_rank, seed=3)
    ref = _grads(model, img.to(dev()), labels.to(dev()))
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k                                    # all-reduce: the same bits on every rank
        assert rel(g0[k], ref[k].cpu()) < 2e-3 or float(ref[k].abs().max()) < 1e-6, k   # mean of shard gradients == global-batch gradient (bf16 operands: per-shard rounding differs)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k                                    # replicas stay in lock-step through 3 Adam steps
    q0, q1 = torch.load(tmp_path / "graph0.pt"), torch.load(tmp_path / "graph1.pt")
    assert q0["worst"] < 1e-5 and q1["worst"] < 1e-5, (q0["worst"], q1["worst"])   # captured step + captured collectives == eager reducer
    for k in q0["grads"]:
        assert torch.equal(q0["grads"][k], q1["grads"][k]), k
