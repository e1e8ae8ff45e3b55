"""CPU, world_size 2, gloo: the bucketed gradient reducer (xvit/ddp.py) used for the N>1 path.
Checks: replicas are broadcast-identical, reduced gradients equal the single-process gradients of
the global batch, every rank ends bit-identical, buckets overlap backward (launched from hooks),
and an unused parameter does not hang the step."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(24, 64), nn.GELU(), nn.LayerNorm(64), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 3))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
    from xvit.ddp import BucketedGradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model(seed=100 + rank)              # different init per rank: broadcast must fix it
    unused = nn.Parameter(torch.ones(5))         # never receives a gradient
    red = BucketedGradReducer(list(model.parameters()) + [unused], bucket_bytes=1 << 10)
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(16, 24, generator=g), torch.randint(0, 3, (16,), generator=g)
    shard = slice(rank * 8, (rank + 1) * 8)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    launched_in_backward = 0
    for step in range(3):
        red.zero_grad()
        loss = nn.functional.cross_entropy(model(x[shard]), y[shard])
        loss.backward()
        launched_in_backward += sum(b.launched for b in red.buckets)
        red.finish()
        if step == 0:
            torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"g{rank}.pt"))
        opt.step()
    assert launched_in_backward >= 3 * (len(red.buckets) - 1)  # all but the unused-parameter bucket fire from hooks
    assert red.exposed_launches == 3 and float(unused.grad.abs().max()) == 0.0
    torch.save(model.state_dict(), os.path.join(out_dir, f"p{rank}.pt"))
    dist.destroy_process_group()


def test_bucketed_reducer_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    # single-process reference: rank 0's initial weights, the whole batch
    model = _model(seed=100)
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(16, 24, generator=g), torch.randint(0, 3, (16,), generator=g)
    nn.functional.cross_entropy(model(x), y).backward()
    for k, p in model.named_parameters():
        assert torch.equal(g0[k], g1[k]), k                                   # identical on every rank
        assert torch.allclose(g0[k], p.grad, rtol=1e-5, atol=1e-7), k          # == global-batch gradient
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k                                   # replicas stay in lock-step
