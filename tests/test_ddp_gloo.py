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


# ---- sync_dist of the on-device epoch statistics (xvit/metrics.py): the state vectors are summed over the ranks -----
def _oracle_state(steps):
    """the 16-double state xvit_binary_metrics_step would have accumulated, from the oracle (no GPU here)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as R
    st = torch.zeros(16, dtype=torch.float64)
    for logits, labels in steps:
        m = R.binary_step_metrics(logits, labels)
        b = float(len(labels))
        st[0:4] += torch.tensor(m["counts"], dtype=torch.float64)
        st[4] += b
        st[5] += 1
        st[6:13] += b * torch.tensor([m[k] for k in R.METRIC_KEYS], dtype=torch.float64)
    return st


def _rank_steps(rank):
    g = torch.Generator().manual_seed(50 + rank)
    return [(torch.randn(b, 2, generator=g), torch.randint(0, 2, (b,), generator=g)) for b in ((9, 9, 4) if rank == 0 else (9, 9, 9, 2))]


def _metrics_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
    from xvit.metrics import BinaryEpochMetrics
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = BinaryEpochMetrics("cpu")
    acc.state.copy_(_oracle_state(_rank_steps(rank)))
    torch.save((acc.compute("val"), acc.compute("val", sync_dist=False)), os.path.join(out_dir, f"m{rank}.pt"))
    dist.destroy_process_group()


def test_epoch_statistics_sync_dist_world2(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as R
    port = _free_port()
    mp.spawn(_metrics_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (all0, loc0), (all1, loc1) = torch.load(tmp_path / "m0.pt"), torch.load(tmp_path / "m1.pt")
    assert all0 == all1                                           # every rank logs the same epoch values
    ref = R.epoch_metrics(_rank_steps(0) + _rank_steps(1))        # = the weighted mean over the global batch stream
    for k in R.METRIC_KEYS:
        assert abs(all0[f"val_{k}"] - ref[k]) < 1e-12, k
    assert all0["val_confusion"]["samples"] == 22 + 29 and all0["val_confusion"]["steps"] == 7
    assert loc0["val_confusion"]["samples"] == 22 and loc1["val_confusion"]["samples"] == 29


# ---- bucket plan over the REAL ModelCross parameter list (heads first, shared embedding last) --------
def test_bucket_order_over_model_cross_parameters():
    sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as R
    import xvit
    from xvit.ddp import plan_buckets
    cfg = R.make_config("small")
    model = xvit.ModelCross(cfg)                       # parameters only: nothing is launched on CPU
    names = {id(p): k for k, p in model.named_parameters()}
    plan = plan_buckets(model.parameters(), bucket_bytes=256 << 10)
    assert len(plan) >= 4
    flat = [names[id(p)] for b in plan for p in b]
    assert sorted(flat) == sorted(names.values())      # every parameter in exactly one bucket

    def group(k):
        if k.startswith(("mlp_head", "norm.")):
            return 0
        if k.startswith("transformer."):
            return 1 + (cfg.num_multi_blocks - 1 - int(k.split(".")[1]))      # last MultiScaleBlock first
        return 1 + cfg.num_multi_blocks                                        # pos_embedding, patch_to_embedding, cls_token

    order = [group(k) for k in flat]
    assert order == sorted(order), "gradient buckets must follow backward's readiness order"
    assert {k.split(".")[0] for k in flat[-4:]} == {"pos_embedding", "patch_to_embedding", "cls_token"}   # the shared embedding comes last
    assert all(group(names[id(p)]) == 0 for p in plan[0][:4])
