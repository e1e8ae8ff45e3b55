"""Rank body of tests/test_ddp_gpu.py::test_model_cross_two_rccl_ranks (started by torch.distributed.run, one process per
GPU): ModelCross "tiny" with the branch/fusion streams ON, batch sharded over the ranks, 3 Adam steps through
xvit.ddp.BucketedGradReducer.  Saves the reduced gradients of step 0 and the final parameters per rank."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cross-attention-vit_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    out_dir, per_rank = sys.argv[1], int(sys.argv[2])
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import ref_cpu as R
    import xvit
    from xvit.ddp import BucketedGradReducer
    os.environ["XVIT_STREAMS"] = "1"
    cfg = R.make_config("tiny")
    torch.manual_seed(1000 + rank)                       # different init per rank: the reducer's broadcast must fix it
    model = xvit.ModelCross(cfg).to(dev)
    if rank == 0:
        model.load_state_dict(R.make_state_dict(cfg, seed=0))
    model.train()
    red = BucketedGradReducer(list(model.parameters()), bucket_bytes=64 << 10)   # small buckets: each spans both branches
    img, labels = R.make_inputs(cfg, per_rank * world, seed=3)
    sl = slice(rank * per_rank, (rank + 1) * per_rank)
    img, labels = img[sl].to(dev), labels[sl].to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for step in range(3):
        red.zero_grad()
        _, loss = model(img, labels)
        loss.backward()
        red.finish()
        if step == 0:
            torch.save({k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"g{rank}.pt"))
        opt.step()
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(out_dir, f"p{rank}.pt"))
    # the captured step with the collectives inside the graph: its reduced gradients must be the eager reducer's, on every rank
    from xvit.graph import GraphedStep
    red.zero_grad()
    _, loss = model(img, labels)
    loss.backward()
    red.finish()
    torch.cuda.synchronize()
    eager = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    step = GraphedStep(model, img, labels, reducer=red)
    step()
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in model.named_parameters():
        ref = eager[k]
        if float(ref.abs().max()) > 1e-6:
            worst = max(worst, float((p.grad - ref).norm() / ref.norm()))
    torch.save({"worst": worst, "grads": {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}}, os.path.join(out_dir, f"graph{rank}.pt"))
    if rank == 0:
        print(f"rccl ranks: {dist.get_world_size()} buckets: {len(red.buckets)}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
