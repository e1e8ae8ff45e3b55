#!/usr/bin/env python3
"""Headline benchmark: 3-D patch-tokens/sec, forward + backward, 2-modality 128^3 p16 cross-attention
ViT (BASELINE.json configs[1]: d=768, 12 heads, mlp 3072, 2x2 blocks, N=513) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = zero-grad, per-step bf16 re-cast of the fp32 master weights (what an optimizer step
forces in training), forward incl. loss, backward, and for N>1 the complete bucketed gradient
all-reduce over RCCL (overlapped with backward on a side stream).  Optimizer step and data
loading are excluded (metric definition).  Inputs are synthetic and already resident
in HBM; weak scaling (fixed per-GPU batch).  Rank 0 prints ONE JSON line.

Also reported in that line:
  roofline      the kernel family with the largest share of step time, priced live with HIP events
                on the launch stream (extra profiled steps after the timed region)
  kernels       the same pricing for every kernel family (MFMA TFLOP/s or HBM GB/s)
  cpu_baseline  the CPU oracle (oracle/ref_cpu.py, a port of the reference path) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

def load_traffic():
    """HBM bytes per launch of each kernel family, from the newest rocprofv3 PMC summary committed under profiles/
    (`tools/profile_session.sh` + `tools/profile_summary.py`: separate --pmc FETCH_SIZE / WRITE_SIZE passes over this very
    command, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note).  PMC counters cannot be collected from inside the
    timed process, so `roofline.traffic` quotes that file (and names it); None when the batch differs or no file exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, {}, None
    try:
        d = json.load(open(files[-1]))
        return os.path.basename(files[-1]), d.get("traffic", {}), d.get("per_gpu_batch")
    except Exception:
        return None, {}, None


MFMA_PEAK_TF = 2516.0   # bf16 dense, MI355X_MICROARCH.md: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0   # HBM3E spec


def base_config(dropout=0.0):
    from types import SimpleNamespace
    return SimpleNamespace(
        hidden_dim=768, mlp_dim=3072, num_heads=12, num_multi_blocks=2, num_self_blocks=2,
        img_size=(128, 128, 128), patch_size=(16, 16, 16), num_modalities=2, attn_order={"0": "1", "1": "0"},
        num_classes=2, dropout=dropout, lr=1e-4, weight_decay=0.0, optim_params={"T_max": 1, "eta_min": 0.0}, label_smoothing=0.0)


def flops_per_sample(cfg):
    """Closed form of BASELINE.md §3 -> (fwd, fwd+bwd) matmul FLOPs per sample."""
    D, H, W = cfg.img_size
    dp, hp, wp = cfg.patch_size
    P = (D // dp) * (H // hp) * (W // wp)
    N, pd, d, f, M = P + 1, dp * hp * wp, cfg.hidden_dim, cfg.mlp_dim, cfg.num_modalities
    PE = 2 * M * P * pd * d
    SAB = 2 * N * d * 3 * d + 2 * (2 * N * N * d) + 2 * N * d * d + 2 * (2 * N * d * f)
    CAB = 2 * (2 * N * d * d) + 2 * (2 * d * d) + 2 * (2 * N * d) + 2 * (2 * d * f)
    HEAD = M * (2 * d * f + 2 * f * cfg.num_classes)
    fwd = PE + cfg.num_multi_blocks * (M * cfg.num_self_blocks * SAB + len(cfg.attn_order) * CAB) + HEAD
    return fwd, 3 * fwd - PE, P


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


T_START = time.perf_counter()


def usable_cores():
    """Cores this process may actually use: min(affinity, cgroup CPU quota, 32).  The GPU box is a 256-CPU host shared by the
    8 GPUs' tenants, a 1-GPU job's share is 16: torch.set_num_threads(os.cpu_count()) would put 256 threads on
    16 cores and time the oversubscription, not the path — so the share is used, and both numbers are stated."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(cfg, seconds_budget=30.0):
    """The oracle (a CPU port of the reference path, oracle/ref_cpu.py) on this host's cores: BASELINE.md 4 protocol — config[1] at
    B = 2 and B = 8, forward only and forward + backward, median after one warm-up, bounded to ~seconds_budget of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as R
    cores = usable_cores()
    torch.set_num_threads(cores)
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    log(f"cpu_baseline: {cores} threads (os.cpu_count()={os.cpu_count()}, {cpu_model})")
    _, both_f, P = flops_per_sample(cfg)
    sd = R.make_state_dict(cfg, seed=0)
    t_all = time.perf_counter()

    def med(fn, most):
        times = []
        for i in range(most + 1):
            t0 = time.perf_counter()
            fn()
            if i > 0:
                times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > seconds_budget and times:
                break
        return statistics.median(times), len(times)

    points = {}
    for B, most in ((2, 5), (8, 2)):
        if points and time.perf_counter() - t_all > seconds_budget * 0.6:
            break
        img, labels = R.make_inputs(cfg, B, seed=0)

        def fwd_only():
            with torch.no_grad():
                R.model_cross_forward(sd, img, labels, cfg)

        tf, nf = med(fwd_only, min(most, 3))
        tb, nb = med(lambda: R.model_cross_loss_and_grads(sd, img, labels, cfg), most)
        points[f"B{B}"] = {"fwd_ms": round(tf * 1e3, 1), "fwd_bwd_ms": round(tb * 1e3, 1), "patch_tokens_per_s": round(B * cfg.num_modalities * P / tb, 1),
                           "tflops_fwd_bwd": round(both_f * B / tb / 1e12, 3), "steps": [nf, nb]}
        log(f"cpu_baseline B={B}: {points[f'B{B}']}")
    head = points["B2"]
    return {"value": head["patch_tokens_per_s"], "unit": "patch-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/ref_cpu.py (eager fp32 PyTorch, op for op the reference path) at config[1], batch 2, fwd+bwd: median of {head['steps'][1]} steps after 1 warm-up "
                      f"({head['fwd_bwd_ms']:.0f} ms/step; fwd only {head['fwd_ms']:.0f} ms); B = 8 beside it under `points`",
            "points": points, "os_cpu_count": os.cpu_count(), "threads": torch.get_num_threads(), "cpu_model": cpu_model, "torch": torch.__version__,
            "threads_note": "threads = this job's CPU share on the GPU box (sched_getaffinity / cgroup quota), not os.cpu_count(): the other cores belong to other tenants"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one process per GPU, RCCL)
    and pass rank 0's JSON line through.  Runs BEFORE this process touches the GPU (it never does): the children are
    fresh processes started by `python -m torch.distributed.run`, exactly the command line the driver uses."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    log(f"launching {n} ranks: {' '.join(cmd[1:9])} ...")
    return subprocess.run(cmd, env=env).returncode          # stdout/stderr inherited: rank 0 prints the one JSON line


def dry_run(args, rank, world):
    """XVIT_BENCH_DRYRUN=1: the launch / rendezvous / one-JSON-line flow without any GPU work (gloo), so the N>1 path of
    this file is exercised by the CPU test suite (tests/test_bench_launch.py)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "world": world, "pid": os.getpid()})
    dist.barrier()
    if rank == 0:
        emit({"dryrun": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks": seen})
    dist.destroy_process_group()


def small_batch_points(model, cfg, dev, P, batches, steps=12):
    """Same step at the reference's batch sizes, eager and as ONE HIP-graph replay (xvit.graph.GraphedStep): below
    ~B=32 the eager step is bound by host launch issue, which the graph removes.  Extra keys only — `value` stays the
    headline batch."""
    import xvit
    from xvit.graph import GraphedStep
    pts = {}
    M = cfg.num_modalities
    params = list(model.parameters())
    for B in batches:
        gen = torch.Generator().manual_seed(77 + B)
        img = torch.randn(B, M, 1, *cfg.img_size, generator=gen).to(dev, torch.bfloat16)
        labels = torch.randint(0, cfg.num_classes, (B,), generator=gen).to(dev)

        def eager():
            for p in params:
                p.grad = None
            xvit.invalidate_shadows()
            model(img, labels)[1].backward()

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / steps

        te = timed(eager)
        entry = {"eager_ms": round(te * 1e3, 3), "eager_tokens_per_s": round(B * M * P / te, 1)}
        try:
            g = GraphedStep(model, img, labels)
            tg = timed(lambda: g())
            entry.update(graph_ms=round(tg * 1e3, 3), graph_tokens_per_s=round(B * M * P / tg, 1))
            del g
        except Exception as exc:   # capture is an optimisation: report, never fail the bench line
            entry["graph_error"] = f"{type(exc).__name__}: {exc}"[:200]
        pts[f"B{B}"] = entry
        log(f"small batch B={B}: {entry}")
    for p in params:
        p.grad = None
    return pts


def small_batch_child(timeout=240):
    """Run `bench.py --small-batch-only` as a child (the parent's GPU work is finished and synchronised; it keeps its memory)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--small-batch-only"], capture_output=True, text=True, timeout=timeout)
        sys.stderr.write(r.stderr[-4000:])
        line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
        if r.returncode != 0 or line is None:
            return {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-300:]}
        return json.loads(line)
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner at
    communicator creation): from here on file descriptor 1 is stderr for everybody, and emit() writes the line to the real stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=126, help="per-GPU batch of volume pairs.  Throughput rises with batch until ~126 (42: 2.79 M, 84: 2.99 M, 126: 3.03 M, "
                    "252: 3.06 M tokens/s; 18.7 GB of the 288 GB HBM at 126); 126*513 rows = 253 row tiles of 256 -> 759 / 1518 / 2277 / 3036 tiles per GEMM "
                    "= 2.96 .. 11.9 rounds of 256 CUs, so the last round of every launch is nearly full")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true", help="(accepted for old command lines; the rocm-smi clock field is gone: the in-kernel clock of the "
                    "GEMM / attention loops is measured by the diagnostic builds under tools/ and kept in profiles/)")
    ap.add_argument("--small-batch-only", action="store_true", help="internal: print the small-batch points as one JSON line and exit (run by the parent bench in a child process)")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the extra B=8 / B=32 points (eager and HIP-graph replay)")
    ap.add_argument("--detail", action="store_true", help="per-shape GEMM table on stderr (diagnostic)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph (xvit.graph.GraphedStep); "
                    "pays off when the step is host-bound, i.e. at small per-GPU batch")
    ap.add_argument("--single-stream", action="store_true", help="run the modality branches on one stream (used for the rocprof "
                    "summary under profiles/, so per-kernel durations are not stretched by a concurrently running kernel)")
    args = ap.parse_args()

    if args.single_stream:
        os.environ["XVIT_STREAMS"] = "0"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if os.environ.get("XVIT_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("XVIT_FORCE_DIST") == "1"   # the latter: rehearse the N>1 code path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import xvit
    from xvit import ops
    from xvit.ddp import BucketedGradReducer

    cfg = base_config()
    fwd_f, both_f, P = flops_per_sample(cfg)
    M, B = cfg.num_modalities, args.batch
    torch.manual_seed(0)
    model = xvit.ModelCross(cfg).to(dev)
    model.train()
    if args.small_batch_only:
        emit(small_batch_points(model, cfg, dev, P, (8, 32)))
        return
    params = [p for p in model.parameters()]
    reducer = BucketedGradReducer(params, bucket_bytes=32 << 20) if use_dist else None
    exposed = []                          # (start, end) events around reducer.finish(): the time the compute stream waits for the collectives' tail

    gen = torch.Generator().manual_seed(1234 + rank)
    img = torch.randn(B, M, 1, *cfg.img_size, generator=gen).to(dev, torch.bfloat16)   # random (not zero) data: MI355X_MICROARCH.md DVFS note
    labels = torch.randint(0, cfg.num_classes, (B,), generator=gen).to(dev)

    def step():
        for p in params:
            p.grad = None
        xvit.invalidate_shadows()          # weights "changed": pay the fp32 -> bf16 operand cast every step
        _, loss = model(img, labels)
        loss.backward()
        if reducer is not None:
            if timing[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                reducer.finish()
                e1.record()
                exposed.append((e0, e1))
            else:
                reducer.finish()
        return loss

    timing = [False]
    if reducer is not None:
        import xvit.functional as XF
        model(img, labels)                 # the first forward moves the weights into their flat buffers: map the sink afterwards
        XF.GRAD_SINK = reducer.grad_sink(model)   # weight-gradient kernels write into the bucket views: no pack pass
    if args.graph:
        from xvit.graph import GraphedStep
        graphed = GraphedStep(model, img, labels, reducer=reducer)   # N > 1: the bucket all-reduces are nodes of the captured graph
        eager_step = step

        def step():                                   # noqa: F811
            return graphed()[1]

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    log(f"model + inputs ready (B={B}/GPU, world={world})")
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize(dev)
            log("first step done")
    fence()
    log("warm-up done")
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timing[0] = False
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    log(f"timed region: {ms:.2f} ms/step")
    tokens_per_s = world * B * M * P * args.steps / dt
    loss_val = float(loss.detach())
    del loss          # frees the last step's autograd graph (a live graph pins its AccumulateGrad nodes to the streams they were created
                      # on, which a later HIP-graph capture of the step cannot synchronise with: see xvit/graph.py)

    out = {
        "metric": "3D patch-tokens/sec fwd+bwd, 2-modality 128^3 p16 ViT", "value": round(tokens_per_s, 1), "unit": "patch-tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "configs[1]: ModelCross d=768 H=12 mlp=3072 2x2 blocks, 2 modalities, 128^3 volume, 16^3 patches (N=513), dropout 0",
                   "per_gpu_batch": B, "global_batch": world * B, "seq_len": P + 1, "parallelism": f"dp{world}",
                   "step": "zero-grad + weight bf16 cast + fwd + loss + bwd" + (" + bucketed grad all-reduce (RCCL, side stream)" if world > 1 else ""),
                   "optimizer_step": "excluded", "launch": "one HIP graph replay per step" if args.graph else "eager (one launch per kernel)"},
        "model_tflops_per_gpu": round(both_f * B / (dt / args.steps) / 1e12, 1),
        "mfma_frac_of_peak_step": round(both_f * B / (dt / args.steps) / 1e12 / MFMA_PEAK_TF, 4),
        "loss": round(loss_val, 5),
        "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
    }
    if use_dist:
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        out["collective"] = {"backend": dist.get_backend(), "rccl_version": ver, "ranks": dist.get_world_size(),
                             "buckets": len(reducer.buckets), "bucket_mib": 32, "exposed_launches": reducer.exposed_launches,
                             "gradient_sink": "weight gradients written into the bucket views (no pack copy); mean applied by the collective (ReduceOp.AVG)",
                             "exposed_ms": round(sum(a.elapsed_time(b) for a, b in exposed) / max(len(exposed), 1), 3) if exposed else None,
                             "exposed_ms_note": "per step, rank 0: time reducer.finish() holds the compute stream waiting for the tail of the bucket all-reduces "
                                                "(eager launches; with --graph the collectives are nodes of the replayed graph and are not separable)"}

    # ---- per-kernel pricing with HIP events on the launch stream (rank 0) --------------------
    if args.graph:
        step = eager_step                             # per-kernel pricing needs individual launches
    rec = None
    if args.profile_steps > 0:
        # kernels are priced one at a time: the two modality streams are merged for these extra steps, otherwise
        # concurrently running kernels stretch each other's event brackets (the timed region above keeps them).
        # EVERY rank runs the same extra steps (their backward launches the reducer's collectives from its hooks, so a
        # rank running them alone would wait for its peers forever); only rank 0 records.
        prev_streams = os.environ.get("XVIT_STREAMS")
        os.environ["XVIT_STREAMS"] = "0"
        step()
        torch.cuda.synchronize(dev)
        if rank == 0:
            ops.PROFILE, ops.PROFILE_SHAPES = [], args.detail
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize(dev)
        if rank == 0:
            rec, ops.PROFILE = ops.PROFILE, None
        if prev_streams is None:
            os.environ.pop("XVIT_STREAMS")
        else:
            os.environ["XVIT_STREAMS"] = prev_streams
        if use_dist:
            dist.barrier()
    if rec is not None:
        if args.detail:
            det = {}
            for name, work, kind, s, e in rec:
                d = det.setdefault(name, [0, 0.0, 0.0])
                d[0] += 1; d[1] += s.elapsed_time(e); d[2] += work
            for name, (n, ms_, w) in sorted(det.items(), key=lambda kv: -kv[1][1]):
                unit = "TF/s" if name.startswith(("gemm", "attn")) else "GB/s"
                rate = w / (ms_ * 1e-3) / (1e12 if unit == "TF/s" else 1e9)
                print(f"  {name:58s} x{n // args.profile_steps:3d} {ms_ / n * 1e3:9.1f} us  {rate:8.1f} {unit}  total {ms_ / args.profile_steps:7.3f} ms/step", file=sys.stderr)
            rec = [(n.split("[")[0], w, k, s, e) for n, w, k, s, e in rec]
        agg = {}
        for name, work, kind, s, e in rec:
            a = agg.setdefault(name, {"kind": kind, "launches": 0, "ms": 0.0, "work": 0.0})
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["work"] += work
        total_ms = sum(a["ms"] for a in agg.values())
        kernels = {}
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            rate = a["work"] / (a["ms"] * 1e-3)
            k = {"launches_per_step": a["launches"] // args.profile_steps, "avg_us": round(a["ms"] / a["launches"] * 1e3, 2),
                 "share_of_kernel_time": round(a["ms"] / total_ms, 4)}
            if a["kind"] == "flop":
                k.update(bound="mfma", achieved=round(rate / 1e12, 1), peak=MFMA_PEAK_TF, unit="TFLOP/s", frac=round(rate / 1e12 / MFMA_PEAK_TF, 4))
            else:
                k.update(bound="hbm", achieved=round(rate / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(rate / 1e9 / HBM_PEAK_GBS, 4))
            kernels[name] = k
        dom = next(k for k in kernels if "+splitk" not in k)   # a family that is exactly one kernel symbol
        t_file, t_map, t_batch = load_traffic()
        out["roofline"] = {"kernel": dom, **{k: kernels[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac")},
                           "avg_launch_us": kernels[dom]["avg_us"], "traffic": t_map.get(dom) if t_batch == B else None,
                           "traffic_source": f"profiles/{t_file}" if t_file and t_batch == B else None,
                           "note": "family with the largest share of step time; algorithmic FLOPs of all its launches / their summed duration "
                                   "(HIP events on the launch stream, modality streams merged while pricing); traffic = HBM bytes per launch "
                                   "from the rocprofv3 PMC passes of this command committed under profiles/ (FETCH_SIZE doubled per the gfx950 note)"}
        out["kernels"] = kernels
        # the north-star's attention target, stated separately
        if "attn_fwd" in kernels:
            out["attention_mfma_frac"] = {"fwd": kernels["attn_fwd"]["frac"], "bwd": kernels.get("attn_bwd", {}).get("frac")}

    # ---- the reference's own batch sizes (main_mist.py:206 trains at 8 per GPU), next to the headline batch -------------
    # Measured in a FRESH CHILD PROCESS after the headline: the points capture HIP graphs, and a native crash inside a capture
    # (seen in round 2) must not cost the line above.  The child is a new `python bench.py --small-batch-only`, never a re-exec.
    if world == 1 and not use_dist and not args.no_small_batch:
        out["small_batch"] = small_batch_child()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0:
        emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
