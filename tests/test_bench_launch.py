"""CPU: `python bench.py --gpus N` must start its own N ranks when it is not already running under
torch.distributed.run (the driver's N=1 call and its N>1 `python -m torch.distributed.run ... bench.py` call both reach the
same main(); a bare `--gpus 2` used to exit with an error).  XVIT_BENCH_DRYRUN=1 swaps the GPU work for a gloo rendezvous so
the whole launch -> rendezvous -> one-JSON-line flow runs here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XVIT_BENCH_DRYRUN="1", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bare_gpus_2_launches_two_ranks():
    out = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1      # flags reach the children
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert all(r["world"] == 2 for r in ranks) and ranks[0]["pid"] != ranks[1]["pid"]


def test_under_torch_distributed_run_env_no_second_launch():
    # the driver's form: RANK / WORLD_SIZE already set -> main() must use them, not spawn again
    out = _run({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"}, "--gpus", "1")
    assert out["n_gpus"] == 1 and len(out["ranks"]) == 1


def test_child_failure_propagates():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(XVIT_BENCH_DRYRUN="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
