"""CPU: static guard on bench.py's multi-GPU flow.  With world_size > 1 every step's backward launches the gradient
reducer's collectives from autograd hooks, so a step executed by rank 0 alone (e.g. for per-kernel pricing) would wait for
its peers forever.  No call of step() / eager_step() may therefore sit under a condition on the rank."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mentions_rank(node):
    return any(isinstance(n, ast.Name) and n.id == "rank" for n in ast.walk(node))


def test_no_training_step_is_conditional_on_the_rank():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    offenders = []
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and _mentions_rank(node.test):
            for sub in node.body:
                for call in ast.walk(sub):
                    if isinstance(call, ast.Call) and isinstance(call.func, ast.Name) and call.func.id in ("step", "eager_step"):
                        offenders.append(call.lineno)
    assert not offenders, f"bench.py: step() under a rank condition at lines {offenders}"


def test_single_gpu_extras_are_gated_on_world_size_one():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "world == 1 and not use_dist and not args.no_small_batch" in src and "world == 1 and not args.no_cpu_baseline" in src


def test_small_batch_points_run_in_a_child_process_after_the_headline():
    """The optional small-batch points capture HIP graphs; a native crash there must not cost the headline line: they run in a fresh
    `bench.py --small-batch-only` child (subprocess, never a re-exec of the GPU process), and the headline is already measured."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--small-batch-only" in src and "subprocess.run([sys.executable, os.path.abspath(__file__), \"--small-batch-only\"]" in src
    assert "os.exec" not in src
    body = src[src.index("def main():"):]
    assert body.index("small_batch_child()") > body.index("tokens_per_s = world * B * M * P * args.steps / dt")
