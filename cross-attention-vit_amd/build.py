"""Build libxvit_hip.so (gfx950 code objects + the C-ABI) in-tree with hipcc.

    python cross-attention-vit_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "xvit", "libxvit_hip.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["core.hip", "gemm.hip", "linear_f32.hip", "layernorm.hip", "attention.hip", "cls_xattn.hip", "misc.hip", "metrics.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def _deps_mtime():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "xvit.h"), __file__]
    return max(os.path.getmtime(f) for f in files)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _deps_mtime():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB)")
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    build(force=ap.parse_args().force)
