"""Build libxvit_hip.so (gfx950 code objects + the C-ABI) in-tree with hipcc.

    python cross-attention-vit_amd/build.py [--force] [--out PATH] [-D MACRO ...]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree.
-D adds diagnostic macros (README "Diagnostic build macros"); such a build belongs in its own --out file (e.g.
tools/_bin/libxvit_probe.so, loaded through XVIT_LIB) so that the product library stays the plain build.
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
SOURCES = ["core.hip", "gemm.hip", "linear_f32.hip", "layernorm.hip", "attention.hip", "attention_fp8.hip", "cls_xattn.hip", "head_linear.hip", "misc.hip", "metrics.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def _deps_mtime():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "xvit.h"), __file__]
    return max(os.path.getmtime(f) for f in files)


def build(force: bool = False, verbose: bool = True, out: str = OUT, defines=()) -> str:
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _deps_mtime():
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = OBJ if out == OUT else os.path.join(OBJ, os.path.basename(out) + ".d")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)

    def compile_one(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *os.environ.get("XVIT_EXTRA_FLAGS", "").split(), *[f"-D{d}" for d in defines], "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {out} ({os.path.getsize(out) / 1e6:.1f} MB)")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--out", default=OUT)
    ap.add_argument("-D", dest="defines", action="append", default=[])
    a = ap.parse_args()
    build(force=a.force or bool(a.defines), out=a.out, defines=a.defines)
