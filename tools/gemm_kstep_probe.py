#!/usr/bin/env python3
"""What bounds the main loop of gemm_big_kernel?  (diagnostic, GPU box)

    python cross-attention-vit_amd/build.py --out tools/_bin/libxvit_probe.so -D XVIT_GEMM_CLOCK_PROBE
    XVIT_LIB=tools/_bin/libxvit_probe.so python tools/gemm_kstep_probe.py

1. K = 16384 (256 K-steps per tile: prologue and epilogue negligible) on grids of 16 .. 512 tiles, one workgroup per CU (the
   256x256 kernel forced with xvit_set_option("gemm_tile", 2)): wall time per K-step against the number of busy CUs, and —
   with the probe build — the shader cycles each workgroup took and the clock it ran at (clock64 / wall_clock64).
2. The same 256 tiles all reading ONE A panel and ONE B panel (stride-0 batch): no HBM / fabric traffic to speak of.
3. The model's Linear shapes at configs[1] (M = 54016): clock and cycles per tile, split into the K loop and the rest.
tools/mfma_clock_probe.hip is the companion: the bare MFMA loop from registers."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
HAVE_CLK = hasattr(lib, "xvit_dbg_read_clk")
if not HAVE_CLK:
    print("(library built without -D XVIT_GEMM_CLOCK_PROBE: wall times only)")


def clocks(n):
    """(mean MHz, mean / min / max shader cycles per workgroup) of the last big-tile launch's first n workgroups."""
    n = min(n, 4096)
    buf = (ctypes.c_longlong * (2 * n))()
    lib.xvit_dbg_read_clk(buf, n)
    mhz = [buf[2 * i] / (buf[2 * i + 1] / 100.0) for i in range(n) if buf[2 * i + 1] > 0]
    cyc = [buf[2 * i] for i in range(n) if buf[2 * i + 1] > 0]
    return sum(mhz) / len(mhz), sum(cyc) / len(cyc), min(cyc), max(cyc)


def timed(fn, reps):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


QUICK = os.environ.get("XVIT_PROBE_QUICK") == "1"      # section 1, NT, 16 and 256 tiles only (energy decomposition builds)
ops.set_option("gemm_tile", 2)
K = 16384
print(f"--- 1. K = {K}: one K-step = 256x256x64 per workgroup, MFMA floor 2 waves x 64 v_mfma_f32_16x16x32_bf16 x 16 cycles = 2048 cycles per SIMD")
for layout, name in ((ops.NT, "NT"),) if QUICK else ((ops.NT, "NT"), (ops.NN, "NN"), (ops.TN, "TN")):
    for tm, tn in ((4, 4), (16, 16)) if QUICK else ((4, 4), (8, 8), (16, 8), (16, 16), (32, 16)):
        M, N = 256 * tm, 256 * tn
        if layout == ops.NT:
            A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
        elif layout == ops.NN:
            A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        else:
            A, B = torch.randn(K, M, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        C = torch.empty(M, N, dtype=torch.bfloat16 if layout != ops.TN else torch.float32, device=dev)
        us = timed(lambda: ops.gemm(layout, A, B, C), 20 if QUICK else 5)
        tiles = tm * tn
        rounds = (tiles + 255) // 256
        line = f"{name} {tiles:4d} tiles ({min(tiles, 256):3d} CUs busy, {rounds} round{'s' if rounds > 1 else ' '}): {us / rounds / (K // 64) * 1e3:7.1f} ns per K-step  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s"
        if HAVE_CLK:
            mhz, cyc, _, _ = clocks(tiles)
            line += f" | {cyc / (K // 64):6.0f} shader cycles per K-step at {mhz:5.0f} MHz"
        print(line, flush=True)

if QUICK:
    sys.exit(0)
print("--- 2. every tile reads the same 8 MiB A panel and 8 MiB B panel (L2 hits only)")
for nb in (16, 64, 256):
    A = torch.randn(1, 256, K, device=dev).bfloat16().expand(nb, 256, K)
    B = torch.randn(1, 256, K, device=dev).bfloat16().expand(nb, 256, K)
    C = torch.empty(nb, 256, 256, dtype=torch.bfloat16, device=dev)
    us = timed(lambda: ops.gemm(ops.NT, A, B, C), 5)
    line = f"NT shared operands, {nb:3d} tiles: {us / (K // 64) * 1e3:7.1f} ns per K-step"
    if HAVE_CLK:
        mhz, cyc, _, _ = clocks(nb)
        line += f" | {cyc / (K // 64):6.0f} shader cycles per K-step at {mhz:5.0f} MHz"
    print(line, flush=True)
ops.set_option("gemm_tile", 0)

print("--- 3. the Linear shapes of configs[1] (M = 54016 rows), forward (NT)")
M = 54016
for (N, Kk, act, name) in ((2304, 768, ops.ACT_NONE, "qkv"), (768, 768, ops.ACT_NONE, "768 -> 768"), (3072, 768, ops.ACT_GELU, "FFN1 (GELU, z + a)"), (768, 3072, ops.ACT_NONE, "3072 -> 768")):
    A = torch.randn(M, Kk, device=dev).bfloat16(); W = torch.randn(N, Kk, device=dev).bfloat16()
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    z = torch.empty_like(C) if act == ops.ACT_GELU else None
    bias = torch.randn(N, device=dev)
    us = timed(lambda: ops.gemm(ops.NT, A, W, C, bias=bias, aux=z, act=act), 10)
    tiles = ((M + 255) // 256) * (N // 256)
    line = f"{name:20s} N={N:4d} K={Kk:4d}: {us:7.1f} us {2.0 * M * N * Kk / us / 1e6:6.0f} TFLOP/s, {tiles} tiles = {tiles / 256:.2f} rounds"
    if HAVE_CLK:
        mhz, cyc, cmin, cmax = clocks(tiles)
        ks = Kk // 64
        line += f" | {mhz:5.0f} MHz, {cyc:7.0f} cycles per tile ({cmin} .. {cmax}); {ks} K-steps x 2500 = {ks * 2500}, so {cyc - ks * 2500:6.0f} outside the K loop"
    print(line, flush=True)
