#!/usr/bin/env python3
"""A/B of the CLS-peel attention form against the tile-grid form at N = 64 m + 1 (and the neighbouring N = 64 m), interleaved in one
process (HIP events, random data).  Under `rocprofv3 --kernel-trace --stats` the per-kernel split (main kernel, merge kernel) is in the stats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    H, d = 12, 768
    shapes = [(126, 513), (8, 4097), (8, 513)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    for B, N in shapes:
        res = {}
        for rnd in range(3):
            for label, n, peel in (("peel", N, 2), ("grid", N, 0), ("N-1", N - 1, 1)):
                ops.set_option("attn_peel", peel)
                qkv = torch.randn(B * n, 3 * d, device=dev).bfloat16()
                do = torch.randn(B * n, d, device=dev).bfloat16()
                o, lse = ops.attn_fwd(qkv, B, n, H, 0.125)
                f = timed(lambda: ops.attn_fwd(qkv, B, n, H, 0.125))
                b = timed(lambda: ops.attn_bwd(qkv, o, do, lse, B, n, H, 0.125))
                res.setdefault(label, []).append((f, b))
        ops.set_option("attn_peel", 1)
        for label, v in res.items():
            f = sorted(x[0] for x in v)[len(v) // 2]
            b = sorted(x[1] for x in v)[len(v) // 2]
            n = N - 1 if label == "N-1" else N
            print(f"B={B:3d} N={n:5d} {label:5s}: fwd {f:7.1f} us ({4.0 * B * H * n * n * 64 / f / 1e6 / 25.16:5.1f} % of MFMA peak)   "
                  f"bwd {b:7.1f} us ({10.0 * B * H * n * n * 64 / b / 1e6 / 25.16:5.1f} %)", flush=True)


if __name__ == "__main__":
    main()
