#!/usr/bin/env python3
"""K sweep of the big-tile GEMM at a grid of exactly 2 tile rounds (M=16384, N=2048 -> 512 tiles on 256 CUs):
time(K) = fixed (launch + prologue + epilogue) + slope * K separates the main-loop rate from the per-tile overhead."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, N = 16384, 2048
    for layout, name in ((ops.NT, "NT"), (ops.NN, "NN"), (ops.TN, "TN")):
        for out_dtype in (torch.bfloat16, torch.float32):
            pts = []
            for K in (64, 256, 768, 1536, 3072):
                if layout == ops.NT:
                    A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
                elif layout == ops.NN:
                    A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
                else:
                    A, B = torch.randn(K, M, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
                C = torch.empty(M, N, dtype=out_dtype, device=dev)
                for _ in range(3):
                    ops.gemm(layout, A, B, C, split_k=1)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    ops.gemm(layout, A, B, C, split_k=1)
                e.record()
                torch.cuda.synchronize()
                pts.append((K, s.elapsed_time(e) / 20 * 1e3))
            (k0, t0), (k1, t1) = pts[2], pts[-1]
            slope = (t1 - t0) / (k1 - k0)
            peak_rate = 2.0 * M * N / slope / 1e6
            print(f"{name} {str(out_dtype)[6:]:8s} " + "  ".join(f"K={k}:{t:6.1f}us" for k, t in pts) +
                  f"   main-loop {peak_rate:6.0f} TF/s, fixed {t0 - slope * k0:5.1f} us per 2 rounds", flush=True)


if __name__ == "__main__":
    main()
