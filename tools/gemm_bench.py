#!/usr/bin/env python3
"""Micro-benchmark of xvit_gemm on the model's real shapes (configs[1], per-GPU batch 32).
HIP-event timed, random bf16 operands (never zeros: DVFS)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402

T = 32 * 513
SHAPES = [  # (layout, M, N, K, split, label)
    (ops.NT, T, 2304, 768, 1, "qkv"), (ops.NT, T, 768, 768, 1, "out-proj"), (ops.NT, T, 3072, 768, 1, "ffn1"), (ops.NT, T, 768, 3072, 1, "ffn2"),
    (ops.NT, 2 * T, 768, 4096, 1, "patch-embed"), (ops.NT, T, 1536, 768, 1, "kv"),
    (ops.NN, T, 768, 2304, 1, "d-qkv"), (ops.NN, T, 768, 768, 1, "d-out"), (ops.NN, T, 3072, 768, 1, "d-ffn2"), (ops.NN, T, 768, 3072, 1, "d-ffn1"),
    (ops.TN, 2304, 768, T, 0, "w-qkv"), (ops.TN, 768, 768, T, 0, "w-out"), (ops.TN, 3072, 768, T, 0, "w-ffn1"), (ops.TN, 768, 3072, T, 0, "w-ffn2"),
    (ops.TN, 768, 4096, 2 * T, 0, "w-patch"),
]


def main():
    from xvit.functional import _wgrad_split
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("ITERS", "20"))
    only = set(sys.argv[1:])
    tot_t = tot_f = 0.0
    for layout, M, N, K, split, label in SHAPES:
        if only and label not in only:
            continue
        if layout == ops.NT:
            A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
        elif layout == ops.NN:
            A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        else:
            A, B = torch.randn(K, M, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        split = split or _wgrad_split(M, N, K)
        C = torch.empty(M, N, dtype=torch.float32 if layout == ops.TN else torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm(layout, A, B, C, split_k=split)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.gemm(layout, A, B, C, split_k=split)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / iters * 1e3
        fl = 2.0 * M * N * K
        tot_t += us; tot_f += fl
        print(f"{('NT','NN','TN')[layout]} {label:12s} M={M:6d} N={N:5d} K={K:6d} split={split:2d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
    print(f"TOTAL {tot_t:9.1f} us  {tot_f / tot_t / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
