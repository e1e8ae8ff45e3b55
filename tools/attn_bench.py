#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels at the model's shape (HIP-event timed, random data)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H = 12
    for B, N in [(126, 513), (126, 512), (8, 513), (8, 4097)]:
        d = H * 64
        qkv = torch.randn(B * N, 3 * d, device=dev).bfloat16()
        do = torch.randn(B * N, d, device=dev).bfloat16()
        o, lse = ops.attn_fwd(qkv, B, N, H, 0.125)
        for name, fn, fl in (("fwd", lambda: ops.attn_fwd(qkv, B, N, H, 0.125), 4.0), ("fwd fp8 (quantise + attention)", lambda: ops.attn_fwd_fp8(qkv, B, N, H, 0.125), 4.0),
                             ("bwd", lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, 0.125), 10.0)):
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            flops = fl * B * H * N * N * 64
            print(f"attn {name:32s} B={B:3d} N={N:5d}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s algorithmic ({flops / us / 1e6 / 25.16:5.1f} % of MFMA peak)", flush=True)


if __name__ == "__main__":
    main()
