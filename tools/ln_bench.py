#!/usr/bin/env python3
"""Micro-benchmark of LayerNorm fwd/bwd at the model's shape (HIP-event timed)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402


def timeit(fn, n=30):
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
    for rows, d in [(42 * 513, 768), (32 * 513, 768)]:
        x = torch.randn(rows, d, device=dev)
        g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        dy = torch.randn(rows, d, device=dev).bfloat16()
        dres = torch.randn(rows, d, device=dev)
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
        acc = torch.zeros(4 * d, device=dev)
        a0, a1, a2, a3 = acc.split(d)
        us_f = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
        us_b = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, a0, a1, dres=dres, want_bf16=True, dxsum=a2, dressum=a3))
        print(f"LN rows={rows} d={d}: fwd {us_f:6.1f} us {rows * d * 6 / us_f / 1e3:7.1f} GB/s | bwd {us_b:6.1f} us {rows * d * 16 / us_b / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
