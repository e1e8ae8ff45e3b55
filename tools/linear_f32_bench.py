#!/usr/bin/env python3
"""The one-token Linears of the fusion / heads (xvit_linear_f32, fp32 operands, M = batch rows) at configs[1]: time per call (HIP events).
    python tools/linear_f32_bench.py [batch=126]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 126
d, f = 768, 3072


def timed(fn, it=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for name, K, N, act in (("wq / proj", d, d, ops.ACT_NONE), ("ffn1 +gelu", d, f, ops.ACT_GELU), ("ffn2 +res", f, d, ops.ACT_NONE), ("head 2", f, 2, ops.ACT_NONE)):
    x, W, b = torch.randn(B, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
    res = torch.randn(B, N, device=dev) if "res" in name else None
    t = timed(lambda: ops.linear_f32(x, W, b, act=act, residual=res, want_bf16=True))
    print(f"{name:12s} [{B} x {K}] x [{K} x {N}]: {t:6.1f} us   (weights {N * K * 4 / 1e6:.1f} MB -> {N * K * 4 / t / 1e3:.0f} GB/s)")
