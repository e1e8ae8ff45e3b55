#!/usr/bin/env python3
"""LayerNorm forward / backward at the step's shapes (rows = 126 x 513, d = 768), GB/s of algorithmic bytes.
    [XVIT_LIB=other.so] python tools/ln_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 126) * 513, 768
x = torch.randn(rows, d, device=dev)
g, b = torch.randn(d, device=dev), torch.randn(d, device=dev)
dy = torch.randn(rows, d, device=dev).bfloat16()
dres = torch.randn(rows, d, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
us = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
print(f"ln_fwd           {us:7.1f} us  {rows * d * 6 / us / 1e3:7.0f} GB/s")
dg, db, s1, s2 = (torch.zeros(d, device=dev) for _ in range(4))
for name, kw, bytes_per in (("ln_bwd (LN1: dres)", dict(dres=dres), 14), ("ln_bwd (LN2: dres, bf16 copy, 2 sums)", dict(dres=dres, want_bf16=True, dxsum=s1, dressum=s2), 16),
                            ("ln_bwd (no dres)", dict(), 10)):
    us = timed(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, **kw))
    print(f"{name:40s} {us:7.1f} us  {rows * d * bytes_per / us / 1e3:7.0f} GB/s")
