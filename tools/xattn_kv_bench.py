#!/usr/bin/env python3
"""Backward of the fusion's K/V projection at configs[1] (B = 126, N = 513, d = 768, H = 12): the low-rank form
(cls_xattn_bwd -> coefficients, xattn_kv_dgrad, xattn_kv_wgrad, small host products) against the dense chain
(cls_xattn_bwd -> dkv, dgrad GEMM, wgrad GEMM, column sums)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402
from xvit import functional as XF  # noqa: E402

dev = torch.device("cuda:0")
B, N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 126, 513, 12
d = 64 * H
q = torch.randn(B, d, device=dev).bfloat16(); kv = torch.randn(B * N, 2 * d, device=dev).bfloat16()
do = torch.randn(B, d, device=dev).bfloat16(); wkv = (torch.randn(2 * d, d, device=dev) / d ** 0.5).bfloat16()
hn = torch.randn(B * N, d, device=dev).bfloat16()
p = ops.cls_xattn_fwd(q, kv, B, N, H, 0.125)[1]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def dense():
    dq, dkv = ops.cls_xattn_bwd(q, kv, p, do, B, N, H, 0.125)
    return XF._dgrad(dkv, wkv), XF._wgrad(dkv, hn), ops.colsum(dkv)


def low():
    dq, coef = ops.cls_xattn_bwd(q, kv, p, do, B, N, H, 0.125, low_rank=True)
    return ops.xattn_kv_backward(coef, q, do, wkv, hn, B, N, H)


print(f"dense chain {timed(dense):8.1f} us   low-rank chain {timed(low):8.1f} us")
ops.PROFILE = []
low(); torch.cuda.synchronize()
for name, work, kind, s, e in ops.PROFILE:
    print(f"   {name:18s} {s.elapsed_time(e) * 1e3:8.1f} us")
ops.PROFILE = []
dense(); torch.cuda.synchronize()
for name, work, kind, s, e in ops.PROFILE:
    print(f"   {name:18s} {s.elapsed_time(e) * 1e3:8.1f} us")
ops.PROFILE = None
