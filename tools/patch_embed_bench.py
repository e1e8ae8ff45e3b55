#!/usr/bin/env python3
"""Patch embedding at configs[1] (B = 126, 2 modalities, 128^3 volumes, 16^3 patches, d = 768) — or, with `ucsf` as the second
argument, at configs[2]'s shape (4 modalities, 240^3 volumes: 15 patches per axis, the per-K-step placement of the weight gradient):
the fused gather kernels against patchify + GEMM on a stored patch matrix, forward and weight gradient (HIP events).
    python tools/patch_embed_bench.py [B] [ucsf]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402
from xvit import functional as XF  # noqa: E402

dev = torch.device("cuda:0")
UCSF = "ucsf" in sys.argv
B, M, patch, d = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else (4 if UCSF else 126), 4 if UCSF else 2, (16, 16, 16), 768
side = 240 if UCSF else 128
img = torch.randn(B, M, 1, side, side, side, device=dev).bfloat16()
P, pd = (side // 16) ** 3, 4096
w = (torch.randn(d, pd, device=dev) / 64).bfloat16()
bias, pos = torch.randn(d, device=dev), torch.randn(1 + P, d, device=dev)
rows = M * B * (1 + P)
dx = torch.randn(rows, d, device=dev).bfloat16()


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


flops = 2.0 * M * B * P * d * pd
x = torch.empty(rows, d, device=dev)


def unfused_fwd():
    patches = ops.patchify(img, patch, pad_cls_row=True).reshape(-1, pd)
    ops.gemm(ops.NT, patches, w, x, bias=bias, residual=pos, res_row_mod=1 + P, res_row_off=0)
    return patches


patches = unfused_fwd()
t_pat = timed(lambda: ops.patchify(img, patch, pad_cls_row=True))
t_gemm = timed(lambda: ops.gemm(ops.NT, patches, w, x, bias=bias, residual=pos, res_row_mod=1 + P, res_row_off=0))
t_fused = timed(lambda: ops.patch_embed_fwd(img, patch, w, bias, pos))
print(f"forward : patchify {t_pat:7.1f} us + GEMM {t_gemm:7.1f} us ({flops / t_gemm / 1e6:6.0f} TFLOP/s) = {t_pat + t_gemm:7.1f} us   |   fused {t_fused:7.1f} us ({flops / t_fused / 1e6:6.0f} TFLOP/s)")
t_wg = timed(lambda: XF._wgrad(dx, patches))
t_wf = timed(lambda: ops.patch_embed_wgrad(img, patch, dx))
print(f"wgrad   : TN GEMM on the stored patch matrix {t_wg:7.1f} us ({flops / t_wg / 1e6:6.0f} TFLOP/s)   |   fused {t_wf:7.1f} us ({flops / t_wf / 1e6:6.0f} TFLOP/s)")
print(f"patch matrix not stored: {patches.numel() * 2 / 1e9:.2f} GB resident, {patches.numel() * 2 * 3 / 1e9:.2f} GB of HBM traffic per step (write + 2 reads)")
