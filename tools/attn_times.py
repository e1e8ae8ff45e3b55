#!/usr/bin/env python3
"""Debug: per-wave cycle split of one attention-forward iteration (needs a -DXVIT_DEBUG_ATTN_TIMES build via XVIT_LIB)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for B, N in ((126, 513), (32, 512), (32, 513), (4, 4097)):
        H, d = 12, 768
        qkv = torch.randn(B, N, 3 * d, device=dev).bfloat16()
        for _ in range(3):
            o, lse = ops.attn_fwd(qkv.view(B * N, 3 * d), B, N, H, 0.125)
        torch.cuda.synchronize()
        nx = (N + 127) // 128
        nfull = N // 128        # x-blocks whose 4 waves are all active
        need = B * H * nx * 32
        assert lse.numel() >= need or True
        w = lse.flatten()[: need].view(B * H, nx, 4, 8)[:, :nfull].reshape(-1, 8).double().cpu()
        tot = w[:, :4].sum(1)
        print(f"B={B} N={N}: iteration 3, mean cycles per wave: wait+barrier {w[:,0].mean():6.0f}  QK^T {w[:,1].mean():6.0f}  softmax {w[:,2].mean():6.0f}  PV {w[:,3].mean():6.0f}"
              f"  total {tot.mean():6.0f} (min {tot.min():.0f} max {tot.max():.0f});  MFMA-only time would be 512")
        start = w[:, 4]
        start = (start - start.min()) * 0.01
        pro, loop = w[:, 5] * 0.01, w[:, 6] * 0.01
        ntile = (N + 63) // 64
        print(f"      per wave, us: prologue {pro.mean():5.2f} (max {pro.max():5.2f})  loop({ntile} tiles)+epilogue {loop.mean():6.2f} (min {loop.min():.2f} max {loop.max():.2f})"
              f"  -> {loop.mean() / ntile:5.3f} us per tile;  block starts: median {start.median():6.2f} us, last {start.max():6.2f} us")


if __name__ == "__main__" and "dkv" not in sys.argv:
    main()


def dkv_times():
    """Same for the dK/dV kernel (stamps land in the delta workspace)."""
    from xvit import _lib
    dev = torch.device("cuda:0")
    for B, N in ((126, 513), (32, 512), (32, 513), (4, 4097)):
        H, d, dh = 12, 768, 64
        qkv = torch.randn(B * N, 3 * d, device=dev).bfloat16()
        o, lse = ops.attn_fwd(qkv, B, N, H, 0.125)
        d_o = torch.randn(B * N, d, device=dev).bfloat16()
        dqkv = torch.empty_like(qkv)
        need = _lib.load().xvit_attn_bwd_workspace_bytes(B, H, N)
        delta = torch.zeros(need // 4, dtype=torch.float32, device=dev)
        p, g, ld = qkv.data_ptr(), dqkv.data_ptr(), 3 * d
        for _ in range(3):
            _lib.check(_lib.load().xvit_attn_bwd(p, p + 2 * d, p + 4 * d, N * ld, ld, o.data_ptr(), d_o.data_ptr(), N * d, d, lse.data_ptr(), delta.data_ptr(), need,
                                                 g, g + 2 * d, g + 4 * d, B, H, N, dh, 0.125, 0.0, 0, torch.cuda.current_stream().cuda_stream), "xvit_attn_bwd")
        torch.cuda.synchronize()
        nx, nfull = (N + 127) // 128, N // 128
        w = delta.flatten()[: B * H * nx * 32].view(B * H, nx, 4, 8)[:, :nfull].reshape(-1, 8).double().cpu()
        start = (w[:, 4] - w[:, 4].min()) * 0.01
        pro, loop = w[:, 5] * 0.01, w[:, 6] * 0.01
        ntile = (N + 63) // 64
        print(f"dKV B={B} N={N}: iteration 3 cycles per wave: wait+barrier {w[:,0].mean():6.0f}  compute {w[:,1].mean():6.0f}  (MFMA-only 1024; 2 waves/SIMD)"
              f"   prologue {pro.mean():5.2f} us (max {pro.max():5.2f})  loop({ntile} tiles) {loop.mean():6.2f} us -> {loop.mean() / ntile:5.3f} us per tile;"
              f"  block starts: median {start.median():6.2f}, last {start.max():6.2f} us")


if __name__ == "__main__" and "dkv" in sys.argv:
    dkv_times()
