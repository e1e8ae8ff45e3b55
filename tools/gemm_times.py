#!/usr/bin/env python3
"""Debug: per-block phase timestamps of the big-tile GEMM (needs a -DXVIT_DEBUG_TIMES build, XVIT_LIB=...)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, N = 16384, 2048
    for K in (64, 768, 3072):
        A, B = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        nblk = (M // 256) * (N // 256)
        ts = torch.zeros(nblk * 16 + nblk * 8 * 8, dtype=torch.float32, device=dev)   # 8 x uint64 per block + 4 x uint64 per wave
        for _ in range(3):
            ops.gemm(ops.NT, A, B, C, colsum=ts)
        torch.cuda.synchronize()
        raw = ts.view(torch.int64).cpu()
        t = raw[:nblk * 8].view(nblk, 8).double() * 0.01   # 100 MHz -> us
        t0 = t[:, 0].min()
        start, landed, loop, setup, end = [(t[:, i] - t0) for i in range(5)]
        first = start < 0.5 * end.max()   # blocks of the first round
        def st(x, m): return f"{x[m].mean():6.2f} (min {x[m].min():6.2f} max {x[m].max():6.2f})"
        print(f"K={K}: kernel span {end.max():.2f} us; per-block phases in us (mean over blocks):")
        for name, m in (("round 1", first), ("round 2", ~first)):
            print(f"  {name}: n={int(m.sum())} start {st(start, m)}  first-tile wait {st(landed - start, m)}  main loop {st(loop - landed, m)}"
                  f"  epi setup {st(setup - loop, m)}  epi bodies {st(end - setup, m)}")
        w = raw[nblk * 8:].view(nblk * 8, 4).double()
        if K > 320:
            print(f"  iteration 4, per wave, shader cycles: DMA issue {w[:, 0].mean():7.0f} (max {w[:, 0].max():6.0f})   MFMA section {w[:, 1].mean():7.0f} (max {w[:, 1].max():6.0f})"
                  f"   wait+barrier at next top {w[:, 2].mean():7.0f} (max {w[:, 2].max():6.0f})")


if __name__ == "__main__":
    main()
