#!/usr/bin/env python3
"""Do two CU-masked streams run side by side?  Times a GEMM sequence on each masked stream alone and on both at once."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import cu_mask, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    M, N, K = 21546, 3072, 768
    mk = lambda: (torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16(), torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    sets = [mk(), mk()]
    plain = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    masked = [cu_mask.masked_stream(dev, bits) for bits in cu_mask.split_masks(256, 2)]

    def run(streams, which, iters=30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            for i in which:
                with torch.cuda.stream(streams[i]):
                    A, B, C = sets[i]
                    ops.gemm(ops.NT, A, B, C)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    for name, streams in (("plain", plain), ("masked", masked)):
        run(streams, [0, 1], 5)
        a, b, both = run(streams, [0]), run(streams, [1]), run(streams, [0, 1])
        print(f"{name:7s}: stream0 alone {a:7.1f} us/GEMM   stream1 alone {b:7.1f}   both (2 GEMMs) {both:7.1f}", flush=True)


if __name__ == "__main__":
    main()
