#!/usr/bin/env python3
"""Every GEMM of one training step at configs[1] with its REAL epilogue (bias / GELU + z / GELU' / fp32 residual / split-K
wgrad), timed per tile kernel, variants interleaved in one process (HIP events, random bf16 operands).  Prints the time,
the algorithmic TFLOP/s and the algorithmic HBM GB/s (compulsory operand + epilogue bytes) of every case.

    python tools/gemm_model_bench.py [batch] [variant ...]     variants: auto narrow tile128 tile256 auxz g1..g12
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
from xvit import ops  # noqa: E402
from xvit.functional import _wgrad_split  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 126
T = B * 513


AUX = [1]     # xvit_gemm aux_mode of the GELU / GELU' epilogues: 1 = the saved tensor is gelu'(z) (what the model uses), 0 = z (variant "auxz")


def bf(*shape):
    return torch.randn(*shape, device=dev).bfloat16()


def cases():
    d, f = 768, 3072
    out = []

    def nt(label, M, N, K, bias=True, res=False, gelu=False, f32=False):
        A, W = bf(M, K), bf(N, K)
        C = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        kw = {}
        byt = M * K * 2 + N * K * 2 + M * N * (4 if f32 else 2)
        if bias:
            kw["bias"] = torch.randn(N, device=dev)
        if res:
            kw["residual"] = torch.randn(M, N, device=dev); byt += M * N * 4
        if gelu:
            kw["act"] = ops.ACT_GELU; kw["aux"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev); byt += M * N * 2
        out.append((label, lambda: ops.gemm(ops.NT, A, W, C, **kw, **({"aux_mode": AUX[0]} if gelu else {})), 2.0 * M * N * K, byt))

    def nn(label, M, N, K, dgelu=False):
        A, W = bf(M, K), bf(K, N)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw = {}
        byt = M * K * 2 + N * K * 2 + M * N * 2
        if dgelu:
            kw["act"] = ops.ACT_DGELU; kw["aux"] = bf(M, N); kw["colsum"] = torch.zeros(N, device=dev); byt += M * N * 2
        out.append((label, lambda: ops.gemm(ops.NN, A, W, C, **kw, **({"aux_mode": AUX[0]} if dgelu else {})), 2.0 * M * N * K, byt))

    def tn(label, M, N, K):
        A, Bm = bf(K, M), bf(K, N)
        C = torch.empty(M, N, dtype=torch.float32, device=dev)
        sp = _wgrad_split(M, N, K)
        out.append((f"{label} s{sp}", lambda: ops.gemm(ops.TN, A, Bm, C, split_k=sp), 2.0 * M * N * K, K * (M + N) * 2 + M * N * 4))

    nt("qkv", T, 3 * d, d, bias=False)
    nt("out-proj +res f32", T, d, d, res=True, f32=True)
    nt("ffn1 +gelu +z", T, f, d, gelu=True)
    nt("ffn2 +res f32", T, d, f, res=True, f32=True)
    nt("kv", T, 2 * d, d)
    nt("patch-embed f32", 2 * T, d, 4096, res=True, f32=True)
    nn("d-ffn2 dgelu", T, f, d, dgelu=True)
    nn("d-ffn1", T, d, f)
    nn("d-out", T, d, d)
    nn("d-qkv", T, d, 3 * d)
    tn("w-ffn2", d, f, T)
    tn("w-ffn1", f, d, T)
    tn("w-out", d, d, T)
    tn("w-qkv", 3 * d, d, T)
    tn("w-patch", d, 4096, 2 * T)
    return out


VARIANTS = {"auto": (("gemm_tile", 0), ("gemm_epilogue", 0), ("gemm_group", 0)), "auxz": (("gemm_tile", 0), ("gemm_epilogue", 0), ("gemm_group", 0)), "narrow": (("gemm_tile", 0), ("gemm_epilogue", 1)), "tile128": (("gemm_tile", 1), ("gemm_epilogue", 0)), "tile256": (("gemm_tile", 2), ("gemm_epilogue", 0)),
            **{f"g{n}": (("gemm_tile", 0), ("gemm_epilogue", 0), ("gemm_group", n)) for n in (1, 2, 3, 4, 6, 12)}}


def main():
    names = [a for a in sys.argv[1:] if a in VARIANTS] or ["auto", "narrow"]
    iters, rounds = int(os.environ.get("ITERS", "6")), int(os.environ.get("ROUNDS", "3"))
    weights = {"qkv": 8, "out-proj": 8, "ffn1": 8, "ffn2 ": 8, "kv": 4, "patch": 1, "d-ffn2": 8, "d-ffn1": 8, "d-out": 8, "d-qkv": 8, "w-ffn2": 8, "w-ffn1": 8, "w-out": 8, "w-qkv": 8, "w-patch": 1}
    total = {v: 0.0 for v in names}
    print(f"B={B} T={T}  variants: {names}  (us | TFLOP/s | GB/s algorithmic)")
    for label, fn, flops, byt in cases():
        best = {v: 1e30 for v in names}
        for _ in range(rounds):
            for v in names:
                for k_, v_ in VARIANTS[v]:
                    ops.set_option(k_, v_)
                AUX[0] = 0 if v == "auxz" else 1
                fn(); fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn()
                e.record(); torch.cuda.synchronize()
                best[v] = min(best[v], s.elapsed_time(e) / iters * 1e3)
        w = next((n for k, n in weights.items() if label.startswith(k)), 1)
        for v in names:
            total[v] += best[v] * w
        ideal = max(flops / 2516e6, byt / 6.3e6)
        print(f"{label:22s} x{w}  ideal {ideal:6.0f} us | " + " | ".join(f"{v}: {best[v]:7.1f} us {flops / best[v] / 1e6:6.0f} TF {byt / best[v] / 1e3:5.0f} GB/s" for v in names), flush=True)
    print("per-step GEMM time (weighted by launches per step, ms): " + "  ".join(f"{v}: {total[v] / 1e3:.2f}" for v in names))
    ops.set_option("gemm_tile", 0); ops.set_option("gemm_epilogue", 0); ops.set_option("gemm_group", 0)


if __name__ == "__main__":
    main()
