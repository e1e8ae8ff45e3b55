"""GPU parity: xvit_gemm (all layouts + epilogues) against fp32 matmul on the same bf16 operands."""
import math

import pytest
import torch

from _util import assert_close, dev, randn, rt

pytestmark = pytest.mark.gpu


def _ops():
    from xvit import ops
    return ops


@pytest.fixture(autouse=True, params=[(0, 0), (1, 0), (0, 1)], ids=["auto", "tile128", "narrow-epilogue"])
def gemm_variant(request):
    """Every case runs under the automatic choices, with the 128x128 kernel forced, and with the 8-byte-per-lane epilogue
    forced (bf16 outputs of the 256x256 kernel normally take the 16-byte-per-lane epilogue) — xvit_set_option."""
    tile, epi = request.param
    _ops().set_option("gemm_tile", tile)
    _ops().set_option("gemm_epilogue", epi)
    yield request.param
    _ops().set_option("gemm_tile", 0)
    _ops().set_option("gemm_epilogue", 0)


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


SHAPES = [(64, 128, 64), (200, 192, 128), (1026, 768, 768), (130, 576, 192), (513, 2304, 768), (33, 3072, 768), (2, 768, 768),
          (256, 256, 64), (1300, 1000, 320), (2052, 768, 3072),
          (4100, 2304, 128), (2600, 3072, 64), (9000, 1536, 192), (33000, 768, 64)]  # > 128 tiles of 256x256: the big-tile kernel (the first three
                                                                                    # with >= 6 column tiles: super-column tile walk, ragged last group)


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("out_f32", [False, True])
def test_nt(M, N, K, out_f32):
    ops = _ops()
    a, w = rt(randn(M, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5))
    C = torch.full((M, N), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C)
    assert_close(C, a @ w.T, f"NT {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(200, 192, 128), (1026, 768, 3072), (513, 768, 2304), (17, 192, 576)])
def test_nn_dgrad(M, N, K):
    """dx[M,N] = dy[M,K] @ W[K,N]  (W stored [out=K, in=N], n contiguous)."""
    ops = _ops()
    dy, w = rt(randn(M, K, seed=3)), rt(randn(K, N, seed=4, scale=K ** -0.5))
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.gemm(ops.NN, dy.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C)
    assert_close(C, dy @ w, f"NN {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,split", [(192, 128, 200, 1), (768, 768, 1026, 1), (768, 3072, 1026, 3), (2304, 768, 513, 2),
                                         (576, 192, 34, 1), (768, 768, 4104, 8), (128, 4096, 1024, 2), (304, 264, 130, 3), (768, 768, 330, 4)])
def test_tn_wgrad(M, N, K, split):
    """dW[M,N] = dy[K,M]^T @ x[K,N]: the contraction (tokens) is the row index of both operands and
    need not be a multiple of anything (zero-filled by the DMA)."""
    ops = _ops()
    dy, x = rt(randn(K, M, seed=5)), rt(randn(K, N, seed=6))
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())   # split-K must not depend on C's contents
    ops.gemm(ops.TN, dy.to(dev(), torch.bfloat16), x.to(dev(), torch.bfloat16), C, split_k=split)
    assert_close(C, dy.T @ x, f"TN {M}x{N}x{K} split{split}")
    # accumulate on top (beta = 1)
    ops.gemm(ops.TN, dy.to(dev(), torch.bfloat16), x.to(dev(), torch.bfloat16), C, split_k=split, accumulate=True)
    assert_close(C, 2 * (dy.T @ x), f"TN accumulate {M}x{N}x{K} split{split}")


@pytest.mark.parametrize("M,N,K", [(513, 768, 192), (2600, 3072, 128), (4100, 2560, 64)])
def test_epilogue_bias_gelu_aux(M, N, K):
    ops = _ops()
    a, w, b = rt(randn(M, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5)), randn(N, seed=3, scale=0.1)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    Z = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, bias=b.to(dev()), act=ops.ACT_GELU, aux=Z)
    z = a @ w.T + b
    assert_close(Z, z, "pre-activation")
    assert_close(C, _gelu(z), "gelu")


@pytest.mark.parametrize("M,N,K,drop", [(513, 768, 192, 0.0), (2600, 3072, 128, 0.0), (4100, 2560, 64, 0.0), (33000, 3072, 64, 0.0), (2600, 3072, 128, 0.25), (130, 192, 64, 0.0)])
def test_gelu_saves_its_derivative_and_dgrad_multiplies(M, N, K, drop):
    """aux_mode 1 (what the FFNs use): the GELU epilogue writes gelu'(z) instead of z, the dgrad epilogue multiplies by it —
    on the wide 256x256 kernels (no dropout), the narrow ones (dropout) and the 128x128 kernel (small shapes)."""
    ops = _ops()
    a, w, b = rt(randn(M, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5)), randn(N, seed=3, scale=0.1)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    D = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    dp = (drop, 1234) if drop > 0 else None
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, bias=b.to(dev()), act=ops.ACT_GELU, aux=D, aux_mode=1, dropout=dp)
    z = a @ w.T + b
    assert_close(D, _dgelu(z), "saved gelu'")
    if drop == 0.0:
        assert_close(C, _gelu(z), "gelu")
    # backward: dz = (dy W2) * saved derivative, with the column sums of the result
    F = 256
    dy, w2 = rt(randn(M, F, seed=4)), rt(randn(F, N, seed=5, scale=F ** -0.5))
    dz = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    cs = torch.zeros(N, device=dev())
    ops.gemm(ops.NN, dy.to(dev(), torch.bfloat16), w2.to(dev(), torch.bfloat16), dz, act=ops.ACT_DGELU, aux=D, aux_mode=1, colsum=cs)
    ref = (dy @ w2) * D.float().cpu()
    assert_close(dz, ref, "dgrad x saved derivative")
    assert_close(cs, ref.sum(0), "colsum")


def test_epilogue_dgelu():
    ops = _ops()
    M, N, K = 260, 192, 768
    dy, w, z = rt(randn(M, K, seed=1)), rt(randn(K, N, seed=2, scale=K ** -0.5)), rt(randn(M, N, seed=3))
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(ops.NN, dy.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, act=ops.ACT_DGELU, aux=z.to(dev(), torch.bfloat16))
    assert_close(C, (dy @ w) * _dgelu(z), "dgelu")


@pytest.mark.parametrize("M,split", [(1026, 1), (3000, 1), (21546, 1), (32, 1), (32, 6), (300, 3)])
def test_epilogue_dgelu_colsum_and_split(M, split):
    """GELU' epilogue + column sums of the result (bias gradient) on every path: the 256x256 tile kernel (ragged
    last row tile included), the 128x128 kernel, and the split-K second pass (small-M GEMMs are split to hide
    their serial K loop)."""
    ops = _ops()
    N, K = 3072, 768
    dy, w, z = rt(randn(M, K, seed=1)), rt(randn(K, N, seed=2, scale=K ** -0.5)), rt(randn(M, N, seed=3))
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    cs = torch.full((N,), 0.25, device=dev())          # the kernel accumulates: a non-zero start value must survive
    ops.gemm(ops.NN, dy.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, act=ops.ACT_DGELU, aux=z.to(dev(), torch.bfloat16),
             colsum=cs, split_k=split)
    ref = (dy @ w) * _dgelu(z)
    assert_close(C, ref, "dgelu")
    assert_close(cs - 0.25, ref.sum(0), "colsum of the epilogue output")


@pytest.mark.parametrize("M,N,K,split", [(32, 768, 3072, 8), (32, 3072, 768, 3), (130, 192, 1024, 4)])
def test_split_k_with_full_epilogue(M, N, K, split):
    ops = _ops()
    a, w, b, r = rt(randn(M, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5)), randn(N, seed=3), randn(M, N, seed=4)
    C = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, bias=b.to(dev()), residual=r.to(dev()), split_k=split)
    assert_close(C, a @ w.T + b + r, "split-K bias+residual")
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    Z = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), Cb, bias=b.to(dev()), act=ops.ACT_GELU, aux=Z, split_k=split)
    assert_close(Z, a @ w.T + b, "split-K pre-activation"); assert_close(Cb, _gelu(a @ w.T + b), "split-K gelu")


def test_epilogue_bias_residual_f32():
    ops = _ops()
    M, N, K = 1026, 768, 768
    a, w, b, r = rt(randn(M, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5)), randn(N, seed=3), randn(M, N, seed=4)
    C = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, bias=b.to(dev()), residual=r.to(dev()))
    assert_close(C, a @ w.T + b + r, "bias+residual")


def test_epilogue_patch_embed_rows():
    """pos broadcast by row modulo + the row remap that leaves the CLS row free (model_cross.py:194-197)."""
    ops = _ops()
    Bm, P, K, N = 3, 16, 128, 192
    a, w, b = rt(randn(Bm * P, K, seed=1)), rt(randn(N, K, seed=2, scale=K ** -0.5)), randn(N, seed=3)
    pos = randn(P + 1, N, seed=4)
    X = torch.full((Bm * (P + 1), N), 7.0, dtype=torch.float32, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), X, bias=b.to(dev()), residual=pos.to(dev()),
             res_row_mod=P, res_row_off=1, out_seg=(P, 1, 1), M=Bm * P)
    ref = torch.full((Bm, P + 1, N), 7.0)
    ref[:, 1:] = (a @ w.T + b).reshape(Bm, P, N) + pos[1:]
    assert_close(X.reshape(Bm, P + 1, N), ref, "patch-embed epilogue")


def test_batched_and_strided_views():
    ops = _ops()
    G, M, N, K = 2, 130, 192, 256
    a, w = rt(randn(G, M, K, seed=1)), rt(randn(G, N, K, seed=2, scale=K ** -0.5))
    bias = randn(G, N, seed=3)
    C = torch.empty(G, M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(ops.NT, a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), C, bias=bias.to(dev()))
    assert_close(C, torch.einsum("gmk,gnk->gmn", a, w) + bias[:, None], "batched NT")
    # column-sliced operand (lda > K), e.g. rows of a [B*N, 3d] tensor
    big = rt(randn(M, 3 * K, seed=5))
    bd = big.to(dev(), torch.bfloat16)
    C2 = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm(ops.NT, bd[:, K:2 * K], w[0].to(dev(), torch.bfloat16), C2)
    assert_close(C2, big[:, K:2 * K] @ w[0].T, "strided A")


def test_rejects_bad_arguments():
    ops = _ops()
    a = torch.zeros(64, 100, dtype=torch.bfloat16, device=dev())  # K=100 not a multiple of 64
    w = torch.zeros(64, 100, dtype=torch.bfloat16, device=dev())
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(ops.NT, a[:, :96], w[:, :96], torch.empty(64, 64, dtype=torch.float32, device=dev()))
    with pytest.raises(RuntimeError):
        ops.gemm(ops.NT, torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 4))  # CPU tensors
