// fp32 Linear for the single-token CLS path (reference model_cross.py:91 wq, :100 proj, :112-113 the 1-token FFN,
// :177-181 the heads): y[M, N] = x[M, K] W[N, K]^T (+bias) (GELU) (dropout) (+residual), fp32 operands, on the f32-input
// matrix instruction v_mfma_f32_32x32x2_f32 (exact fp32 fma chains, the fp32 vector rate).
//
// Why fp32: M is the batch (one CLS row per sample), <0.1 % of the model's FLOPs, but that one row per sample crosses
// ~10 Linear / LayerNorm stages on its way to 2 small logits.  With bf16 operands every stage adds ~1.6e-3 of storage
// rounding to a vector nothing averages over (a patch token's update is small against its residual stream; the CLS
// row's is not), which measured 5e-3 on the CLS rows and 1.2e-2 .. 2.4e-2 on the logits against the fp32 reference —
// for the bf16-emulating oracle exactly as for the kernels.  fp32 operands here cost two more bytes per weight on
// matrices that are streamed once per call, and nothing measurable in step time.
//
// Shape of the work: M <= a few hundred rows, N x K weights of 0.6 .. 2.4 M elements.  One WAVE per 32 x 32 output tile
// and K-chunk (no LDS, operands straight to registers: both are k-contiguous, a lane takes 16 bytes of its row per
// 8-deep step and the four MFMAs of the step pair the k's as (k0+e, k0+4+e) on BOTH operands); the K-chunks' partial
// tiles go to a caller-owned fp32 slab and a second kernel sums them in a fixed order and applies the epilogue
// (bit-reproducible, no atomics) — ~1000 waves keep every SIMD of the chip streaming the weight matrix.
#include "xvit_common.h"

namespace xvit {

struct F32Params {
  const float* A; const float* W; float* C; const float* bias; const float* res; float* slab;
  bf16* c_bf16; bf16* z_bf16;
  int64_t lda, ldw, ldc, ldr, ldcb, ldzb;
  int M, N, K, k_per_split, split_k, act;
  float drop_p, drop_inv; uint64_t drop_seed; const uint64_t* drop_epoch;
};

__device__ __forceinline__ void f32_epilogue(const F32Params& p, float v, int row, int col) {
  if (p.bias) v += p.bias[col];
  if (p.act == XVIT_ACT_GELU) {
    if (p.z_bf16) p.z_bf16[(int64_t)row * p.ldzb + col] = f2bf(v);
    v = gelu_f(v);
  }
  if (p.drop_p > 0.f) {   // the mask of xvit_dropout on a contiguous [M, N] tensor with this seed
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
    v = (hash32(drop_seed_at(p.drop_seed, p.drop_epoch), (uint64_t)row * p.N + col) & 0xFFFFFFu) >= thr ? v * p.drop_inv : 0.f;
  }
  if (p.res) v += p.res[(int64_t)row * p.ldr + col];
  p.C[(int64_t)row * p.ldc + col] = v;
  if (p.c_bf16) p.c_bf16[(int64_t)row * p.ldcb + col] = f2bf(v);
}

__global__ __launch_bounds__(64) void linear_f32_kernel(const F32Params p) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int ntn = (p.N + 31) >> 5;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn, split = blockIdx.y;
  const int m0 = tm * 32, n0 = tn * 32;
  const int kb = split * p.k_per_split, ke = min(p.K, kb + p.k_per_split);
  // rows past the end are clamped for the loads (their results are never stored)
  const float* ap = p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + 4 * h;
  const float* wp = p.W + (int64_t)min(n0 + r, p.N - 1) * p.ldw + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 2
  for (int k = kb; k < ke; k += 16) {   // k_per_split and K are multiples of 16
    const f32x4 a0 = *(const f32x4*)(ap + k), a1 = *(const f32x4*)(ap + k + 8);
    const f32x4 w0 = *(const f32x4*)(wp + k), w1 = *(const f32x4*)(wp + k + 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], w0[e], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], w1[e], acc, 0, 0, 0);
  }
  const int col = n0 + r;
  if (col >= p.N) return;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int row = m0 + (t & 3) + 8 * (t >> 2) + 4 * h;
    if (row < p.M) {
      if (p.slab) p.slab[((int64_t)split * p.M + row) * p.N + col] = acc[t];
      else f32_epilogue(p, acc[t], row, col);
    }
  }
}

__global__ void linear_f32_reduce_kernel(const F32Params p) {
  const int64_t total = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / p.N), col = (int)(i - (int64_t)row * p.N);
    float v = 0.f;
    for (int s = 0; s < p.split_k; ++s) v += p.slab[(int64_t)s * total + i];   // fixed order
    f32_epilogue(p, v, row, col);
  }
}

static int f32_split(int M, int N, int K) {
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  int split = 1024 / (tiles > 0 ? tiles : 1);      // ~1000 waves
  const int max_split = K / 64;                    // at least 4 sixteen-deep steps per wave
  if (split > max_split) split = max_split;
  if (split > 32) split = 32;
  return split < 1 ? 1 : split;
}

}  // namespace xvit

using namespace xvit;

extern "C" int64_t xvit_linear_f32_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int split = f32_split(M, N, K);
  return split > 1 ? (int64_t)split * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int xvit_linear_f32(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* y, int64_t ldy, int M, int N, int K,
                               int act, void* z_bf16, int64_t ldz, const float* residual, int64_t ldr, void* y_bf16, int64_t ldyb, float dropout_p,
                               uint64_t dropout_seed, void* workspace, int64_t workspace_bytes, xvit_stream_t stream) {
  XVIT_REQUIRE(x && W && y, "xvit_linear_f32: null x/W/y");
  XVIT_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, "xvit_linear_f32: need M, N > 0 and K a positive multiple of 16 (got %d, %d, %d)", M, N, K);
  XVIT_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K && ldy >= N, "xvit_linear_f32: ldx/ldw must be multiples of 4 and >= K, ldy >= N");
  XVIT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0, "xvit_linear_f32: x and W must be 16-byte aligned");
  XVIT_REQUIRE(act == XVIT_ACT_NONE || act == XVIT_ACT_GELU, "xvit_linear_f32: act must be NONE or GELU");
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "xvit_linear_f32: dropout_p must be in [0, 1)");
  XVIT_REQUIRE(!residual || ldr >= N, "xvit_linear_f32: ldr < N");
  XVIT_REQUIRE((!z_bf16 || ldz >= N) && (!y_bf16 || ldyb >= N), "xvit_linear_f32: ldz / ldyb < N");
  F32Params p;
  p.A = x; p.W = W; p.C = y; p.bias = bias; p.res = residual; p.c_bf16 = (bf16*)y_bf16; p.z_bf16 = (bf16*)z_bf16;
  p.lda = ldx; p.ldw = ldw; p.ldc = ldy; p.ldr = ldr; p.ldcb = ldyb; p.ldzb = ldz;
  p.M = M; p.N = N; p.K = K; p.act = act;
  p.split_k = f32_split(M, N, K);
  p.k_per_split = (((K + 15) / 16 + p.split_k - 1) / p.split_k) * 16;
  p.drop_p = dropout_p; p.drop_inv = 1.0f / (1.0f - dropout_p); p.drop_seed = dropout_seed; p.drop_epoch = dropout_p > 0.f ? drop_epoch_ptr() : nullptr;
  const int64_t need = xvit_linear_f32_workspace_bytes(M, N, K);
  XVIT_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "xvit_linear_f32: needs %lld bytes of workspace (got %lld)", (long long)need,
               (long long)workspace_bytes);
  p.slab = need > 0 ? (float*)workspace : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(((M + 31) / 32) * ((N + 31) / 32), p.split_k);
  hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(64), 0, s, p);
  if (p.slab) {
    const int64_t work = (int64_t)M * N;
    const int g = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    hipLaunchKernelGGL(linear_f32_reduce_kernel, dim3(g), dim3(256), 0, s, p);
  }
  return check_launch("xvit_linear_f32");
}
