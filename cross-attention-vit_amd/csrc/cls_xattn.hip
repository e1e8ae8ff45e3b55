// CLS-query cross-attention (reference model_cross.py:91-99): ONE query row per (batch, head)
// against N keys/values.  1xN per head is a GEMV, not an MFMA shape: the kernel is bound by the
// single pass over K and V (2*N*64*2 bytes per (b,h)), so everything is 16-byte coalesced loads
// straight to registers (8 lanes cover one 128-byte key row), wave shuffles and one LDS score row.
#include "xvit_common.h"

namespace xvit {

constexpr int XA_THREADS = 256;
constexpr int XA_DH = 64;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red /*[4]*/) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < XA_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// dot of this thread's 8 dims of a (loaded once) with the 8 dims of row n of mat; 8 lanes per row
__device__ __forceinline__ float row_dot8(const float (&a)[8], const bf16* mat, int64_t stride_n, int n, int part) {
  const bf16x8 kv = *(const bf16x8*)(mat + (int64_t)n * stride_n + part * 8);
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc = fmaf(a[e], bf2f(kv[e]), acc);
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  return acc;
}

// q_f32 / o_f32 (optional): the query in fp32 (used instead of q) and an fp32 copy of the output — the single-token CLS
// path keeps its operands in fp32 (see linear_f32.hip); o (bf16) is still written for the backward kernel.
__global__ __launch_bounds__(XA_THREADS) void cls_xattn_fwd_kernel(const bf16* __restrict__ q, int64_t ldq, const float* __restrict__ q_f32, int64_t ldqf,
                                                                   const bf16* __restrict__ k, const bf16* __restrict__ v, int64_t sb, int64_t sn,
                                                                   bf16* __restrict__ o, int64_t ldo, float* __restrict__ o_f32, int64_t ldof,
                                                                   float* __restrict__ p, int H, int N, float scale, float drop_p,
                                                                   uint64_t drop_seed_in, const uint64_t* __restrict__ drop_epoch) {
  const uint64_t drop_seed = drop_seed_at(drop_seed_in, drop_epoch);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = (float*)smem_raw;            // [N] scores -> probabilities
  float* red = sc + ((N + 3) & ~3);        // [4] + [32][64] partial outputs
  float* part_o = red + 4;
  const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int part = tid & 7, slice = tid >> 3;  // 32 row slices x 8 lanes
  const bf16* kb = k + (int64_t)b * sb + head * XA_DH;
  const bf16* vb = v + (int64_t)b * sb + head * XA_DH;

  float qv[8];
  if (q_f32) {
    const float* t = q_f32 + (int64_t)b * ldqf + head * XA_DH + part * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = t[e];
  } else {
    const bf16x8 t = *(const bf16x8*)(q + (int64_t)b * ldq + head * XA_DH + part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = bf2f(t[e]);
  }
  float mx = -INFINITY;
  for (int n = slice; n < N; n += 32) {
    const float s = row_dot8(qv, kb, sn, n, part) * scale;
    if (part == 0) sc[n] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_reduce(mx, true, red);
  float sum = 0.f;
  for (int n = tid; n < N; n += XA_THREADS) {
    const float e = __expf(sc[n] - mx);
    sc[n] = e;
    sum += e;
  }
  sum = block_reduce(sum, false, red);
  const float inv = 1.0f / sum;
  float* prow = p + ((int64_t)b * H + head) * N;
  const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);
  const float dinv = 1.0f / (1.0f - drop_p);
  const uint64_t pidx = ((uint64_t)b * H + head) * N;
  for (int n = tid; n < N; n += XA_THREADS) {
    const float pr = sc[n] * inv;
    prow[n] = pr;                                   // saved pre-dropout (backward regenerates the mask)
    const bool keep = drop_p <= 0.f || (hash32(drop_seed, pidx + n) & 0xFFFFFFu) >= thr;
    sc[n] = keep ? pr * (drop_p > 0.f ? dinv : 1.0f) : 0.f;
  }
  __syncthreads();
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = slice; n < N; n += 32) {
    const bf16x8 vv = *(const bf16x8*)(vb + (int64_t)n * sn + part * 8);
    const float pr = sc[n];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(pr, bf2f(vv[e]), acc[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part_o[slice * 64 + part * 8 + e] = acc[e];
  __syncthreads();
  if (tid < 64) {
    float r = 0.f;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) r += part_o[s * 64 + tid];
    if (o) o[(int64_t)b * ldo + head * XA_DH + tid] = f2bf(r);
    if (o_f32) o_f32[(int64_t)b * ldof + head * XA_DH + tid] = r;
  }
}

__global__ __launch_bounds__(XA_THREADS) void cls_xattn_bwd_kernel(const bf16* __restrict__ q, int64_t ldq, const bf16* __restrict__ k,
                                                                   const bf16* __restrict__ v, int64_t sb, int64_t sn,
                                                                   const float* __restrict__ p, const bf16* __restrict__ d_o, int64_t lddo,
                                                                   float* __restrict__ dq, int64_t lddq, bf16* __restrict__ dk,
                                                                   bf16* __restrict__ dv, float* __restrict__ coef, int H, int N, float scale,
                                                                   float drop_p, uint64_t drop_seed_in, const uint64_t* __restrict__ drop_epoch) {
  const uint64_t drop_seed = drop_seed_at(drop_seed_in, drop_epoch);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* ds = (float*)smem_raw;            // [N] dp -> ds
  float* red = ds + ((N + 3) & ~3);
  float* part_q = red + 4;                 // [32][64]
  const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int part = tid & 7, slice = tid >> 3;
  const int64_t boff = (int64_t)b * sb + head * XA_DH;
  const float* prow = p + ((int64_t)b * H + head) * N;

  float qv[8], gov[8];
  {
    const bf16x8 t = *(const bf16x8*)(q + (int64_t)b * ldq + head * XA_DH + part * 8);
    const bf16x8 g = *(const bf16x8*)(d_o + (int64_t)b * lddo + head * XA_DH + part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { qv[e] = bf2f(t[e]); gov[e] = bf2f(g[e]); }
  }
  // p' = mask * p / (1-pd);  dp[n] = mask/(1-pd) * (do . v[n]);   dsum = sum_n p[n] dp[n]
  const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);
  const float dinv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t pidx = ((uint64_t)b * H + head) * N;
  float dsum = 0.f;
  for (int n = slice; n < N; n += 32) {
    const float mk = (drop_p <= 0.f || (hash32(drop_seed, pidx + n) & 0xFFFFFFu) >= thr) ? dinv : 0.f;
    const float dp = row_dot8(gov, v + boff, sn, n, part) * mk;
    if (part == 0) { ds[n] = dp; dsum += prow[n] * dp; }
  }
  dsum = block_reduce(dsum, false, red);
  // ds[n] = p (dp - dsum);  dq += scale * ds[n] k[n];  dk[n] = scale * ds[n] q;  dv[n] = p[n] do
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = slice; n < N; n += 32) {
    const float mk = (drop_p <= 0.f || (hash32(drop_seed, pidx + n) & 0xFFFFFFu) >= thr) ? dinv : 0.f;
    const float pr0 = prow[n];
    const float pr = pr0 * mk;                       // dropped probability feeds dv
    const float dsn = pr0 * (ds[n] - dsum) * scale;
    const bf16x8 kk = *(const bf16x8*)(k + boff + (int64_t)n * sn + part * 8);
    bf16x8 okk, ovv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[e] = fmaf(dsn, bf2f(kk[e]), acc[e]);
      okk[e] = f2bf(dsn * qv[e]);
      ovv[e] = f2bf(pr * gov[e]);
    }
    if (dk) {
      *(bf16x8*)(dk + boff + (int64_t)n * sn + part * 8) = okk;
      *(bf16x8*)(dv + boff + (int64_t)n * sn + part * 8) = ovv;
    }
    // dK and dV of this (b, head) are rank one — dk[n] = dsn q, dv[n] = pr dO: the two coefficients are all the K/V path's
    // backward needs (xattn_kv_dgrad_kernel below), coef[b][n][head] = dsn, coef[b][n][H + head] = pr
    if (coef && part == 0) {
      float* c = coef + ((int64_t)b * N + n) * (2 * H);
      c[head] = dsn;
      c[H + head] = pr;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part_q[slice * 64 + part * 8 + e] = acc[e];
  __syncthreads();
  if (tid < 64) {
    float r = 0.f;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) r += part_q[s * 64 + tid];
    dq[(int64_t)b * lddq + head * XA_DH + tid] = r;
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The K/V path's backward in its low-rank form.  For one sample, dkv[n, :] = sum_h (dsn[n,h] q_h | pr[n,h] dO_h) has rank <= 2 H,
// so the gradient of the normed tokens,  dhn[n, :] = dk[n, :] Wk + dv[n, :] Wv = sum_j coef[n, j] R[j, :]  with
// R[h, :] = q_h Wk[64 h .. 64 h + 63, :],  R[H + h, :] = dO_h Wv[64 h .., :]   (2 H x d, fp32, per sample: 2 H small products the host runs),
// and the weight gradient,  dWk[64 h + e, :] = sum_b q[b,h,e] T[b, h, :],  T[b, j, :] = sum_n coef[b, n, j] hn[b, n, :],
// need no [B N, 2 d] tensor, no K = 2 d GEMMs (2 x 152 GFLOP per fusion at configs[1]) and no column-sum pass: one kernel writes
// dhn (99 MB), one reads hn (99 MB), both with 2 H = 24 fma per element.  Block = (row slice, sample); a thread owns 4 columns.
// ------------------------------------------------------------------------------------------------------------------
constexpr int XKV_MAXJ = 32;    // 2 H <= 32

constexpr int XKV_ROWS = 64;    // rows of coefficients staged in LDS at a time (64 x 32 floats = 8 KiB)

// stage coef[b][n0 .. n0 + rows)[0 .. J) into LDS rows of J2 floats (zero padded): every thread then reads a row's coefficients
// as broadcast ds_read_b128 instead of 2 H global loads
template <int J2>
__device__ __forceinline__ void xkv_stage(float* lds, const float* __restrict__ coef, int64_t row0, int rows, int J) {
  for (int i = threadIdx.x; i < rows * J2; i += blockDim.x) {
    const int r = i / J2, j = i - r * J2;
    lds[i] = j < J ? coef[(row0 + r) * J + j] : 0.f;
  }
}

template <int J2>   // J2 = 2 H rounded up to a multiple of 8 (register-array bound)
__global__ __launch_bounds__(256) void xattn_kv_dgrad_kernel(const float* __restrict__ coef, const float* __restrict__ R /*[2 H][B][d]*/,
                                                             bf16* __restrict__ dhn, int64_t lddh, int H, int N, int d, int rows_per_block) {
  __shared__ __attribute__((aligned(16))) float cl[XKV_ROWS * J2];
  const int b = blockIdx.y, tid = threadIdx.x, J = 2 * H;
  const int n0 = blockIdx.x * rows_per_block, n1 = min(N, n0 + rows_per_block);
  const int c0 = 4 * tid;                      // d <= 1024: one 4-column group per thread (host-checked)
  f32x4 r[J2];
#pragma unroll
  for (int j = 0; j < J2; ++j) r[j] = (j < J && c0 < d) ? *(const f32x4*)(R + ((int64_t)j * gridDim.y + b) * d + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int nb = n0; nb < n1; nb += XKV_ROWS) {
    const int rows = min(XKV_ROWS, n1 - nb);
    __syncthreads();
    xkv_stage<J2>(cl, coef, (int64_t)b * N + nb, rows, J);
    __syncthreads();
    if (c0 < d) {
      for (int rr = 0; rr < rows; ++rr) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j4 = 0; j4 < J2; j4 += 4) {
          const f32x4 cf = *(const f32x4*)(cl + rr * J2 + j4);   // same address on every lane: an LDS broadcast
          o += cf[0] * r[j4] + cf[1] * r[j4 + 1] + cf[2] * r[j4 + 2] + cf[3] * r[j4 + 3];
        }
        const bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
        *(bf16x4*)(dhn + ((int64_t)b * N + nb + rr) * lddh + c0) = ob;
      }
    }
  }
}

}  // namespace xvit

using namespace xvit;

static size_t xa_lds(int N) { return (size_t)(((N + 3) & ~3) + 4 + 32 * 64) * sizeof(float); }

extern "C" int xvit_cls_xattn_fwd(const void* q, int64_t ldq, const float* q_f32, int64_t ldqf, const void* k, const void* v, int64_t sb, int64_t sn,
                                  void* o, int64_t ldo, float* o_f32, int64_t ldof, float* p, int B, int H, int N, int dh, float scale, float drop_p,
                                  uint64_t drop_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "xvit_cls_xattn_fwd: dropout_p must be in [0, 1)");
  XVIT_REQUIRE((q || q_f32) && k && v && (o || o_f32) && p, "xvit_cls_xattn_fwd: null pointer");
  XVIT_REQUIRE(dh == XA_DH, "xvit_cls_xattn_fwd: head dim %d unsupported (only 64)", dh);
  XVIT_REQUIRE(B > 0 && H > 0 && N > 0 && B <= 65535, "xvit_cls_xattn_fwd: bad B/H/N");
  XVIT_REQUIRE((!q || ldq % 8 == 0) && sb % 8 == 0 && sn % 8 == 0, "xvit_cls_xattn_fwd: strides must be multiples of 8 elements");
  XVIT_REQUIRE(xa_lds(N) <= 160 * 1024, "xvit_cls_xattn_fwd: N=%d too long for the LDS score row", N);
  const size_t lds = xa_lds(N);
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    (void)hipFuncSetAttribute((const void*)cls_xattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  hipLaunchKernelGGL(cls_xattn_fwd_kernel, dim3(H, B), dim3(XA_THREADS), lds, (hipStream_t)stream, (const bf16*)q, ldq, q_f32, ldqf, (const bf16*)k,
                     (const bf16*)v, sb, sn, (bf16*)o, ldo, o_f32, ldof, p, H, N, scale, drop_p, drop_seed,
                     drop_p > 0.f ? drop_epoch_ptr() : nullptr);
  return check_launch("xvit_cls_xattn_fwd");
}

extern "C" int xvit_cls_xattn_bwd(const void* q, int64_t ldq, const void* k, const void* v, int64_t sb, int64_t sn, const float* p, const void* d_o,
                                  int64_t lddo, float* dq, int64_t lddq, void* dk, void* dv, float* coef, int B, int H, int N, int dh, float scale,
                                  float drop_p, uint64_t drop_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "xvit_cls_xattn_bwd: dropout_p must be in [0, 1)");
  XVIT_REQUIRE(q && k && v && p && d_o && dq && ((dk && dv) || coef) && (!dk == !dv), "xvit_cls_xattn_bwd: null pointer (dk and dv together, or coef)");
  XVIT_REQUIRE(dh == XA_DH, "xvit_cls_xattn_bwd: head dim %d unsupported (only 64)", dh);
  XVIT_REQUIRE(B > 0 && H > 0 && N > 0 && B <= 65535, "xvit_cls_xattn_bwd: bad B/H/N");
  XVIT_REQUIRE(ldq % 8 == 0 && lddo % 8 == 0 && sb % 8 == 0 && sn % 8 == 0, "xvit_cls_xattn_bwd: strides must be multiples of 8 elements");
  XVIT_REQUIRE(xa_lds(N) <= 160 * 1024, "xvit_cls_xattn_bwd: N=%d too long for the LDS score row", N);
  const size_t lds = xa_lds(N);
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    (void)hipFuncSetAttribute((const void*)cls_xattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  hipLaunchKernelGGL(cls_xattn_bwd_kernel, dim3(H, B), dim3(XA_THREADS), lds, (hipStream_t)stream, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, sb,
                     sn, p, (const bf16*)d_o, lddo, dq, lddq, (bf16*)dk, (bf16*)dv, coef, H, N, scale, drop_p, drop_seed,
                     drop_p > 0.f ? drop_epoch_ptr() : nullptr);
  return check_launch("xvit_cls_xattn_bwd");
}

static int xkv_slices(int B, int N, int target = 1024) {   // row slices per sample: ~target blocks, at least 32 rows each
  int s = (target + B - 1) / B;
  if (s > (N + 31) / 32) s = (N + 31) / 32;
  return s < 1 ? 1 : s;
}

extern "C" int xvit_xattn_kv_dgrad(const float* coef, const float* R, void* dhn, int64_t lddh, int B, int H, int N, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(coef && R && dhn, "xvit_xattn_kv_dgrad: null pointer");
  XVIT_REQUIRE(B > 0 && B <= 65535 && N > 0 && H > 0 && 2 * H <= XKV_MAXJ && d % 4 == 0 && d <= 1024, "xvit_xattn_kv_dgrad: need 2 H <= %d, d %% 4 == 0 and d <= 1024 (B=%d H=%d N=%d d=%d)", XKV_MAXJ, B, H, N, d);
  XVIT_REQUIRE(lddh % 4 == 0 && lddh >= d && ((uintptr_t)R & 15) == 0 && ((uintptr_t)dhn & 7) == 0, "xvit_xattn_kv_dgrad: lddh must be a multiple of 4 and >= d, pointers aligned");
  const int slices = xkv_slices(B, N), rpb = (N + slices - 1) / slices;
  const dim3 grid(slices, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (2 * H <= 8) hipLaunchKernelGGL((xattn_kv_dgrad_kernel<8>), grid, block, 0, s, coef, R, (bf16*)dhn, lddh, H, N, d, rpb);
  else if (2 * H <= 16) hipLaunchKernelGGL((xattn_kv_dgrad_kernel<16>), grid, block, 0, s, coef, R, (bf16*)dhn, lddh, H, N, d, rpb);
  else if (2 * H <= 24) hipLaunchKernelGGL((xattn_kv_dgrad_kernel<24>), grid, block, 0, s, coef, R, (bf16*)dhn, lddh, H, N, d, rpb);
  else hipLaunchKernelGGL((xattn_kv_dgrad_kernel<32>), grid, block, 0, s, coef, R, (bf16*)dhn, lddh, H, N, d, rpb);
  return check_launch("xvit_xattn_kv_dgrad");
}
