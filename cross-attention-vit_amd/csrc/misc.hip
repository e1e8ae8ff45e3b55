// HBM-bound helpers around the GEMMs: 3-D patchify, CLS/pos rows, column sums (bias grads),
// casts, dropout, the num_classes head and the mean + cross-entropy tail.
#include "xvit_common.h"

namespace xvit {

// ------------------------------------------------------------------------------------------
// patchify (reference model_cross.py:193).  Threads walk the INPUT in memory order (8 voxels =
// one 16/32-byte run each), so reads are fully coalesced; a wave then covers 4 H-rows of 8
// patches and its writes land as whole 128-byte lines of 8 different token rows.
// ------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void patchify_kernel(const T* __restrict__ img, bf16* __restrict__ out, int B, int M, int D, int H, int W, int dp, int hp, int wp,
                                int64_t stride_b, int64_t stride_m, int row_off, int64_t total_vec) {
  const int Wv = W / VEC;
  const int Dn = D / dp, Wn = W / wp;
  const int pd = dp * hp * wp;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_vec; idx += (int64_t)gridDim.x * blockDim.x) {
    const int wv = (int)(idx % Wv);
    int64_t r = idx / Wv;
    const int hh = (int)(r % H); r /= H;
    const int dd = (int)(r % D); r /= D;
    const int vol = (int)r;  // b*M + m
    const int b = vol / M, m = vol - b * M;
    const int w0 = wv * VEC;
    const int w = w0 / wp, p3 = w0 - w * wp;
    const int h = hh / hp, p2 = hh - h * hp;
    const int d = dd / dp, p1 = dd - d * dp;
    const int t = (h * Wn + w) * Dn + d;
    const int f = (p1 * hp + p2) * wp + p3;
    bf16* dst = out + ((int64_t)b * stride_b + (int64_t)m * stride_m + t + row_off) * pd + f;
    const T* src = img + idx * VEC;
    if constexpr (VEC == 8) {
      bf16x8 o;
      if constexpr (sizeof(T) == 4) {
        const f32x4 a = ((const f32x4*)src)[0], c = ((const f32x4*)src)[1];
        o = bf16x8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(c[0]), f2bf(c[1]), f2bf(c[2]), f2bf(c[3])};
      } else {
        o = *(const bf16x8*)src;
      }
      *(bf16x8*)dst = o;
    } else {
      *dst = f2bf((float)*src);
    }
  }
}

// zero row 0 of every [rows_per, pd] sample (the CLS slot of the padded patch matrix)
__global__ void zero_rows_kernel(bf16* __restrict__ out, int samples, int64_t sample_stride, int pd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int pv = pd >> 3;
  if (i < (int64_t)samples * pv) {
    const int s = (int)(i / pv), c = (int)(i - (int64_t)s * pv);
    ((bf16x8*)(out + s * sample_stride))[c] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}

__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x, int MB, int64_t row_stride, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < MB * d) {
    const int c = i % d, mb = i / d;
    x[(int64_t)mb * row_stride + c] = cls[c] + pos[c];
  }
}

// dpos[n, c] += sum_mb dx[mb, n, c];  dcls[c] += sum_mb dx[mb, 0, c]
__global__ void embed_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dpos, float* __restrict__ dcls, int MB, int N, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over N*d/4
  const int dv = d >> 2;
  if (i >= (int64_t)N * dv) return;
  const int n = (int)(i / dv), c = (int)(i - (int64_t)n * dv) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int mb = 0; mb < MB; ++mb) acc += *(const f32x4*)(dx + ((int64_t)mb * N + n) * d + c);
  f32x4* dp = (f32x4*)(dpos + (int64_t)n * d + c);
  *dp += acc;
  if (n == 0) *(f32x4*)(dcls + c) += acc;
}

__global__ void cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 a = ((const f32x4*)src)[2 * i], c = ((const f32x4*)src)[2 * i + 1];
    ((bf16x8*)dst)[i] = bf16x8{f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(c[0]), f2bf(c[1]), f2bf(c[2]), f2bf(c[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nv << 3) + threadIdx.x] = f2bf(src[(nv << 3) + threadIdx.x]);
}

// out = a + b (fp32) with a bf16 copy of the sum: the fan-out sum of two gradient tensors (cross_vit._FanOut) handed on in both dtypes,
// one pass instead of an add and a cast (n % 8 == 0: rows of d % 8 == 0 floats)
__global__ void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, bf16* __restrict__ outb, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 s0 = ((const f32x4*)a)[2 * i] + ((const f32x4*)b)[2 * i], s1 = ((const f32x4*)a)[2 * i + 1] + ((const f32x4*)b)[2 * i + 1];
    ((f32x4*)out)[2 * i] = s0;
    ((f32x4*)out)[2 * i + 1] = s1;
    ((bf16x8*)outb)[i] = bf16x8{f2bf(s0[0]), f2bf(s0[1]), f2bf(s0[2]), f2bf(s0[3]), f2bf(s1[0]), f2bf(s1[1]), f2bf(s1[2]), f2bf(s1[3])};
  }
}

// dst[r, :] (and dst2[r, :]) = a[r, :] + b[r, :] over `rows` rows of d columns; a missing operand is zero; every tensor has its own
// dtype and row stride (the B CLS rows of a [B, N, d] token tensor are rows of stride N d).  One block per row.  The CLS-row
// bookkeeping of the fusions (model_cross.py:140-142: split off / re-attach the CLS token, and the same on the gradients) in ONE launch
// per site instead of 2-4 strided torch copies / adds.  dst may be a or b.
template <typename T>
__device__ __forceinline__ float ld1(const void* p, int64_t i) {
  if constexpr (sizeof(T) == 4) return ((const float*)p)[i];
  else return bf2f(((const bf16*)p)[i]);
}
__device__ __forceinline__ float ld_dt(const void* p, int dt, int64_t i) { return dt == XVIT_F32 ? ld1<float>(p, i) : ld1<bf16>(p, i); }
__device__ __forceinline__ void st_dt(void* p, int dt, int64_t i, float v) {
  if (dt == XVIT_F32) ((float*)p)[i] = v;
  else ((bf16*)p)[i] = f2bf(v);
}
__global__ __launch_bounds__(256) void rows_combine_kernel(void* __restrict__ dst, int dst_dt, int64_t ld_dst, void* __restrict__ dst2, int dst2_dt, int64_t ld_dst2,
                                                           const void* a, int a_dt, int64_t ld_a, const void* b, int b_dt, int64_t ld_b, int d) {
  const int64_t r = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    float v = 0.f;
    if (a) v = ld_dt(a, a_dt, r * ld_a + c);
    if (b) v += ld_dt(b, b_dt, r * ld_b + c);
    if (dst_dt == XVIT_BF16) v = bf2f(f2bf(v));      // what the bf16 destination holds is what a second destination gets too
    st_dt(dst, dst_dt, r * ld_dst + c, v);
    if (dst2) st_dt(dst2, dst2_dt, r * ld_dst2 + c, v);
  }
}

__global__ void zero_f32_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// column sums: block = 64 column-quads (256 columns) x 4 row lanes over a contiguous slab of rows;
// 4 independent 16-B loads in flight per thread; one atomic per column per block
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *(const f32x4*)p;
  } else {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
  }
}
// `part` != nullptr: deterministic form — each row chunk stores its partial sums to part[chunk][n] (plain stores) and
// colsum_reduce_kernel adds them up in chunk order; otherwise the chunks meet in fp32 atomics on `out`.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ out, float* __restrict__ part, int rows, int n,
                                                     int rows_per_block) {
  __shared__ f32x4 red[4][64];
  const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cq) * 4;
  const int r_begin = blockIdx.y * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (col < n) {
    const T* base = x + col;
    int r = r_begin + rl;
    for (; r + 12 < r_end; r += 16) {
      a0 += ld4(base + (int64_t)r * ldx);
      a1 += ld4(base + (int64_t)(r + 4) * ldx);
      a2 += ld4(base + (int64_t)(r + 8) * ldx);
      a3 += ld4(base + (int64_t)(r + 12) * ldx);
    }
    for (; r < r_end; r += 4) a0 += ld4(base + (int64_t)r * ldx);
  }
  red[rl][cq] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && col < n) {
    const f32x4 acc = red[0][cq] + red[1][cq] + red[2][cq] + red[3][cq];
    if (part) {
      *(f32x4*)(part + (int64_t)blockIdx.y * n + col) = acc;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(out + col + e, acc[e]);
    }
  }
}

// out[c] (+)= sum over chunks of part[chunk][c], in chunk order (bit-reproducible)
__global__ void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int chunks, int n, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float acc = accumulate ? out[c] : 0.f;
  for (int k = 0; k < chunks; ++k) acc += part[(int64_t)k * n + c];
  out[c] = acc;
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p, float inv_keep, uint64_t seed_in, const uint64_t* __restrict__ epoch) {
  const uint64_t seed = drop_seed_at(seed_in, epoch);
  const uint32_t thr = (uint32_t)(p * 16777216.0f);  // drop when the 24-bit hash < p * 2^24
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool keep = (hash32(seed, (uint64_t)i) & 0xFFFFFFu) >= thr;
    y[i] = keep ? (T)((float)x[i] * inv_keep) : (T)0.0f;
  }
}

// ---- tiny fp32 linear (num_classes head) ---------------------------------------------------
__global__ void small_linear_fwd_kernel(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ W, const float* __restrict__ b,
                                        float* __restrict__ y, int M, int N, int K) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= M * N) return;
  const int m = wave / N, n = wave - m * N;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc = fmaf(bf2f(x[(int64_t)m * ldx + k]), W[(int64_t)n * K + k], acc);
  acc = wave_sum(acc);
  if (lane == 0) y[wave] = acc + (b ? b[n] : 0.f);
}

// grid (K / 256, row chunks): a block handles `rows` consecutive samples for 256 columns; the dW / db partial sums of
// the chunks meet in fp32 atomics (the caller zero-fills dW and db).  One thread per column looping over ALL rows
// (the first version) is a chain of M dependent loads: 119 us at M = 126 for 0.8 MB of data.
__global__ void small_linear_bwd_kernel(const float* __restrict__ dy, const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ W,
                                        const bf16* __restrict__ z, int64_t ldz, bf16* __restrict__ dx, int64_t lddx, float* __restrict__ dW,
                                        float* __restrict__ db, int M, int N, int K, int rows) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int m0 = blockIdx.y * rows, m1 = min(M, m0 + rows);
  if (k < K) {
    for (int m = m0; m < m1; ++m) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc = fmaf(dy[m * N + n], W[(int64_t)n * K + k], acc);
      if (z) acc *= dgelu_f(bf2f(z[(int64_t)m * ldz + k]));
      dx[(int64_t)m * lddx + k] = f2bf(acc);
    }
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      for (int m = m0; m < m1; ++m) acc = fmaf(dy[m * N + n], bf2f(x[(int64_t)m * ldx + k]), acc);
      unsafeAtomicAdd(dW + (int64_t)n * K + k, acc);
    }
  }
  if (k < N) {   // block column 0 only (K >= N)
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) acc += dy[m * N + k];
    unsafeAtomicAdd(db + k, acc);
  }
}

// logits = mean_m logits_m; loss = mean_b CE(logits_b, label_b) with label smoothing;
// dlogits_m = (softmax - target) / (B * M).   One block; B*C small.
__global__ void mean_ce_kernel(const float* __restrict__ logits_m, const int64_t* __restrict__ labels, float eps, float* __restrict__ logits,
                               float* __restrict__ loss, float* __restrict__ dlogits_m, int M, int B, int C) {
  __shared__ float red[4];
  float part = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int m = 0; m < M; ++m) s += logits_m[((int64_t)m * B + b) * C + c];
      s /= (float)M;
      logits[b * C + c] = s;
      mx = fmaxf(mx, s);
    }
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += expf(logits[b * C + c] - mx);
    const float lz = mx + logf(z);
    const int y = (int)labels[b];
    float l = 0.f;
    for (int c = 0; c < C; ++c) {
      const float logp = logits[b * C + c] - lz;
      const float tgt = (c == y ? 1.0f - eps : 0.f) + eps / (float)C;
      l -= tgt * logp;
      const float g = (expf(logp) - tgt) / (float)(B * M);
      for (int m = 0; m < M; ++m) dlogits_m[((int64_t)m * B + b) * C + c] = g;
    }
    part += l;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    *loss = t / (float)B;
  }
}

}  // namespace xvit

using namespace xvit;

static int grid_for(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int xvit_patchify(const void* img, int img_dtype, void* out, int B, int M, int D, int H, int W, int dp, int hp, int wp,
                             int64_t stride_b, int64_t stride_m, int row_off, int zero_rows, int64_t zero_row_stride, xvit_stream_t stream) {
  XVIT_REQUIRE(img && out, "xvit_patchify: null pointer");
  XVIT_REQUIRE(B > 0 && M > 0 && D > 0 && H > 0 && W > 0 && dp > 0 && hp > 0 && wp > 0, "xvit_patchify: bad sizes");
  XVIT_REQUIRE(D % dp == 0 && H % hp == 0 && W % wp == 0, "xvit_patchify: image dimensions must be divisible by the patch size");
  XVIT_REQUIRE(img_dtype == XVIT_F32 || img_dtype == XVIT_BF16, "xvit_patchify: bad dtype");
  XVIT_REQUIRE(stride_b > 0 && stride_m > 0 && row_off >= 0 && zero_rows >= 0, "xvit_patchify: bad output placement");
  XVIT_REQUIRE(zero_rows == 0 || (dp * hp * wp) % 8 == 0, "xvit_patchify: zero rows need patch_dim %% 8 == 0");
  const int64_t total = (int64_t)B * M * D * H * W;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (wp % 8 == 0) && ((reinterpret_cast<uintptr_t>(img) & 31) == 0);
  bf16* o = (bf16*)out;
  if (vec) {
    const int64_t tv = total / 8;
    if (img_dtype == XVIT_F32) hipLaunchKernelGGL((patchify_kernel<float, 8>), dim3(grid_for(tv, 256)), dim3(256), 0, s, (const float*)img, o, B, M, D, H, W, dp, hp, wp, stride_b, stride_m, row_off, tv);
    else hipLaunchKernelGGL((patchify_kernel<bf16, 8>), dim3(grid_for(tv, 256)), dim3(256), 0, s, (const bf16*)img, o, B, M, D, H, W, dp, hp, wp, stride_b, stride_m, row_off, tv);
  } else {
    if (img_dtype == XVIT_F32) hipLaunchKernelGGL((patchify_kernel<float, 1>), dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)img, o, B, M, D, H, W, dp, hp, wp, stride_b, stride_m, row_off, total);
    else hipLaunchKernelGGL((patchify_kernel<bf16, 1>), dim3(grid_for(total, 256)), dim3(256), 0, s, (const bf16*)img, o, B, M, D, H, W, dp, hp, wp, stride_b, stride_m, row_off, total);
  }
  if (zero_rows > 0) {
    const int pd = dp * hp * wp;
    const int64_t work = (int64_t)zero_rows * (pd / 8);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, o, zero_rows, zero_row_stride * pd, pd);
  }
  return check_launch("xvit_patchify");
}

extern "C" int xvit_cls_row_fwd(const float* cls, const float* pos, float* x, int MB, int N, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(cls && pos && x && MB > 0 && N > 0 && d > 0, "xvit_cls_row_fwd: bad arguments");
  hipLaunchKernelGGL(cls_row_kernel, dim3((MB * d + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos, x, MB, (int64_t)N * d, d);
  return check_launch("xvit_cls_row_fwd");
}

extern "C" int xvit_embed_bwd(const float* dx, float* dpos, float* dcls, int MB, int N, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(dx && dpos && dcls && MB > 0 && N > 0 && d > 0 && d % 4 == 0, "xvit_embed_bwd: bad arguments");
  const int64_t work = (int64_t)N * (d / 4);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, dpos, dcls, MB, N, d);
  return check_launch("xvit_embed_bwd");
}

extern "C" int xvit_cast_f32_bf16(const float* src, void* dst, int64_t n, xvit_stream_t stream) {
  XVIT_REQUIRE(src && dst && n > 0, "xvit_cast_f32_bf16: bad arguments");
  XVIT_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, "xvit_cast_f32_bf16: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n);
  return check_launch("xvit_cast_f32_bf16");
}

extern "C" int xvit_add_cast_f32_bf16(const float* a, const float* b, float* out, void* out_bf16, int64_t n, xvit_stream_t stream) {
  XVIT_REQUIRE(a && b && out && out_bf16 && n > 0 && n % 8 == 0, "xvit_add_cast_f32_bf16: bad arguments (n must be a positive multiple of 8)");
  XVIT_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out_bf16)) & 15) == 0,
               "xvit_add_cast_f32_bf16: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(add_cast_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (bf16*)out_bf16, n);
  return check_launch("xvit_add_cast_f32_bf16");
}

extern "C" int xvit_rows_combine(void* dst, int dst_dtype, int64_t ld_dst, void* dst2, int dst2_dtype, int64_t ld_dst2, const void* a, int a_dtype, int64_t ld_a,
                                 const void* b, int b_dtype, int64_t ld_b, int rows, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(dst && rows > 0 && d > 0, "xvit_rows_combine: bad arguments");
  auto ok = [](int dt) { return dt == XVIT_F32 || dt == XVIT_BF16; };
  XVIT_REQUIRE(ok(dst_dtype) && (!dst2 || ok(dst2_dtype)) && (!a || ok(a_dtype)) && (!b || ok(b_dtype)), "xvit_rows_combine: dtypes must be XVIT_F32 or XVIT_BF16");
  XVIT_REQUIRE(ld_dst >= d && (!dst2 || ld_dst2 >= d) && (!a || ld_a >= d) && (!b || ld_b >= d), "xvit_rows_combine: a row stride is smaller than d");
  hipLaunchKernelGGL(rows_combine_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dst, dst_dtype, ld_dst, dst2, dst2_dtype, ld_dst2, a, a_dtype, ld_a, b, b_dtype, ld_b, d);
  return check_launch("xvit_rows_combine");
}

static int colsum_rows_per_block(int rows, int n) {
  const int gx = (n / 4 + 63) / 64;
  int rpb = 64;                                   // rows per block: >= 64, and at most ~2048 blocks in all
  while ((int64_t)gx * ((rows + rpb - 1) / rpb) > 2048) rpb *= 2;
  return rpb;
}

extern "C" int64_t xvit_colsum_workspace_bytes(int rows, int n) {
  if (rows <= 0 || n <= 0) return 0;
  const int rpb = colsum_rows_per_block(rows, n);
  return (int64_t)((rows + rpb - 1) / rpb) * n * (int64_t)sizeof(float);
}

extern "C" int xvit_colsum(const void* x, int x_dtype, int64_t ldx, float* out, int rows, int n, int accumulate, float* workspace,
                           int64_t workspace_bytes, xvit_stream_t stream) {
  XVIT_REQUIRE(x && out && rows > 0 && n > 0, "xvit_colsum: bad arguments");
  XVIT_REQUIRE(n % 4 == 0 && ldx % 4 == 0, "xvit_colsum: n and ldx must be multiples of 4");
  XVIT_REQUIRE(!workspace || workspace_bytes >= xvit_colsum_workspace_bytes(rows, n), "xvit_colsum: workspace too small (%lld bytes)", (long long)workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && !workspace)   // a kernel, not hipMemsetAsync: the memset node was observed not to replay from a captured HIP graph
    hipLaunchKernelGGL(zero_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, n);
  const int gx = (n / 4 + 63) / 64;
  const int rpb = colsum_rows_per_block(rows, n);
  const dim3 grid(gx, (rows + rpb - 1) / rpb), block(256);
  if (x_dtype == XVIT_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, s, (const float*)x, ldx, out, workspace, rows, n, rpb);
  else hipLaunchKernelGGL((colsum_kernel<bf16>), grid, block, 0, s, (const bf16*)x, ldx, out, workspace, rows, n, rpb);
  if (workspace) hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, workspace, out, (int)grid.y, n, accumulate);
  return check_launch("xvit_colsum");
}

extern "C" int xvit_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, xvit_stream_t stream) {
  XVIT_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "xvit_dropout: bad arguments");
  const float inv = 1.0f / (1.0f - p);
  if (dtype == XVIT_F32) hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, p, inv, seed, drop_epoch_ptr());
  else hipLaunchKernelGGL((dropout_kernel<bf16>), dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n, p, inv, seed, drop_epoch_ptr());
  return check_launch("xvit_dropout");
}

extern "C" int xvit_small_linear_fwd(const void* x, int64_t ldx, const float* W, const float* b, float* y, int M, int N, int K, xvit_stream_t stream) {
  XVIT_REQUIRE(x && W && y && M > 0 && N > 0 && K > 0, "xvit_small_linear_fwd: bad arguments");
  const int waves = M * N;
  hipLaunchKernelGGL(small_linear_fwd_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, W, b, y, M, N, K);
  return check_launch("xvit_small_linear_fwd");
}

extern "C" int xvit_small_linear_bwd(const float* dy, const void* x, int64_t ldx, const float* W, const void* z, int64_t ldz, void* dx, int64_t lddx,
                                     float* dW, float* db, int M, int N, int K, int deterministic, xvit_stream_t stream) {
  XVIT_REQUIRE(dy && x && W && dx && dW && db && M > 0 && N > 0 && K >= N, "xvit_small_linear_bwd: bad arguments");
  const int rows = deterministic ? M : 8;   // one row chunk: every dW / db element receives exactly one add (bit-reproducible, slower)
  hipLaunchKernelGGL(small_linear_bwd_kernel, dim3((K + 255) / 256, (M + rows - 1) / rows), dim3(256), 0, (hipStream_t)stream, dy, (const bf16*)x, ldx, W,
                     (const bf16*)z, ldz, (bf16*)dx, lddx, dW, db, M, N, K, rows);
  return check_launch("xvit_small_linear_bwd");
}

extern "C" int xvit_mean_ce(const float* logits_m, const int64_t* labels, float label_smoothing, float* logits, float* loss, float* dlogits_m, int M,
                            int B, int C, xvit_stream_t stream) {
  XVIT_REQUIRE(logits_m && labels && logits && loss && dlogits_m && M > 0 && B > 0 && C > 0, "xvit_mean_ce: bad arguments");
  hipLaunchKernelGGL(mean_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits_m, labels, label_smoothing, logits, loss, dlogits_m, M, B, C);
  return check_launch("xvit_mean_ce");
}

// ------------------------------------------------------------------------------------------
// Fused multi-tensor Adam (reference: torch.optim.Adam(lr, weight_decay) at model_cross.py:277 — L2 decay added
// to the gradient, bias-corrected moments, eps outside the square root).  One launch updates every tensor of a
// parameter group: p, m, v in fp32 and, when given, the bf16 operand copy of p (so no separate cast pass).
// ------------------------------------------------------------------------------------------
namespace xvit {
struct AdamTensor { float* p; const float* g; float* m; float* v; bf16* shadow; int64_t n; };
constexpr int ADAM_CHUNK = 16384;   // elements per block

__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor* __restrict__ table, const int2* __restrict__ chunks, float lr_over_bc1,
                                                   float beta1, float beta2, float eps, float wd, float inv_sqrt_bc2, float grad_scale) {
  const int2 ch = chunks[blockIdx.x];
  const AdamTensor t = table[ch.x];
  const int64_t begin = (int64_t)ch.y * ADAM_CHUNK;
  const int64_t end = begin + ADAM_CHUNK < t.n ? begin + ADAM_CHUNK : t.n;
  const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                     reinterpret_cast<uintptr_t>(t.v)) & 15) == 0 && (!t.shadow || (reinterpret_cast<uintptr_t>(t.shadow) & 7) == 0);
  auto upd = [&](float& p, float g, float& m, float& v) {
    g = g * grad_scale + wd * p;
    m = beta1 * m + (1.0f - beta1) * g;
    v = beta2 * v + (1.0f - beta2) * g * g;
    p -= lr_over_bc1 * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
  };
  if (vec) {
    for (int64_t i = begin + threadIdx.x * 4; i + 3 < end; i += 256 * 4) {
      f32x4 p = *(f32x4*)(t.p + i), m = *(f32x4*)(t.m + i), v = *(f32x4*)(t.v + i);
      const f32x4 g = *(const f32x4*)(t.g + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // vector elements cannot bind to references: go through scalars
        float pe = p[e], me = m[e], ve = v[e];
        upd(pe, g[e], me, ve);
        p[e] = pe; m[e] = me; v[e] = ve;
      }
      *(f32x4*)(t.p + i) = p; *(f32x4*)(t.m + i) = m; *(f32x4*)(t.v + i) = v;
      if (t.shadow) *(bf16x4*)(t.shadow + i) = bf16x4{f2bf(p[0]), f2bf(p[1]), f2bf(p[2]), f2bf(p[3])};
    }
    const int64_t tail = begin + ((end - begin) & ~(int64_t)3);
    for (int64_t i = tail + threadIdx.x; i < end; i += 256) {
      upd(t.p[i], t.g[i], t.m[i], t.v[i]);
      if (t.shadow) t.shadow[i] = f2bf(t.p[i]);
    }
  } else {
    for (int64_t i = begin + threadIdx.x; i < end; i += 256) {
      upd(t.p[i], t.g[i], t.m[i], t.v[i]);
      if (t.shadow) t.shadow[i] = f2bf(t.p[i]);
    }
  }
}
}  // namespace xvit

extern "C" int xvit_adam_step(const void* table_dev, const void* chunks_dev, int n_chunks, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, xvit_stream_t stream) {
  XVIT_REQUIRE(table_dev && chunks_dev && n_chunks > 0 && step >= 1, "xvit_adam_step: bad arguments");
  XVIT_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "xvit_adam_step: bad hyper-parameters");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(xvit::adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const xvit::AdamTensor*)table_dev, (const int2*)chunks_dev,
                     (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale);
  return xvit::check_launch("xvit_adam_step");
}

// ------------------------------------------------------------------------------------------
// Input stage (reference dataset_ucsf.py:84-88,152-158): MONAI ResizeWithPadOrCropd(spatial_size, constant_values=-1)
// = symmetric pad (before = deficit/2, after = the rest) then centre crop (start = size/2 - target/2), then .float().
// Here: int16 NIfTI payload [nvol, Ds, Hs, Ws] on the device -> bf16 [nvol, D, H, W], one pass, no fp32 volume.
// ------------------------------------------------------------------------------------------
namespace xvit {
__global__ void resize_pad_crop_i16_kernel(const int16_t* __restrict__ src, bf16* __restrict__ dst, int Ds, int Hs, int Ws, int D, int H, int W,
                                           int od, int oh, int ow, float pad_value, int64_t total) {
  // (od, oh, ow): source index = destination index + offset; negative offsets are padding
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    int64_t r = i / W;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D);
    const int64_t vol = r / D;
    const int sd = d + od, sh = h + oh, sw = w + ow;
    const bool in = sd >= 0 && sd < Ds && sh >= 0 && sh < Hs && sw >= 0 && sw < Ws;
    dst[i] = f2bf(in ? (float)src[((vol * Ds + sd) * Hs + sh) * (int64_t)Ws + sw] : pad_value);
  }
}
}  // namespace xvit

extern "C" int xvit_resize_pad_crop_i16(const void* src_i16, void* dst_bf16, int nvol, int Ds, int Hs, int Ws, int D, int H, int W,
                                        float pad_value, xvit_stream_t stream) {
  XVIT_REQUIRE(src_i16 && dst_bf16 && nvol > 0 && Ds > 0 && Hs > 0 && Ws > 0 && D > 0 && H > 0 && W > 0, "xvit_resize_pad_crop_i16: bad arguments");
  // per dimension: size < target -> pad, before = (target - size) / 2; size > target -> crop, start = size / 2 - target / 2
  auto offset = [](int size, int target) { return size >= target ? size / 2 - target / 2 : -((target - size) / 2); };
  const int64_t total = (int64_t)nvol * D * H * W;
  int64_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(xvit::resize_pad_crop_i16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const int16_t*)src_i16, (xvit::bf16*)dst_bf16,
                     Ds, Hs, Ws, D, H, W, offset(Ds, D), offset(Hs, H), offset(Ws, W), pad_value, total);
  return xvit::check_launch("xvit_resize_pad_crop_i16");
}

// ------------------------------------------------------------------------------------------
// Placement trace: where does the dispatcher put the workgroups of a stream?  Each block records its XCC
// (XCD) id and HW_ID, then lingers ~`linger_us` so that the launch spreads over every CU the stream may
// use.  Used to map hipExtStreamCreateWithCUMask bits to XCDs and to check the blockIdx -> XCD round-robin
// the GEMM tile remap assumes.
// ------------------------------------------------------------------------------------------
namespace xvit {
__global__ void cu_trace_kernel(uint32_t* __restrict__ out, int linger_us) {
  if (threadIdx.x == 0) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
  const uint64_t t0 = wall_clock64();   // 100 MHz
  while (wall_clock64() - t0 < (uint64_t)linger_us * 100) __builtin_amdgcn_s_sleep(32);
}
}  // namespace xvit

extern "C" int xvit_cu_trace(uint32_t* out, int nblocks, int linger_us, xvit_stream_t stream) {
  XVIT_REQUIRE(out && nblocks > 0 && linger_us >= 0 && linger_us <= 10000, "xvit_cu_trace: bad arguments");
  hipLaunchKernelGGL(xvit::cu_trace_kernel, dim3((unsigned)nblocks), dim3(64), 0, (hipStream_t)stream, out, linger_us);
  return xvit::check_launch("xvit_cu_trace");
}
