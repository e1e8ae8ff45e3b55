// LayerNorm forward / backward over the fp32 residual stream (HBM-bound).
// One wave per row, the row held in registers (float4 per lane per 256 columns), so x is read
// once; the backward also folds in the upstream residual gradient and emits the bf16 copy of
// dx that the following GEMMs consume.  dgamma/dbeta: per-lane column partials across all rows
// a wave visits, one LDS cross-wave reduction, one atomic per column per block.
#include "xvit_common.h"

namespace xvit {

constexpr int LN_WAVES = 8;    // forward: waves per block (rows are latency-bound per wave: keep many in flight per CU)
constexpr int LNB_WAVES = 4;   // backward: 4-wave blocks, [4][2][d] LDS reduction buffer (24 KB at d = 768)

template <int V>  // V float4 per lane: d <= 256*V
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x_alt, int64_t ldx,
                                                               int seq_len, int64_t ld_alt, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, bf16* __restrict__ y, int64_t ldy, float* __restrict__ yf, int64_t ldyf,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;  // float4 per row
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * LN_WAVES + wave; row < rows; row += gridDim.x * LN_WAVES) {
    // row 0 of every sequence may come from x_alt (the CLS rows of the other operand, one every ld_alt elements)
    const f32x4* xr = (const f32x4*)((x_alt && (row % seq_len) == 0) ? x_alt + (int64_t)(row / seq_len) * ld_alt : x + (int64_t)row * ldx);
    f32x4 v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      v[i] = c < nv ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(s) * inv_d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mu; ss += t * t; }
      }
    }
    const float rs = rsqrtf(wave_sum(ss) * inv_d + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const f32x4 g = ((const f32x4*)gamma)[c], b = ((const f32x4*)beta)[c];
        f32x4 of;
#pragma unroll
        for (int e = 0; e < 4; ++e) of[e] = (v[i][e] - mu) * rs * g[e] + b[e];
        if (y) {
          bf16x4 o = {f2bf(of[0]), f2bf(of[1]), f2bf(of[2]), f2bf(of[3])};
          *(bf16x4*)(y + (int64_t)row * ldy + c * 4) = o;
        }
        if (yf) *(f32x4*)(yf + (int64_t)row * ldyf + c * 4) = of;   // the single-token CLS path keeps its operands in fp32
      }
    }
  }
}

#ifndef XVIT_LNB_OCC
#define XVIT_LNB_OCC 2      // waves per SIMD the d <= 768 backward is compiled for: 2 = 184 registers, no spills; 3 = 168 registers with 15 spilled
                            // dwords, measured slower on the LN2 form (174 vs 150 us at 64 k rows; tools/ln_bench.py)
#endif
template <int V>
__global__ __launch_bounds__(LNB_WAVES * 64, V <= 3 ? XVIT_LNB_OCC : (V <= 4 ? 2 : 1)) void ln_bwd_kernel(const bf16* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                               const float* __restrict__ x_alt, int64_t ldx, int seq_len, int64_t ld_alt,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ dres, int64_t lddres,
                                                               float* __restrict__ dx, int64_t lddx, bf16* __restrict__ dxb, int64_t lddxb,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum,
                                                               float* __restrict__ dressum, float* __restrict__ part, int rows, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;  // [LNB_WAVES][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;
  const float inv_d = 1.0f / (float)d;
  f32x4 g[V], dg[V], db[V], sx[V], sr[V];   // sx/sr: column sums of dx and dres (bias gradients of the adjacent Linears)
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = lane + i * 64;
    g[i] = c < nv ? ((const f32x4*)gamma)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    sx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    sr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // A row costs one trip to memory (x, dy and dres are all requested up front) and two wave reductions; with one row per wave at a
  // time the kernel was bound by exactly that latency chain (87 % of wave time parked, 4.6 TB/s).  Two rows per wave are kept in
  // flight: the second row's loads are issued before the first row's arithmetic starts.
  struct RowIn { f32x4 xv[V], rv[V]; bf16x4 dv[V]; float mu, rs; };
  auto load_row = [&](RowIn& in, int row) {
    const f32x4* xr = (const f32x4*)((x_alt && (row % seq_len) == 0) ? x_alt + (int64_t)(row / seq_len) * ld_alt : x + (int64_t)row * ldx);
    const bf16* dyr = dy + (int64_t)row * lddy;
    in.mu = mean[row]; in.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      const bool ok = c < nv;
      in.xv[i] = ok ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
      in.dv[i] = ok ? *(const bf16x4*)(dyr + c * 4) : bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
      in.rv[i] = (ok && dres) ? *(const f32x4*)(dres + (int64_t)row * lddres + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto do_row = [&](const RowIn& in, int row) {
    const float mu = in.mu, rs = in.rs;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {     // columns past d: x - mu != 0 there, masked through dy = 0 and g = 0
        const float h = (in.xv[i][e] - mu) * rs, dyv = bf2f(in.dv[i][e]), gyv = dyv * g[i][e];
        dg[i][e] += dyv * h;
        db[i][e] += dyv;
        s1 += gyv;
        s2 += gyv * h;
      }
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)    // x-hat and dy g are recomputed rather than kept: 24 registers per row in flight
          o[e] = rs * (bf2f(in.dv[i][e]) * g[i][e] - m1 - (in.xv[i][e] - mu) * rs * m2);
        if (dres) {
          o += in.rv[i];
          sr[i] += in.rv[i];
        }
        sx[i] += o;
        *(f32x4*)(dx + (int64_t)row * lddx + c * 4) = o;
        if (dxb) {
          bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
          *(bf16x4*)(dxb + (int64_t)row * lddxb + c * 4) = ob;
        }
      }
    }
  };
  const int stride = gridDim.x * LNB_WAVES;
  if constexpr (V <= 4) {
    for (int row = blockIdx.x * LNB_WAVES + wave; row < rows; row += 2 * stride) {
      RowIn ra, rb;
      const int row2 = row + stride;
      const bool two = row2 < rows;     // wave-uniform
      load_row(ra, row);
      if (two) load_row(rb, row2);
      do_row(ra, row);
      if (two) do_row(rb, row2);
    }
  } else {                               // wide rows (d > 1024): one row fills the register file
    for (int row = blockIdx.x * LNB_WAVES + wave; row < rows; row += stride) {
      RowIn ra;
      load_row(ra, row);
      do_row(ra, row);
    }
  }
  // cross-wave reduction of the column partials through LDS in two rounds ([waves][2][d] each: dgamma|dbeta, then
  // the optional sum(dx)|sum(dres)), one global atomic per column per block.  (LDS float atomics into a single
  // accumulator were measured 2x slower: 8 waves hammer the same addresses.)
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (round == 1 && !dxsum && !dressum) break;
    if (round == 1) __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        *(f32x4*)(red + (wave * 2 + 0) * d + c * 4) = round == 0 ? dg[i] : sx[i];
        *(f32x4*)(red + (wave * 2 + 1) * d + c * 4) = round == 0 ? db[i] : sr[i];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) { a += red[(w * 2 + 0) * d + c]; b += red[(w * 2 + 1) * d + c]; }
      if (part) {   // deterministic form: per-block partial sums [block][4][d], added up in block order by ln_bwd_reduce_kernel
        part[((int64_t)blockIdx.x * 4 + round * 2 + 0) * d + c] = a;
        part[((int64_t)blockIdx.x * 4 + round * 2 + 1) * d + c] = b;
      } else if (round == 0) { unsafeAtomicAdd(dgamma + c, a); unsafeAtomicAdd(dbeta + c, b); }
      else { if (dxsum) unsafeAtomicAdd(dxsum + c, a); if (dressum) unsafeAtomicAdd(dressum + c, b); }
    }
  }
}

// dgamma / dbeta / dxsum / dressum += sum over blocks of part[block][k][c], in block order (bit-reproducible)
__global__ void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum,
                                     float* __restrict__ dressum, int blocks, int d) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float* const dst[4] = {dgamma, dbeta, dxsum, dressum};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!dst[k]) continue;
    float acc = dst[k][c];
    for (int b = 0; b < blocks; ++b) acc += part[((int64_t)b * 4 + k) * d + c];
    dst[k][c] = acc;
  }
}

}  // namespace xvit

using namespace xvit;

// Every block adds its dgamma / dbeta partials to the SAME d addresses: more blocks = atomic contention (measured slower beyond
// 768 at 64 k rows).  With few rows the atomics dominate outright — at 4104 rows (the reference's batch 8) one row per wave on 768
// blocks took 26 us, 768 adds queueing on every address — so a block takes at least 16 rows (4 per wave).
static int ln_bwd_grid(int rows) {
  int g = (rows + 4 * LNB_WAVES - 1) / (4 * LNB_WAVES);
  if (g > 768) g = 768;
  return g < 1 ? 1 : g;
}

extern "C" int64_t xvit_layernorm_bwd_workspace_bytes(int rows, int d) {
  if (rows <= 0 || d <= 0) return 0;
  const int g = ln_bwd_grid(rows);
  return (int64_t)g * 4 * d * (int64_t)sizeof(float);
}

static int ln_grid(int rows) {
  const int want = (rows + LN_WAVES - 1) / LN_WAVES;
  return want < 2048 ? want : 2048;  // grid-stride beyond 8 blocks/CU
}

extern "C" int xvit_layernorm_fwd(const float* x, const float* x_alt, int64_t ldx, int seq_len, int64_t ld_alt, const float* gamma, const float* beta,
                                  float eps, void* y, int64_t ldy, float* y_f32, int64_t ldyf, float* mean, float* rstd, int rows, int d,
                                  xvit_stream_t stream) {
  XVIT_REQUIRE(x && gamma && beta && (y || y_f32) && mean && rstd, "xvit_layernorm_fwd: null pointer");
  XVIT_REQUIRE(!y_f32 || (ldyf % 4 == 0 && ldyf >= d), "xvit_layernorm_fwd: ldyf must be a multiple of 4 and >= d");
  XVIT_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 4096, "xvit_layernorm_fwd: need 0 < d <= 4096, d %% 4 == 0 (d=%d rows=%d)", d, rows);
  XVIT_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, "xvit_layernorm_fwd: ldx/ldy must be multiples of 4 and >= d");
  XVIT_REQUIRE(!x_alt || (seq_len > 0 && ld_alt >= d && ld_alt % 4 == 0), "xvit_layernorm_fwd: x_alt needs seq_len > 0 and ld_alt >= d, a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(ln_grid(rows)), block(LN_WAVES * 64);
  bf16* yb = (bf16*)y;
  if (d <= 768) hipLaunchKernelGGL((ln_fwd_kernel<3>), grid, block, 0, s, x, x_alt, ldx, seq_len, ld_alt, gamma, beta, eps, yb, ldy, y_f32, ldyf, mean, rstd, rows, d);
  else if (d <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<4>), grid, block, 0, s, x, x_alt, ldx, seq_len, ld_alt, gamma, beta, eps, yb, ldy, y_f32, ldyf, mean, rstd, rows, d);
  else hipLaunchKernelGGL((ln_fwd_kernel<16>), grid, block, 0, s, x, x_alt, ldx, seq_len, ld_alt, gamma, beta, eps, yb, ldy, y_f32, ldyf, mean, rstd, rows, d);
  return check_launch("xvit_layernorm_fwd");
}

extern "C" int xvit_layernorm_bwd(const void* dy, int64_t lddy, const float* x, const float* x_alt, int64_t ldx, int seq_len, int64_t ld_alt,
                                  const float* mean, const float* rstd, const float* gamma, const float* dres, int64_t lddres, float* dx,
                                  int64_t lddx, void* dxb, int64_t lddxb, float* dgamma, float* dbeta, float* dxsum, float* dressum, int rows,
                                  int d, float* workspace, int64_t workspace_bytes, xvit_stream_t stream) {
  XVIT_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "xvit_layernorm_bwd: null pointer");
  XVIT_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 4096, "xvit_layernorm_bwd: need 0 < d <= 4096, d %% 4 == 0 (d=%d rows=%d)", d, rows);
  XVIT_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (!dres || lddres % 4 == 0) && (!dxb || lddxb % 4 == 0),
               "xvit_layernorm_bwd: leading dimensions must be multiples of 4");
  XVIT_REQUIRE(!x_alt || (seq_len > 0 && ld_alt >= d && ld_alt % 4 == 0), "xvit_layernorm_bwd: x_alt needs seq_len > 0 and ld_alt >= d, a multiple of 4");
  XVIT_REQUIRE(!dressum || dres, "xvit_layernorm_bwd: dressum needs dres");
  hipStream_t s = (hipStream_t)stream;
  const int g = ln_bwd_grid(rows);
  XVIT_REQUIRE(!workspace || workspace_bytes >= (int64_t)g * 4 * d * (int64_t)sizeof(float), "xvit_layernorm_bwd: workspace too small (%lld bytes)",
               (long long)workspace_bytes);
  const dim3 grid(g), block(LNB_WAVES * 64);
  const size_t lds = (size_t)LNB_WAVES * 2 * d * sizeof(float);
  const bf16* dyb = (const bf16*)dy;
  bf16* dxbb = (bf16*)dxb;
  if (d <= 768)
    hipLaunchKernelGGL((ln_bwd_kernel<3>), grid, block, lds, s, dyb, lddy, x, x_alt, ldx, seq_len, ld_alt, mean, rstd, gamma, dres, lddres, dx, lddx, dxbb, lddxb, dgamma, dbeta, dxsum, dressum, workspace, rows, d);
  else if (d <= 1024)
    hipLaunchKernelGGL((ln_bwd_kernel<4>), grid, block, lds, s, dyb, lddy, x, x_alt, ldx, seq_len, ld_alt, mean, rstd, gamma, dres, lddres, dx, lddx, dxbb, lddxb, dgamma, dbeta, dxsum, dressum, workspace, rows, d);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<16>), grid, block, lds, s, dyb, lddy, x, x_alt, ldx, seq_len, ld_alt, mean, rstd, gamma, dres, lddres, dx, lddx, dxbb, lddxb, dgamma, dbeta, dxsum, dressum, workspace, rows, d);
  if (workspace) {
    // without dxsum / dressum the kernel skips round 1: those slices of the workspace are never read either
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((d + 255) / 256), dim3(256), 0, s, workspace, dgamma, dbeta, dxsum, dressum, g, d);
  }
  return check_launch("xvit_layernorm_bwd");
}
