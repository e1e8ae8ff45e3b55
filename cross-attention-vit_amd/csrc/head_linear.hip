// Per-head fp32 products of the CLS-query fusion in its low-rank form (reference model_cross.py:88-99, CrossAttention).
//
// The fusion has ONE query row per (sample, head), so the key / value projections over all N tokens (wk, wv: 2 x 152 GFLOP per
// fusion at configs[1], a [B N, 2 d] tensor written and re-read) collapse algebraically:
//   scores[b, n, h] = q_h . (Wk_h hn[b, n] + bk_h)      = hn[b, n] . U[b, h] + const(n)      U[b, h, :] = q[b, h, :] Wk_h    (HEAD_ROWS)
//   out[b, h, :]    = sum_n p[b, n, h] (Wv_h hn[b, n] + bv_h) = Wv_h S[b, h] + bv_h          S[b, h, :] = sum_n p hn[b, n, :] (HEAD_COLS)
// (the constant drops out of the softmax over n; the probabilities sum to one in front of bv; with dropout on the probabilities,
// model_cross.py:97, p'[n] = m[n] p[n] / (1 - rate): S sums the kept weights, 1 / (1 - rate) rides on the row scale and bv is weighted
// by sum_n p'[n] — xvit_cls_softmax_fwd's `stat` rows), and in the backward
//   dp[b, n, h]     = hn[b, n] . Y[b, h] + const(n)                                           Y[b, h, :] = dO[b, h, :] Wv_h   (HEAD_ROWS)
//   dq[b, h, :]     = Wk_h T[b, h]                                                            T[b, h, :] = sum_n ds hn[b, n, :] (HEAD_COLS)
//   dWk_h = sum_b q[b, h, :]^T T[b, h, :],   dWv_h = sum_b dO[b, h, :]^T S[b, h, :]                                            (HEAD_WGRAD)
// The two passes over hn (row dots, weighted row sums) are batched GEMMs (xvit_gemm); what is left are these small per-head
// products with the 64 x d slices of the fp32 MASTER weights: M = batch rows, fp32 operands on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chains), one wave per 32 x 32 output tile — the single-token CLS path stays fp32 end to end.
#include "xvit_common.h"

namespace xvit {

constexpr int HL_DH = 64;

struct HeadParams {
  const float* x; const float* W; const float* t; const float* rs; const float* bias; const float* bias_scale;
  float* out; bf16* out_bf16;
  int64_t ldx, ldw, t_sb, t_sh, rs_ld, bsc_ld, out_sb, out_sh, ob_sb, ob_sh, ldo;
  int B, H, d, ob_heads;
};

// out[b, h, c] = sum_e x[b, 64 h + e] W[64 h + e, c]          grid (ceil(B / 32) * d / 32, H)
__global__ __launch_bounds__(64) void head_rows_kernel(const HeadParams p) {
  const int lane = threadIdx.x, r = lane & 31, hl = lane >> 5, h = blockIdx.y;
  const int ntn = p.d >> 5, tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  if (h >= p.H) {   // padding rows of the bf16 operand copy (heads H .. ob_heads - 1): zeros, written here rather than by a memset in front
    for (int i = lane; i < 32 * 32; i += 64) {
      const int row = tm * 32 + (i >> 5);
      if (row < p.B) p.out_bf16[(int64_t)row * p.ob_sb + (int64_t)h * p.ob_sh + tn * 32 + (i & 31)] = f2bf(0.f);
    }
    return;
  }
  const int b = min(tm * 32 + r, p.B - 1);       // rows past the end are clamped for the loads, never stored
  const float* xp = p.x + (int64_t)b * p.ldx + h * HL_DH + 4 * hl;
  const float* wp = p.W + (int64_t)(h * HL_DH + 4 * hl) * p.ldw + tn * 32 + r;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < HL_DH; k0 += 8) {      // MFMA e of a step contracts k = k0 + e (lane half 0) and k0 + 4 + e (half 1) on both operands
    const f32x4 a = *(const f32x4*)(xp + k0);
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = wp[(int64_t)(k0 + e) * p.ldw];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], w[e], acc, 0, 0, 0);
  }
  const int col = tn * 32 + r;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * hl;
    if (row < p.B) {
      p.out[(int64_t)row * p.out_sb + (int64_t)h * p.out_sh + col] = acc[i];
      if (p.out_bf16) p.out_bf16[(int64_t)row * p.ob_sb + (int64_t)h * p.ob_sh + col] = f2bf(acc[i]);
    }
  }
}

// out[b, 64 h + e] = rs[b, h] * sum_c t[b, h, c] W[64 h + e, c] + bias_scale[b, h] * bias[64 h + e]      grid (ceil(B / 32) * 2, H), 4 waves split K = d
__global__ __launch_bounds__(256) void head_cols_kernel(const HeadParams p) {
  __shared__ float part[3][32 * 33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hl = lane >> 5, h = blockIdx.y;
  const int tm = blockIdx.x >> 1, tn = blockIdx.x & 1;
  const int b = min(tm * 32 + r, p.B - 1);
  const float* tp = p.t + (int64_t)b * p.t_sb + (int64_t)h * p.t_sh + 4 * hl;
  const float* wp = p.W + (int64_t)(h * HL_DH + tn * 32 + r) * p.ldw + 4 * hl;
  const int kq = ((p.d >> 3) + 3) >> 2 << 3, kb = wave * kq, ke = min(p.d, kb + kq);     // d % 8 == 0 (host-checked)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
  for (int k0 = kb; k0 < ke; k0 += 8) {
    const f32x4 a = *(const f32x4*)(tp + k0), w = *(const f32x4*)(wp + k0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], w[e], acc, 0, 0, 0);
  }
  // the four K-quarters meet in LDS and are added in wave order (fixed: bit-reproducible)
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave - 1][((i & 3) + 8 * (i >> 2) + 4 * hl) * 33 + r] = acc[i];
  }
  __syncthreads();
  if (wave > 0) return;
  const int col = h * HL_DH + tn * 32 + r;
  const float bs = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rl = (i & 3) + 8 * (i >> 2) + 4 * hl, row = tm * 32 + rl;
    if (row < p.B) {
      const float v = ((acc[i] + part[0][rl * 33 + r]) + part[1][rl * 33 + r]) + part[2][rl * 33 + r];
      const float o = fmaf(v, p.rs ? p.rs[(int64_t)row * p.rs_ld + h] : 1.f, p.bias_scale ? bs * p.bias_scale[(int64_t)row * p.bsc_ld + h] : bs);
      p.out[(int64_t)row * p.ldo + col] = o;
      if (p.out_bf16) p.out_bf16[(int64_t)row * p.ob_sb + col] = f2bf(o);
    }
  }
}

// dW[64 h + e, c] = sum_b x[b, 64 h + e] rs[b, h] t[b, h, c]          grid (2 * d / 32, H)
__global__ __launch_bounds__(64) void head_wgrad_kernel(const HeadParams p) {
  const int lane = threadIdx.x, r = lane & 31, hl = lane >> 5, h = blockIdx.y;
  const int ntn = p.d >> 5, tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  const int xcol = h * HL_DH + tm * 32 + r;
  const float* tp = p.t + (int64_t)h * p.t_sh + tn * 32 + r;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < p.B; k0 += 32) {     // four 8-deep steps (48 loads) in flight before their MFMAs: the loop is latency-, not rate-bound
    float a[16], w[16], sc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int b = k0 + 8 * (i >> 2) + 4 * hl + (i & 3);
      const bool in = b < p.B;
      const int bc = in ? b : 0;
      sc[i] = in ? (p.rs ? p.rs[(int64_t)bc * p.rs_ld + h] : 1.f) : 0.f;
      a[i] = p.x[(int64_t)bc * p.ldx + xcol];
      w[i] = tp[(int64_t)bc * p.t_sb];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i] * sc[i], w[i], acc, 0, 0, 0);
  }
  const int col = tn * 32 + r;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = h * HL_DH + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * hl;
    p.out[(int64_t)row * p.ldo + col] = acc[i];
  }
}

// softmax over the tokens of one (sample, head) column of s [B, N, ld] (fp32 scores, scale folded in here):
//   e[b, n, h] = exp(scale (s - max_n s)) as bf16 (the weights of the row-sum GEMM; columns h >= H of the padded rows are zeroed),
//   rz[b, h] = 1 / sum_n e (the sum of the ROUNDED weights, so that the weights the GEMM sees sum to one).   grid (B), block 1024
// Dropout on the probabilities (drop_p > 0; mask of xvit_dropout on a contiguous [B, H, N] tensor, the one xvit_cls_xattn_fwd applies):
//   e_m = the kept weights (the operand of the row-sum GEMM; e itself stays whole for the backward), and with inv = 1 / (1 - drop_p)
//   stat[0][b, h] = rz,  stat[1][b, h] = rz inv (row scale of Wv_h S),  stat[2][b, h] = rz inv sum_n e_m (weight of bv).
constexpr int SM_T = 1024;   // 64 rows x 16 columns per pass: the kernels are chains of dependent row passes, so more rows per pass = fewer trips
__global__ __launch_bounds__(SM_T) void cls_softmax_kernel(const float* __restrict__ s, int64_t ld, bf16* __restrict__ e, int64_t lde, float* __restrict__ rz,
                                                           int H, int N, float scale, bf16* __restrict__ e_m, float drop_p, uint64_t drop_seed_in,
                                                           const uint64_t* __restrict__ drop_epoch) {
  constexpr int NW = SM_T / 64, RP = SM_T / 16;
  __shared__ float red[NW][16];
  __shared__ float redm[NW][16];
  const bool drop = drop_p > 0.f;
  const uint64_t drop_seed = drop ? drop_seed_at(drop_seed_in, drop_epoch) : 0;
  const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sb = s + (int64_t)b * N * ld;
  bf16* eb = e + (int64_t)b * N * lde;
  bf16* emb = drop ? e_m + (int64_t)b * N * lde : nullptr;
  // thread t owns column t % 16 of rows t / 16, t / 16 + RP, ...: a row's 16 floats are one 64-byte segment
  const int col = tid & 15, r0 = tid >> 4;
  float mx = -INFINITY;
  if (col < H)
    for (int n = r0; n < N; n += RP) mx = fmaxf(mx, sb[(int64_t)n * ld + col]);
  // reduce over the threads of a column: lanes with equal (lane & 15) inside a wave, then the waves (fixed order)
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (lane < 16) red[wave][lane] = mx;
  __syncthreads();
  mx = red[0][col];
#pragma unroll
  for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w][col]);
  __syncthreads();
  float sum = 0.f, summ = 0.f;
  const float c = scale * 1.4426950408889634f;
  const uint64_t pidx = ((uint64_t)b * H + col) * (uint64_t)N;
  for (int n = r0; n < N; n += RP) {
    bf16 w = f2bf(0.f), wm = f2bf(0.f);
    if (col < H) {
      w = f2bf(__builtin_amdgcn_exp2f((sb[(int64_t)n * ld + col] - mx) * c));
      sum += bf2f(w);
      if (drop && (hash32(drop_seed, pidx + n) & 0xFFFFFFu) >= thr) { wm = w; summ += bf2f(w); }
    }
    if (col < lde) {
      eb[(int64_t)n * lde + col] = w;
      if (drop) emb[(int64_t)n * lde + col] = wm;
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  summ += __shfl_xor(summ, 16);
  summ += __shfl_xor(summ, 32);
  if (lane < 16) { red[wave][lane] = sum; redm[wave][lane] = summ; }
  __syncthreads();
  if (tid < H) {
    float t = red[0][tid], tm = redm[0][tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) { t += red[w][tid]; tm += redm[w][tid]; }
    const float z = 1.0f / t;
    const int64_t BH = (int64_t)gridDim.x * H;           // rz is stat[0] of a [3][B][H] block when dropout is on
    rz[(int64_t)b * H + tid] = z;
    if (drop) {
      const float zi = z / (1.0f - drop_p);
      rz[BH + (int64_t)b * H + tid] = zi;
      rz[2 * BH + (int64_t)b * H + tid] = zi * tm;
    }
  }
}

// backward of that softmax: p = e rz, ds = scale p (dp - sum_n p dp) -> coef[b, n, 0 .. H) = ds, coef[b, n, H .. 2 H) = p (fp32, the input
// of xvit_xattn_kv_dgrad) and ds_bf16[b, n, 0 .. ldb) (the weights of the row-sum GEMM that gives T; columns >= H zeroed).
// With dropout (drop_p > 0, the forward's mask m): dp is the gradient of the DROPPED probabilities, so dp~ = m dp / (1 - drop_p) takes
// its place in ds, and the second half of coef is p' = m p / (1 - drop_p) — the weights that met the values.
__global__ __launch_bounds__(SM_T) void cls_softmax_bwd_kernel(const bf16* __restrict__ e, int64_t lde, const float* __restrict__ rz, const float* __restrict__ dp, int64_t ldp,
                                                               float* __restrict__ coef, bf16* __restrict__ dsb, int64_t ldb, int H, int N, float scale,
                                                               float drop_p, uint64_t drop_seed_in, const uint64_t* __restrict__ drop_epoch) {
  constexpr int NW = SM_T / 64, RP = SM_T / 16;
  __shared__ float red[NW][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = tid & 15, r0 = tid >> 4;
  const bf16* eb = e + (int64_t)b * N * lde;
  const float* dpb = dp + (int64_t)b * N * ldp;
  const float z = col < H ? rz[(int64_t)b * H + col] : 0.f;
  const bool drop = drop_p > 0.f;
  const uint64_t drop_seed = drop ? drop_seed_at(drop_seed_in, drop_epoch) : 0;
  const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);
  const float inv = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint64_t pidx = ((uint64_t)b * H + col) * (uint64_t)N;
  auto mk = [&](int n) { return !drop || (hash32(drop_seed, pidx + n) & 0xFFFFFFu) >= thr ? inv : 0.f; };
  float dsum = 0.f;
  if (col < H)
    for (int n = r0; n < N; n += RP) dsum = fmaf(bf2f(eb[(int64_t)n * lde + col]) * z, mk(n) * dpb[(int64_t)n * ldp + col], dsum);
  dsum += __shfl_xor(dsum, 16);
  dsum += __shfl_xor(dsum, 32);
  if (lane < 16) red[wave][lane] = dsum;
  __syncthreads();
  dsum = red[0][col];
#pragma unroll
  for (int w = 1; w < NW; ++w) dsum += red[w][col];
  float* cb = coef + (int64_t)b * N * 2 * H;
  bf16* db = dsb + (int64_t)b * N * ldb;
  for (int n = r0; n < N; n += RP) {
    float ds = 0.f;
    if (col < H) {
      const float pr = bf2f(eb[(int64_t)n * lde + col]) * z, m = mk(n);
      ds = scale * pr * (m * dpb[(int64_t)n * ldp + col] - dsum);
      cb[(int64_t)n * 2 * H + col] = ds;
      cb[(int64_t)n * 2 * H + H + col] = pr * m;
    }
    if (col < ldb) db[(int64_t)n * ldb + col] = f2bf(ds);
  }
}

// out[j] = sum_b x[b, j] w[b, j / 64]: the gradient of bv when the weights in front of it do not sum to one (dropout on the probabilities).
// One thread per column, samples in order (bit-reproducible).   grid (ceil(d / 256)), block 256
__global__ __launch_bounds__(256) void head_bias_grad_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, int64_t ldw_, float* __restrict__ out,
                                                             int B, int d) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= d) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc = fmaf(x[(int64_t)b * ldx + j], w[(int64_t)b * ldw_ + (j >> 6)], acc);
  out[j] = acc;
}

}  // namespace xvit

using namespace xvit;

static int head_check(const char* who, int B, int H, int d) {
  XVIT_REQUIRE(B > 0 && H > 0 && H <= 65535 && d == H * HL_DH, "%s: need B, H > 0 and d == 64 H (got B=%d H=%d d=%d)", who, B, H, d);
  return XVIT_OK;
}

extern "C" int xvit_head_rows(const float* x, int64_t ldx, const float* W, int64_t ldw, float* out, int64_t out_sb, int64_t out_sh, void* out_bf16, int64_t ob_sb,
                              int64_t ob_sh, int ob_heads, int B, int H, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(x && W && out, "xvit_head_rows: null pointer");
  if (int e = head_check("xvit_head_rows", B, H, d)) return e;
  XVIT_REQUIRE(ldx % 4 == 0 && ldx >= d && ldw >= d && ((uintptr_t)x & 15) == 0, "xvit_head_rows: ldx must be a multiple of 4 and >= d, ldw >= d, x 16-byte aligned");
  HeadParams p = {};
  p.x = x; p.W = W; p.out = out; p.out_bf16 = (bf16*)out_bf16;
  p.ldx = ldx; p.ldw = ldw; p.out_sb = out_sb; p.out_sh = out_sh; p.ob_sb = ob_sb; p.ob_sh = ob_sh;
  p.B = B; p.H = H; p.d = d;
  const int heads = out_bf16 && ob_heads > H ? ob_heads : H;     // the bf16 copy may have more (zero) head rows than H: a 16-row GEMM operand
  XVIT_REQUIRE(heads <= 65535, "xvit_head_rows: too many head rows");
  hipLaunchKernelGGL(head_rows_kernel, dim3(((B + 31) / 32) * (d / 32), heads), dim3(64), 0, (hipStream_t)stream, p);
  return check_launch("xvit_head_rows");
}

extern "C" int xvit_head_cols(const float* t, int64_t t_sb, int64_t t_sh, const float* W, int64_t ldw, const float* row_scale, int64_t rs_ld, const float* bias,
                              const float* bias_scale, int64_t bsc_ld, float* out, int64_t ldo, void* out_bf16, int64_t ldob, int B, int H, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(t && W && out, "xvit_head_cols: null pointer");
  if (int e = head_check("xvit_head_cols", B, H, d)) return e;
  XVIT_REQUIRE(t_sb % 4 == 0 && t_sh % 4 == 0 && ldw % 4 == 0 && ldw >= d && ((uintptr_t)t & 15) == 0 && ((uintptr_t)W & 15) == 0 && ldo >= d,
               "xvit_head_cols: strides must be multiples of 4 floats, t and W 16-byte aligned, ldo >= d");
  HeadParams p = {};
  p.t = t; p.W = W; p.rs = row_scale; p.bias = bias; p.bias_scale = bias ? bias_scale : nullptr; p.out = out; p.out_bf16 = (bf16*)out_bf16;
  p.t_sb = t_sb; p.t_sh = t_sh; p.ldw = ldw; p.rs_ld = rs_ld; p.bsc_ld = bsc_ld; p.ldo = ldo; p.ob_sb = ldob;
  p.B = B; p.H = H; p.d = d;
  hipLaunchKernelGGL(head_cols_kernel, dim3(((B + 31) / 32) * 2, H), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("xvit_head_cols");
}

extern "C" int xvit_head_wgrad(const float* x, int64_t ldx, const float* t, int64_t t_sb, int64_t t_sh, const float* row_scale, int64_t rs_ld, float* dW, int64_t lddw,
                               int B, int H, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(x && t && dW, "xvit_head_wgrad: null pointer");
  if (int e = head_check("xvit_head_wgrad", B, H, d)) return e;
  XVIT_REQUIRE(ldx >= d && lddw >= d, "xvit_head_wgrad: ldx / lddw < d");
  HeadParams p = {};
  p.x = x; p.t = t; p.rs = row_scale; p.out = dW;
  p.ldx = ldx; p.t_sb = t_sb; p.t_sh = t_sh; p.rs_ld = rs_ld; p.ldo = lddw;
  p.B = B; p.H = H; p.d = d;
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(2 * (d / 32), H), dim3(64), 0, (hipStream_t)stream, p);
  return check_launch("xvit_head_wgrad");
}

extern "C" int xvit_cls_softmax_fwd(const float* s, int64_t lds, void* e_bf16, int64_t lde, float* rz, int B, int H, int N, float scale, void* e_masked_bf16,
                                    float dropout_p, uint64_t dropout_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(s && e_bf16 && rz, "xvit_cls_softmax_fwd: null pointer");
  XVIT_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 16 && lds >= H && lde >= H && lde <= 16, "xvit_cls_softmax_fwd: need H <= 16, lds >= H, H <= lde <= 16 (B=%d H=%d N=%d)", B, H, N);
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || e_masked_bf16), "xvit_cls_softmax_fwd: dropout_p must be in [0, 1) and needs e_masked");
  hipLaunchKernelGGL(cls_softmax_kernel, dim3(B), dim3(SM_T), 0, (hipStream_t)stream, s, lds, (bf16*)e_bf16, lde, rz, H, N, scale, (bf16*)e_masked_bf16, dropout_p,
                     dropout_seed, dropout_p > 0.f ? drop_epoch_ptr() : nullptr);
  return check_launch("xvit_cls_softmax_fwd");
}

extern "C" int xvit_cls_softmax_bwd(const void* e_bf16, int64_t lde, const float* rz, const float* dp, int64_t ldp, float* coef, void* ds_bf16, int64_t ldb, int B, int H,
                                    int N, float scale, float dropout_p, uint64_t dropout_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(e_bf16 && rz && dp && coef && ds_bf16, "xvit_cls_softmax_bwd: null pointer");
  XVIT_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 16 && lde >= H && ldp >= H && ldb >= H && ldb <= 16, "xvit_cls_softmax_bwd: need H <= 16, lde, ldp >= H, H <= ldb <= 16");
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "xvit_cls_softmax_bwd: dropout_p must be in [0, 1)");
  hipLaunchKernelGGL(cls_softmax_bwd_kernel, dim3(B), dim3(SM_T), 0, (hipStream_t)stream, (const bf16*)e_bf16, lde, rz, dp, ldp, coef, (bf16*)ds_bf16, ldb, H, N, scale,
                     dropout_p, dropout_seed, dropout_p > 0.f ? drop_epoch_ptr() : nullptr);
  return check_launch("xvit_cls_softmax_bwd");
}

extern "C" int xvit_head_bias_grad(const float* x, int64_t ldx, const float* w, int64_t ldw, float* out, int B, int H, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(x && w && out, "xvit_head_bias_grad: null pointer");
  if (int e = head_check("xvit_head_bias_grad", B, H, d)) return e;
  XVIT_REQUIRE(ldx >= d && ldw >= H, "xvit_head_bias_grad: ldx < d or ldw < H");
  hipLaunchKernelGGL(head_bias_grad_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, w, ldw, out, B, d);
  return check_launch("xvit_head_bias_grad");
}
