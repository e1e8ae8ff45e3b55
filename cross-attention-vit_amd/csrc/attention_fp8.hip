// MX-fp8 forward attention for gfx950 (SURVEY.md section 8 row "fp8 MFMA QK^T / AV path", BASELINE.json configs[4]:
// patch 8^3 -> N = 4097).  Both products run on v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 operands, one e8m0 scale per
// lane per 32 contraction elements): 2x the bf16 MFMA rate, 4 instructions per 64-key tile instead of 16.
//
//   1. attn_quant_fp8_kernel: q, k (bf16, [B, N, H*64] strided) -> e4m3 rows [B, H, Np, 64] with one scale per (row, 32 d);
//      v -> TRANSPOSED e4m3 [B, H, 64, Np] with one scale per (d, 32 keys): the P.V product contracts over keys, so its A
//      operand (V^T rows) is then a plain contiguous read and the scale block lies along the contraction, as MX requires.
//      Scale = 2^ceil(log2(amax / 448)): no saturation (v_cvt_pk_fp8_f32 turns |x| > 448 into NaN, it does not clamp).
//   2. attn_fwd_fp8_kernel: the structure of attn_fwd_kernel (4 waves x 32 queries, query on the MFMA lane, online softmax in
//      fp32, K / V^T tiles by LDS-DMA into a 2-stage ring).  S^T = K Q^T is ONE instruction per 32 keys (K = 64 = d_h).
//      P is quantised as p * 256 with the fixed block scale 2^-8 (p <= 1: the e4m3 grid then reaches down to 2^-17) and
//      rearranged into the B-operand order with one v_permlane32_swap per 8 keys (see `pack_p`).
// Operand lane map (tools/mfma_scale_probe.hip, measured on the device): lane (r = l & 31, h = l >> 5) holds row / column r; its
// bytes 0..15 are k = 16 h + 0..15 and its bytes 16..31 are k = 32 + 16 h + 0..15 (the instruction is two K = 32 steps), while the
// lane's scale byte applies to the CONSECUTIVE block k = 32 h .. 32 h + 31 — i.e. to bytes 0..15 of both lanes of a row for h = 0
// and to bytes 16..31 of both for h = 1.  So a lane loads chunks h and 2 + h of a 64-byte row, and the scale of block h.
// Accuracy is that of 3-bit-mantissa operands: see DESIGN.md section 7 and tests/test_attn_fp8_gpu.py for the stated budget;
// the backward stays on the bf16 kernels.
#include "xvit_common.h"

namespace xvit {
namespace fp8 {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
constexpr int DH = 64, TILE = 64;
constexpr int K_IMG = TILE * DH;          // 4 KiB: [64 keys][64 B]
constexpr int STAGE = 2 * K_IMG;          // K image | V^T image ([64 d][64 B of keys])
constexpr int NST = 2;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int P_SCALE = 119;              // e8m0 of 2^-8

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// e8m0 byte b with amax / 2^(b - 127) <= 448 (0 for an all-zero block: its products vanish anyway)
__device__ __forceinline__ int block_scale(float amax, float& inv) {
  if (!(amax > 0.f)) { inv = 0.f; return 0; }
  int e;
  const float m = frexpf(amax * (1.0f / 448.0f), &e);   // amax / 448 = m 2^e, m in [0.5, 1)
  if (m == 0.5f) e -= 1;                                 // exactly a power of two
  e = max(-126, min(127, e));
  inv = ldexpf(1.0f, -e);
  return e + 127;
}
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// one thread per (b, h, row, 32-wide d block) of q and k; rows >= N are zero-filled (scale byte 0)
__global__ void quant_rows_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, int64_t sb, int64_t sn, uint8_t* __restrict__ q8, uint8_t* __restrict__ k8,
                                  uint8_t* __restrict__ qs, uint8_t* __restrict__ ks, int H, int N, int Np, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int blk = (int)(gid & 1);
  int64_t r = gid >> 1;
  const int n = (int)(r % Np); r /= Np;
  const int head = (int)(r % H);
  const int b = (int)(r / H);
  const int which = blockIdx.y;   // 0: q, 1: k
  const bf16* src = (which ? k : q) + (int64_t)b * sb + (int64_t)n * sn + head * DH + blk * 32;
  uint8_t* dst = (which ? k8 : q8) + (((int64_t)b * H + head) * Np + n) * DH + blk * 32;
  uint8_t* sdst = (which ? ks : qs) + (((int64_t)b * H + head) * Np + n) * 2 + blk;
  float v[32];
  float amax = 0.f;
  if (n < N) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 x = *(const bf16x8*)(src + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[c * 8 + j] = bf2f(x[j]); amax = fmaxf(amax, fabsf(v[c * 8 + j])); }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = 0.f;
  }
  float inv;
  const int sbyte = block_scale(amax, inv);
  uint32_t w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) w[c] = pack4_fp8(v[4 * c] * inv, v[4 * c + 1] * inv, v[4 * c + 2] * inv, v[4 * c + 3] * inv);
  *(u32x4v*)dst = u32x4v{w[0], w[1], w[2], w[3]};
  *(u32x4v*)(dst + 16) = u32x4v{w[4], w[5], w[6], w[7]};
  *sdst = (uint8_t)sbyte;
}

// one thread per (b, h, 32-key block, d): v8t[b, h, d, kb*32 .. +32), vs[b, h, d, kb]
__global__ void quant_vt_kernel(const bf16* __restrict__ v, int64_t sb, int64_t sn, uint8_t* __restrict__ v8t, uint8_t* __restrict__ vs, int H, int N, int Np, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int d = (int)(gid & 63);
  int64_t r = gid >> 6;
  const int nkb = Np >> 5;
  const int kb = (int)(r % nkb); r /= nkb;
  const int head = (int)(r % H);
  const int b = (int)(r / H);
  const bf16* src = v + (int64_t)b * sb + head * DH + d;
  float x[32];
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int n = kb * 32 + j;
    x[j] = n < N ? bf2f(src[(int64_t)n * sn]) : 0.f;
    amax = fmaxf(amax, fabsf(x[j]));
  }
  float inv;
  const int sbyte = block_scale(amax, inv);
  uint32_t w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) w[c] = pack4_fp8(x[4 * c] * inv, x[4 * c + 1] * inv, x[4 * c + 2] * inv, x[4 * c + 3] * inv);
  uint8_t* dst = v8t + (((int64_t)b * H + head) * DH + d) * Np + kb * 32;
  *(u32x4v*)dst = u32x4v{w[0], w[1], w[2], w[3]};
  *(u32x4v*)(dst + 16) = u32x4v{w[4], w[5], w[6], w[7]};
  vs[(((int64_t)b * H + head) * DH + d) * nkb + kb] = (uint8_t)sbyte;
}

struct BlockCoord { int x, head, b; };
__device__ __forceinline__ BlockCoord xcd_block_coord() {   // as attention.hip: the query blocks of one (b, h) share an XCD's L2
  const int nx = gridDim.x, nh = gridDim.y, total = nx * nh * gridDim.z;
  const int lin = blockIdx.x + nx * (blockIdx.y + nh * blockIdx.z);
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  BlockCoord c;
  c.x = logical % nx;
  const int rest = logical / nx;
  c.head = rest % nh;
  c.b = rest / nh;
  return c;
}

// [64 rows][64 B] images: 16-byte chunk c of row r sits at chunk position c ^ ((r >> 2) & 3): the 16 rows one ds_read_b128
// lane group touches then spread over all 64 banks
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

// P^T as the B operand of O^T += V^T P^T.  In the S^T accumulators lane (q, hh) holds the keys 32 kb + 8 g + 4 hh + e (register
// 4 g + e of block kb).  Lane (q, h) of the B operand wants keys 32 kb + 16 h + 0..15 in bytes 16 kb + 0..15, i.e. dword
// 4 kb + 2 (g - 2 h) + hh for g in {2 h, 2 h + 1} from BOTH halves hh: its own registers of those two g and the partner lane's.
// One v_permlane32_swap(X = own dword of (kb, gA), Y = own dword of (kb, gA + 2)), gA in {0, 1}, leaves dwords 4 kb + 2 gA and
// 4 kb + 2 gA + 1 of the wanted operand on both halves: X' = (X.lo | Y.lo), Y' = (X.hi | Y.hi).
__device__ __forceinline__ i32x8 pack_p(const f32x16& p0, const f32x16& p1) {
  i32x8 b;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const f32x16& p = kb ? p1 : p0;
#pragma unroll
    for (int ga = 0; ga < 2; ++ga) {
      const uint32_t x = pack4_fp8(p[4 * ga], p[4 * ga + 1], p[4 * ga + 2], p[4 * ga + 3]);
      const uint32_t y = pack4_fp8(p[4 * (ga + 2)], p[4 * (ga + 2) + 1], p[4 * (ga + 2) + 2], p[4 * (ga + 2) + 3]);
      const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
      b[4 * kb + 2 * ga] = (int)r[0];
      b[4 * kb + 2 * ga + 1] = (int)r[1];
    }
  }
  return b;
}

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

__global__ __launch_bounds__(256, 4) void attn_fwd_fp8_kernel(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8, const uint8_t* __restrict__ v8t,
                                                              const uint8_t* __restrict__ qs, const uint8_t* __restrict__ ks, const uint8_t* __restrict__ vs,
                                                              bf16* __restrict__ o, int64_t osb, int64_t osn, float* __restrict__ lse, int H, int N, int Np,
                                                              float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;   // [NST][K image | V^T image]
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int64_t bh = (int64_t)b * H + head;
  const int q0 = bc.x * 128 + wave * 32;
  const int ntiles = Np / TILE;
  const int nkb = Np >> 5;

  // tile loaders: wave w moves the w-th KiB of each image = rows 16 w .. 16 w + 15, one LDS-DMA instruction each
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(k8 + bh * Np * DH, (uint32_t)(Np * DH));
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(v8t + bh * DH * Np, (uint32_t)(DH * Np));
  uint32_t voff_k, voff_v;
  {
    const int row = wave * 16 + (lane >> 2), chunk = (lane & 3) ^ swz(row);
    voff_k = (uint32_t)(row * DH + chunk * 16);          // K rows are 64 B apart, tiles 4 KiB apart
    voff_v = (uint32_t)(row * Np + chunk * 16);          // V^T rows are Np bytes apart, tiles 64 B apart
  }
  auto issue = [&](int stage, int t) {
    glds16(rk, smem + stage * STAGE + wave * 1024, voff_k, (uint32_t)t * K_IMG);
    glds16(rv, smem + stage * STAGE + K_IMG + wave * 1024, voff_v, (uint32_t)t * TILE);
  };
  issue(0, 0);

  // the wave's queries: lane (query r, half h) holds d = 16 h + 0..15 and 32 + 16 h + 0..15, and the scale of the block d = 32 h .. + 31
  const int qrow = q0 + r;
  i32x8 qf;
  int qsc;
  {
    const int qr = min(qrow, Np - 1);
    const u32x4v* src = (const u32x4v*)(q8 + (bh * Np + qr) * DH);
    const u32x4v lo = src[h], hi = src[2 + h];          // d = 16 h .. + 15 and 32 + 16 h .. + 15
    qf = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    qsc = qs[(bh * Np + qr) * 2 + h];
  }
  // per-lane LDS offsets of the fragment reads: row r (+32 per block), source chunks h and 2 + h
  uint32_t foff[2][2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int row = r + 32 * blk;
#pragma unroll
    for (int c = 0; c < 2; ++c) foff[blk][c] = (uint32_t)(row * 64 + (((2 * c + h) ^ swz(row)) << 4));
  }
  const uint8_t* ks_lane = ks + bh * Np * 2 + r * 2 + h;            // + (64 t + 32 kb) * 2
  const uint8_t* vs_lane = vs + (bh * DH + r) * nkb + h;            // + 32 db * nkb + 2 t

  const float c = scale * LOG2E;
  const bool wave_active = q0 < N;
  float m_run = -INFINITY, l_run = 0.f;   // l_run sums p * 256
  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }

  int ksc[2] = {ks_lane[0], ks_lane[64]}, vsc[2] = {vs_lane[0], vs_lane[32 * nkb]};   // tile 0's scales
  int stage = 0;
  for (int t = 0; t < ntiles; ++t) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 1 < ntiles) issue(stage ^ 1, t + 1);
    int ksn[2] = {0, 0}, vsn[2] = {0, 0};
    if (t + 1 < ntiles) {   // next tile's scale bytes (plain loads: the compiler tracks them)
      ksn[0] = ks_lane[(64 * (t + 1)) * 2]; ksn[1] = ks_lane[(64 * (t + 1) + 32) * 2];
      vsn[0] = vs_lane[2 * (t + 1)]; vsn[1] = vs_lane[32 * nkb + 2 * (t + 1)];
    }
    const XVIT_LDS char* kimg = smem + stage * STAGE;
    const XVIT_LDS char* vimg = kimg + K_IMG;
    stage ^= 1;
    if (wave_active) {
      const int valid = N - t * TILE;
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const u32x4v lo = *(const XVIT_LDS u32x4v*)(kimg + foff[kb][0]), hi = *(const XVIT_LDS u32x4v*)(kimg + foff[kb][1]);
        const i32x8 kf = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, z, 0, 0, 0, ksc[kb], 0, qsc);   // S^T[key][query]
      }
      if (valid < TILE) {   // keys past N (zero rows in the image) must not enter the softmax
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kb * 32 + acc_row(i, h) >= valid) s[kb][i] = -INFINITY;
      }
      float mx = s[0][0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
      mx = half_max(mx);
      if (__any(mx > m_run)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
      }
      const float nmc = 8.0f - m_run * c;   // p * 256 = exp2(s c - m c + 8)
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[kb][i] = __builtin_amdgcn_exp2f(fmaf(s[kb][i], c, nmc));
          psum += s[kb][i];
        }
      l_run += psum;
      const i32x8 pf = pack_p(s[0], s[1]);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const u32x4v lo = *(const XVIT_LDS u32x4v*)(vimg + foff[db][0]), hi = *(const XVIT_LDS u32x4v*)(vimg + foff[db][1]);
        const i32x8 vf = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        oacc[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, oacc[db], 0, 0, 0, vsc[db], 0, P_SCALE);   // O^T[d][query]
      }
    }
    ksc[0] = ksn[0]; ksc[1] = ksn[1]; vsc[0] = vsn[0]; vsc[1] = vsn[1];
  }
  if (!wave_active) return;
  const float l_tot = half_sum(l_run);   // 256 x the softmax denominator
  const bool valid = qrow < N;
  if (valid && h == 0) lse[bh * N + qrow] = m_run * scale + __logf(l_tot) - 5.545177444479562f;   // - ln 256
  if (valid) {
    const float mul = 256.0f / l_tot;   // the MFMA applied P's block scale 2^-8, so oacc = sum p v; the denominator is l_tot / 256
    bf16* dst = o + (int64_t)b * osb + head * DH + (int64_t)qrow * osn;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(oacc[db][g * 4 + e] * mul);
        *(bf16x4*)(dst + db * 32 + 8 * g + 4 * h) = ov;
      }
  }
}

}  // namespace fp8
}  // namespace xvit

using namespace xvit;

static int64_t fp8_np(int N) { return ((int64_t)N + 63) / 64 * 64; }

extern "C" int64_t xvit_attn_fp8_workspace_bytes(int B, int H, int N, int dh) {
  if (B <= 0 || H <= 0 || N <= 0 || dh != 64) return 0;
  const int64_t Np = fp8_np(N), bh = (int64_t)B * H;
  // q8 | k8 | v8t (bh * Np * 64 each) | qs | ks (bh * Np * 2 each) | vs (bh * 64 * Np / 32), each region 256-byte aligned
  auto al = [](int64_t x) { return (x + 255) / 256 * 256; };
  return 3 * al(bh * Np * 64) + 2 * al(bh * Np * 2) + al(bh * 64 * (Np / 32));
}

extern "C" int xvit_attn_fwd_fp8(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, void* o, int64_t osb, int64_t osn, float* lse, int B, int H,
                                 int N, int dh, float scale, void* workspace, int64_t workspace_bytes, xvit_stream_t stream) {
  XVIT_REQUIRE(q && k && v && o && lse && workspace, "xvit_attn_fwd_fp8: null pointer");
  XVIT_REQUIRE(dh == 64, "xvit_attn_fwd_fp8: head dim %d unsupported (only 64)", dh);
  XVIT_REQUIRE(B > 0 && H > 0 && N > 0 && B <= 65535 && H <= 65535, "xvit_attn_fwd_fp8: bad B/H/N (%d,%d,%d)", B, H, N);
  XVIT_REQUIRE(sn % 8 == 0 && sb % 8 == 0 && osn % 4 == 0 && osb % 4 == 0, "xvit_attn_fwd_fp8: strides must be multiples of 8 (inputs) / 4 (output) elements");
  XVIT_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "xvit_attn_fwd_fp8: q, k must be 16-byte and the workspace 256-byte aligned");
  const int64_t need = xvit_attn_fp8_workspace_bytes(B, H, N, dh);
  XVIT_REQUIRE(workspace_bytes >= need, "xvit_attn_fwd_fp8: needs %lld bytes of workspace (got %lld)", (long long)need, (long long)workspace_bytes);
  const int64_t Np = fp8_np(N), bh = (int64_t)B * H;
  XVIT_REQUIRE(Np * 64 < (1ll << 31), "xvit_attn_fwd_fp8: sequence too long");
  auto al = [](int64_t x) { return (x + 255) / 256 * 256; };
  uint8_t* base = (uint8_t*)workspace;
  uint8_t* q8 = base; uint8_t* k8 = q8 + al(bh * Np * 64); uint8_t* v8t = k8 + al(bh * Np * 64);
  uint8_t* qs = v8t + al(bh * Np * 64); uint8_t* ks = qs + al(bh * Np * 2); uint8_t* vs = ks + al(bh * Np * 2);
  hipStream_t s = (hipStream_t)stream;
  {
    const int64_t total = bh * Np * 2;
    hipLaunchKernelGGL(fp8::quant_rows_kernel, dim3((unsigned)((total + 255) / 256), 2), dim3(256), 0, s, (const bf16*)q, (const bf16*)k, sb, sn, q8, k8, qs, ks, H, N, (int)Np, total);
    const int64_t tv = bh * (Np / 32) * 64;
    hipLaunchKernelGGL(fp8::quant_vt_kernel, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, s, (const bf16*)v, sb, sn, v8t, vs, H, N, (int)Np, tv);
  }
  const dim3 grid((N + 127) / 128, H, B), block(256);
  hipLaunchKernelGGL(fp8::attn_fwd_fp8_kernel, grid, block, fp8::NST * fp8::STAGE, s, q8, k8, v8t, qs, ks, vs, (bf16*)o, osb, osn, lse, H, N, (int)Np, scale);
  return check_launch("xvit_attn_fwd_fp8");
}
