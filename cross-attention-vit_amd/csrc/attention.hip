// Fused (flash-style) self-attention for gfx950, head dim 64, bf16 in / fp32 accumulate.
//
// All three kernels put the softmax ROW INDEX ON THE MFMA LANE ("swapped" products), so row
// statistics (max, sum, lse, delta) are lane-local scalars and every second product takes the
// first product's accumulator directly as its B operand (no LDS round trip, no shuffles):
//   fwd : S^T = K Q^T   -> P^T (in regs) -> O^T += V^T P^T          lane = query
//   dQ  : S^T = K Q^T, dP^T = V dO^T -> dS^T -> dQ^T += K^T dS^T    lane = query
//   dKV : S = Q K^T, dP = dO V^T -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS   lane = key
// The lane-side operand (Q / dO, or K / V) is loaded once from HBM into registers; the streamed
// operand tiles (64 rows x 64 d, 8 KiB) arrive by LDS-DMA into a double-buffered, XOR-swizzled
// image that serves both ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads
// conflict-free.  v_mfma_f32_32x32x16_bf16 throughout.  No [N,N] tensor ever reaches HBM; the
// backward recomputes P from the saved log-sum-exp.  dQ and dK/dV are separate kernels, so no
// atomics and bit-reproducible gradients.
#include "xvit_common.h"

namespace xvit {

constexpr int DH = 64;          // head dim
constexpr int TILE_ROWS = 64;   // streamed rows per iteration
constexpr int IMG_BYTES = TILE_ROWS * DH * 2;  // 8 KiB
constexpr float LOG2E = 1.4426950408889634f;
#ifndef XVIT_FWD_NST
#define XVIT_FWD_NST 2
#endif
constexpr int FWD_NST = XVIT_FWD_NST;     // K/V ring depth of the forward kernel.  2 stages = 32 KiB of LDS: four blocks per CU (the VGPR budget
                                          // of 128 allows exactly that) hide each other's prologues; 3 stages (3 blocks per CU) measured 3-4.5 % slower

// chunk swizzle for a [rows][128 B] image: conflict-free for b128 row reads and tr reads
__device__ __forceinline__ int swz_img(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// LDS-DMA loader for one 64x64 bf16 tile whose rows are `stride_n` elements apart in HBM.
struct TileLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[2];
  uint32_t step;
  // base: element (row 0, col 0) of the streamed matrix for this (b, h); rows_total valid rows
  __device__ __forceinline__ void init(const bf16* base, int64_t stride_n, int rows_total, int wave, int lane) {
    rsrc = make_rsrc(base, clamp_bytes(((int64_t)(rows_total - 1) * stride_n + DH) * 2));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 8 + (lane >> 3);  // 8 pieces of 8 rows; wave handles 2
      const int chunk = (lane & 7) ^ swz_img(row);
      voff[j] = (uint32_t)(row * stride_n * 2 + chunk * 16);
    }
    step = (uint32_t)(TILE_ROWS * stride_n * 2);
  }
  __device__ __forceinline__ void issue(XVIT_LDS char* image, int wave, int tile) const {
    const uint32_t soff = (uint32_t)tile * step;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(rsrc, image + (wave * 2 + j) * 1024, voff[j], soff);
  }
};

// Per-lane LDS offsets for the two read kinds on a [64][128 B] swizzled image.
struct ImgReader {
  uint32_t row_off[4];  // b128 row read, per k-step over d (16 each)
  uint32_t tr_off[2][2];  // tr read, per 32-wide d block, per 8-row half (the swizzle differs)
  __device__ __forceinline__ void init(int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row_off[ks] = (uint32_t)(r * 128 + (((ks * 2 + h) ^ swz_img(r)) << 4));
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = g >> 1;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = 8 * half + 4 * hh + q;
        const int chunk = db * 4 + 2 * (g & 1) + (p >> 1);
        tr_off[db][half] = (uint32_t)(row * 128 + ((chunk ^ swz_img(row)) << 4) + 8 * (p & 1));
      }
  }
  // A operand, natural k order: rows rb*32 .. +31 of the image, k = d in [16ks, 16ks+16)
  __device__ __forceinline__ bf16x8 row_frag(const XVIT_LDS char* img, int rb, int ks) const {
    return *(const XVIT_LDS bf16x8*)(img + row_off[ks] + rb * 32 * 128);
  }
  // A operand = image^T: output rows d in [32db, 32db+32), k = image rows rb*32 + 16s + {accumulator order}
  __device__ __forceinline__ bf16x8 tr_frag(const XVIT_LDS char* img, int db, int rb, int s) const {
    const int base = (rb * 32 + 16 * s) * 128;  // +16/+32 rows leave the swizzle unchanged
    const s16x4 lo = lds_read_tr16(img + tr_off[db][0] + base);
    const s16x4 hi = lds_read_tr16(img + tr_off[db][1] + base);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
};

// registers 8s..8s+7 of a 32x32 accumulator as the bf16 B operand of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(x[8 * s + j]);
  return o;
}

// Lane-side operand: 32 rows (one per lane&31) x 64 d from HBM into 4 B-operand fragments.
__device__ __forceinline__ void load_lane_operand(bf16x8 (&f)[4], const bf16* base, int64_t stride_n, int row0, int rows_total, int lane) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, clamp_bytes(((int64_t)(rows_total - 1) * stride_n + DH) * 2));
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    f[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(((int64_t)(row0 + r) * stride_n + ks * 16 + h * 8) * 2), 0, 0));
  }
}

// Make hipcc's vmcnt scoreboard see these (ordinary) loads as COMPLETE here: the empty asm consumes the
// registers, so the compiler waits for them now — otherwise it re-waits at their first use inside the tile
// loop with a conservative vmcnt(0..3), which would also drain the LDS-DMA ring every iteration.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void settle(bf16x8 (&f)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32x4 t = __builtin_bit_cast(u32x4, f[i]);
    asm volatile("" : "+v"(t));
    f[i] = __builtin_bit_cast(bf16x8, t);
  }
}

// Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), each XCD with its own L2.
// Remap the linear id so that a contiguous range of LOGICAL ids runs on one XCD: the query (or key) blocks of
// one (batch, head) then share an L2 and its K/V (Q/dO) tiles are fetched from HBM once, not once per XCD
// (measured before the remap: 350-390 MB fetched per launch against ~100 MB of q/k/v).
struct BlockCoord { int x, head, b; };
__device__ __forceinline__ BlockCoord xcd_block_coord() {
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

// counted wait on the vector-memory counter (the LDS-DMA ring: "at most N loads still in flight")
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// top-of-iteration wait of an NST-deep ring whose waves issue PT DMA instructions per tile: tile t must have
// landed, the (up to NST-2) younger tiles may stay in flight
template <int NST, int PT>
__device__ __forceinline__ void ring_wait(int t, int ntiles) {
  const int ahead = min(NST - 2, ntiles - 1 - t);
  if (NST >= 4 && ahead == 2) wait_vmcnt<2 * PT>();
  else if (NST >= 3 && ahead >= 1) wait_vmcnt<PT>();
  else wait_vmcnt<0>();
}
#ifndef XVIT_BWD_NST
#define XVIT_BWD_NST 2
#endif
constexpr int BWD_NST = XVIT_BWD_NST;      // K/V (dQ kernel) and Q/dO (dK/dV kernel) ring depth of the backward kernels.
                                         // Same-box A/B at N = 512..4097: depth 2, 3 and 4 are within noise (the loops are not load-latency-bound)

#define ZERO16 (f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f})

// element index inside a 32-row accumulator block held in register i by lane half h
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// store a [64 d][32 rows-on-lane] accumulator pair as bf16 rows of a [*, stride_n] matrix
__device__ __forceinline__ void store_lane_rows(const f32x16 (&acc)[2], float mul, bf16* base, int64_t stride_n, int row, bool valid, int lane) {
  if (!valid) return;
  const int h = lane >> 5;
  bf16* dst = base + (int64_t)row * stride_n;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[db][g * 4 + e] * mul);
      *(bf16x4*)(dst + db * 32 + 8 * g + 4 * h) = o;
    }
}

// Probability dropout (reference model.py:169: attention_probs = attn_dropout(softmax(scores))): element (b, h, query, key) is
// kept iff hash32(seed, ((b H + h) N + query) N + key) >= p 2^24 — the mask xvit_dropout applies to a contiguous [B, H, N, N]
// tensor with the same seed — and scaled by 1/(1-p).  The row sums (softmax normaliser) use the probabilities BEFORE the
// mask; the backward kernels regenerate the mask (nothing is stored) and, with O = P_drop V, delta = rowsum(dO O) is unchanged.
struct DropArgs { uint32_t thr; float inv; uint64_t seed; };
__device__ __forceinline__ bool drop_keep(const DropArgs& d, uint64_t bh_base, int query, int key, int N) {
  return (hash32(d.seed, bh_base + (uint64_t)query * (uint64_t)N + (uint64_t)key) & 0xFFFFFFu) >= d.thr;
}

// ------------------------------------------------------------------------------------------
// forward.  Each wave owns QB blocks of 32 queries (QB = 2: 64 queries per wave, 256 per workgroup).  The
// per-wave critical path of one tile (QK^T chain -> row max -> exp -> P.V chain) is latency-bound, so two
// independent query blocks per wave double the instruction-level parallelism, and every K / V fragment
// read from LDS feeds 2*QB MFMAs instead of 2.
// ------------------------------------------------------------------------------------------
// -DXVIT_KNOCKOUT_HALF_MFMA (diagnostic build, WRONG results): the forward kernel issues half of its QK^T and P.V MFMAs —
// the matrix-pipe time an MX-fp8 (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 rate) version of both products would have,
// with none of its extra quantisation work.  Timing it bounds what fp8 QK^T / AV could gain (DESIGN.md section 7, configs[4]).
#ifdef XVIT_KNOCKOUT_HALF_MFMA
constexpr int FWD_KS = 2, FWD_SS = 1;
#else
constexpr int FWD_KS = 4, FWD_SS = 2;
#endif

template <int QB, bool DROP>
__global__ __launch_bounds__(256, QB == 1 ? 4 : 2) void attn_fwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                          int64_t sb, int64_t sn, bf16* __restrict__ o, int64_t osb, int64_t osn,
                                                          float* __restrict__ lse, int H, int N, float scale, const DropArgs drop) {
#ifdef XVIT_DEBUG_ATTN_TIMES
  const uint64_t wc_entry = wall_clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;  // [FWD_NST stages][K image | V image]
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int q0 = bc.x * (128 * QB) + wave * (32 * QB);
  const int64_t off = (int64_t)b * sb + head * DH;
  const int ntiles = (N + TILE_ROWS - 1) / TILE_ROWS;

  // K/V tiles stream through a FWD_NST-deep LDS ring.  Loads stay in flight ACROSS iterations: each wave
  // issues exactly 4 LDS-DMA instructions per tile (2 K + 2 V), so "tile t has landed, tile t+1 may still be
  // in flight" is a counted s_waitcnt vmcnt(4); barriers are raw s_barrier (a __syncthreads() would drain
  // vmcnt to 0 and serialise every iteration behind a full HBM/L2 round trip).
  TileLoader lk, lv;
  lk.init(k + off, sn, N, wave, lane);
  lv.init(v + off, sn, N, wave, lane);
#pragma unroll
  for (int st = 0; st < FWD_NST - 1; ++st)
    if (st < ntiles) {
      lk.issue(smem + st * 2 * IMG_BYTES, wave, st);
      lv.issue(smem + st * 2 * IMG_BYTES + IMG_BYTES, wave, st);
    }
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    load_lane_operand(qf[qb], q + off, sn, q0 + qb * 32, N, lane);   // one round trip together with the first ring tiles
    settle(qf[qb]);                                                  // (drains the prologue's DMAs too: fine, they are needed first)
  }
  ImgReader rd;
  rd.init(lane);

  const int h = lane >> 5;
  const float c = scale * LOG2E;
  const bool wave_active = q0 < N;   // wave-uniform
  float m_run[QB], l_run[QB];
  f32x16 oacc[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[qb][0][i] = 0.f; oacc[qb][1][i] = 0.f; }
  }

#ifdef XVIT_DEBUG_ATTN_TIMES
  const uint64_t wc_loop = wall_clock64();
  uint64_t tk[5] = {0, 0, 0, 0, 0};
#define TK(i) if (t == 3) { tk[i] = __builtin_readcyclecounter(); }
#else
#define TK(i)
#endif
  int stage = 0;
  for (int t = 0; t < ntiles; ++t) {
    TK(0)
    ring_wait<FWD_NST, 4>(t, ntiles);   // tile t landed; up to FWD_NST - 2 younger tiles may be in flight
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t are in LDS; every wave is done with tile t-1
    TK(1)
    if (t + FWD_NST - 1 < ntiles) {
      int ns = stage + FWD_NST - 1;
      if (ns >= FWD_NST) ns -= FWD_NST;
      lk.issue(smem + ns * 2 * IMG_BYTES, wave, t + FWD_NST - 1);
      lv.issue(smem + ns * 2 * IMG_BYTES + IMG_BYTES, wave, t + FWD_NST - 1);
    }
    const XVIT_LDS char* kimg = smem + stage * 2 * IMG_BYTES;
    const XVIT_LDS char* vimg = kimg + IMG_BYTES;
    stage = stage + 1 == FWD_NST ? 0 : stage + 1;

    if (!wave_active) continue;   // this wave's queries are all past N: keep moving tiles and barriers, skip the math
    // keys of this tile past N: with <= 32 valid keys the second 32-key block is skipped entirely
    const int valid = N - t * TILE_ROWS;
    const bool two = valid > 32;

    // S^T[key][query] = K Q^T: every K fragment is read once and used by all QB query blocks
    f32x16 s[QB][2];
#pragma unroll
    for (int ks = 0; ks < FWD_KS; ++ks) {   // the first MFMA of each chain takes a literal-zero accumulator: no v_mov initialisation
      const bf16x8 kf = rd.row_frag(kimg, 0, ks);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[qb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], ks == 0 ? ZERO16 : s[qb][0], 0, 0, 0);
    }
    if (two) {
#pragma unroll
      for (int ks = 0; ks < FWD_KS; ++ks) {
        const bf16x8 kf = rd.row_frag(kimg, 1, ks);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], ks == 0 ? ZERO16 : s[qb][1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][1][i] = -INFINITY;
    }
    if (valid < TILE_ROWS) {  // mask the keys past N
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kb * 32 + acc_row(i, h) >= valid) s[qb][kb][i] = -INFINITY;
    }
#ifdef XVIT_DEBUG_ATTN_TIMES
    if (t == 3) { asm volatile("s_nop 0" ::"v"(s[0][0][0]), "v"(s[0][1][15])); tk[2] = __builtin_readcyclecounter(); }
#endif
    // online softmax per query block; the query is this lane's column, split over the two lane halves
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = s[qb][0][0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[qb][0][i]);
      if (two) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[qb][1][i]);
      }
      mx = half_max(mx);
      // (Deferring the rescale until the max has grown by 2^6 — p up to 64 instead of 1 — measured 525 -> 505 us at N = 4097 and nothing at
      // N = 513, and moved the model's logits from 6.7e-3 to 1.3e-2 off the bf16-emulating oracle, whose P is rounded relative to the
      // exact running max: not kept.)
      if (__any(mx > m_run[qb])) {   // some row's running max moved: rescale (otherwise alpha == 1 exactly, skip the pass)
        const float m_new = fmaxf(m_run[qb], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[qb][0][i] *= alpha; oacc[qb][1][i] *= alpha; }
      }
      // p = exp2(s c - m c), two scores per v_pk_fma_f32 / v_pk_add_f32 (the loop is bound by VALU issue, not by MFMA)
      const f32x2 c2 = {c, c}, nmc2 = {-m_run[qb] * c, -m_run[qb] * c};
      f32x2 psum2 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 t = f32x2{s[qb][0][2 * i], s[qb][0][2 * i + 1]} * c2 + nmc2;
        const f32x2 pv = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        s[qb][0][2 * i] = pv.x; s[qb][0][2 * i + 1] = pv.y;
        psum2 += pv;
      }
      if (two) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x2 t = f32x2{s[qb][1][2 * i], s[qb][1][2 * i + 1]} * c2 + nmc2;
          const f32x2 pv = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
          s[qb][1][2 * i] = pv.x; s[qb][1][2 * i + 1] = pv.y;
          psum2 += pv;
        }
      }
      l_run[qb] += psum2.x + psum2.y;
      if constexpr (DROP) {   // mask P after the row sum, before P.V
        const uint64_t bh_base = ((uint64_t)b * H + head) * (uint64_t)N * (uint64_t)N;
        const int query = q0 + qb * 32 + (lane & 31);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            s[qb][kb][i] = drop_keep(drop, bh_base, query, t * TILE_ROWS + kb * 32 + acc_row(i, h), N) ? s[qb][kb][i] * drop.inv : 0.f;
      }
    }
#ifdef XVIT_DEBUG_ATTN_TIMES
    if (t == 3) { asm volatile("s_nop 0" ::"v"(s[0][0][0]), "v"(s[0][1][15])); tk[3] = __builtin_readcyclecounter(); }
#endif
    // O^T[d][query] += V^T P^T: every V fragment is read once and used by all QB query blocks
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 1 && !two) break;
#pragma unroll
      for (int ss = 0; ss < FWD_SS; ++ss) {
        bf16x8 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) pf[qb] = acc_frag(s[qb][kb], ss);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = rd.tr_frag(vimg, db, kb, ss);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], oacc[qb][db], 0, 0, 0);
        }
      }
    }
#ifdef XVIT_DEBUG_ATTN_TIMES
    if (t == 3) { asm volatile("s_nop 0" ::"v"(oacc[0][0][0]), "v"(oacc[0][1][15])); tk[4] = __builtin_readcyclecounter(); }
#endif
  }
#ifdef XVIT_DEBUG_ATTN_TIMES
  if (lane == 0 && wave_active) {   // borrow the lse buffer: 4 floats per wave = cycles of (wait+barrier, S, softmax, PV) in iteration 3
    float* dbg = lse + ((((int64_t)b * H + head) * gridDim.x + bc.x) * 4 + wave) * 8;
    for (int i = 0; i < 4; ++i) dbg[i] = (float)(tk[i + 1] - tk[i]);
    const uint64_t wc_end = wall_clock64();
    dbg[4] = (float)(wc_entry & 0xFFFFFF); dbg[5] = (float)(wc_loop - wc_entry); dbg[6] = (float)(wc_end - wc_loop); dbg[7] = 0.f;   // 100 MHz ticks
  }
  return;
#endif
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = half_sum(l_run[qb]);
    const int qrow = q0 + qb * 32 + (lane & 31);
    const bool valid = qrow < N;
    if (valid && h == 0) lse[((int64_t)b * H + head) * N + qrow] = m_run[qb] * scale + __logf(l_tot);
    store_lane_rows(oacc[qb], 1.0f / l_tot, o + (int64_t)b * osb + head * DH, osn, qrow, valid, lane);
  }
}

// ------------------------------------------------------------------------------------------
// backward, dQ: one wave = 32 queries, streams K and V tiles
// ------------------------------------------------------------------------------------------
// (133 VGPRs: three blocks per CU; squeezing it to 128 for a fourth spills and measured 6 % slower)
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                             int64_t sb, int64_t sn, const bf16* __restrict__ o, const bf16* __restrict__ d_o, int64_t osb,
                                                             int64_t osn, const float* __restrict__ lse, float* __restrict__ nlse_ws,
                                                             float* __restrict__ delta, bf16* __restrict__ dq, int H, int N, float scale, const DropArgs drop) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int q0 = bc.x * 128 + wave * 32;
  const int64_t off = (int64_t)b * sb + head * DH;
  const int ntiles = (N + TILE_ROWS - 1) / TILE_ROWS;

  TileLoader lk, lv;
  lk.init(k + off, sn, N, wave, lane);
  lv.init(v + off, sn, N, wave, lane);
#pragma unroll
  for (int st = 0; st < BWD_NST - 1; ++st)
    if (st < ntiles) {
      lk.issue(smem + st * 2 * IMG_BYTES, wave, st);
      lv.issue(smem + st * 2 * IMG_BYTES + IMG_BYTES, wave, st);
    }

  bf16x8 qf[4], dof[4];
  load_lane_operand(qf, q + off, sn, q0, N, lane);
  load_lane_operand(dof, d_o + (int64_t)b * osb + head * DH, osn, q0, N, lane);
  settle(qf);
  settle(dof);
  ImgReader rd;
  rd.init(lane);

  const int qrow = q0 + (lane & 31);
  const bool valid = qrow < N;
  const bool wave_active = q0 < N;   // wave-uniform
  const int64_t stat = ((int64_t)b * H + head) * N + qrow;
  const float c = scale * LOG2E;
  // Row statistics of the backward, computed HERE from the query rows this wave already holds (dO in registers, O loaded the
  // same way) and left in the workspace for the dK/dV kernel, which is launched after this one: delta = rowsum(dO . O) and
  // nlse = -lse log2(e) (P = exp2(s c + nlse): no per-tile transform).  A separate pass over O and dO (198 MB at configs[1])
  // did this before.  Invalid query rows: nlse = -big -> P = 0, nothing is accumulated for them.
  float nlse, dlt;
  {
    bf16x8 of[4];
    load_lane_operand(of, o + (int64_t)b * osb + head * DH, osn, q0, N, lane);
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) part = fmaf(bf2f(of[ks][j]), bf2f(dof[ks][j]), part);
    dlt = valid ? half_sum(part) : 0.f;          // the two lane halves hold the two halves of d
    nlse = valid ? -lse[stat] * LOG2E : -1e30f;
    if (valid && (lane >> 5) == 0) { delta[stat] = dlt; nlse_ws[stat] = nlse; }
  }
  asm volatile("" : "+v"(nlse), "+v"(dlt));   // settle these loads (and qf/dof below) before the tile loop: see settle()

  f32x16 dqacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }

  int stage = 0;
  for (int t = 0; t < ntiles; ++t) {
    ring_wait<BWD_NST, 4>(t, ntiles);
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t are in LDS; every wave is done with tile t-1
    if (t + BWD_NST - 1 < ntiles) {
      int ns = stage + BWD_NST - 1;
      if (ns >= BWD_NST) ns -= BWD_NST;
      lk.issue(smem + ns * 2 * IMG_BYTES, wave, t + BWD_NST - 1);
      lv.issue(smem + ns * 2 * IMG_BYTES + IMG_BYTES, wave, t + BWD_NST - 1);
    }
    const XVIT_LDS char* kimg = smem + stage * 2 * IMG_BYTES;
    const XVIT_LDS char* vimg = kimg + IMG_BYTES;
    stage = stage + 1 == BWD_NST ? 0 : stage + 1;
    if (!wave_active) continue;
    const int nkb = (N - t * TILE_ROWS) > 32 ? 2 : 1;   // a tail tile with <= 32 keys needs one key block only
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb >= nkb) break;
      f32x16 s, dp;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(kimg, kb, ks), qf[ks], ks == 0 ? ZERO16 : s, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(vimg, kb, ks), dof[ks], ks == 0 ? ZERO16 : dp, 0, 0, 0);
      // keys past N have K = V = 0 (zero-filled by the DMA), so they add nothing to dQ
#pragma unroll
      for (int i = 0; i < 16; ++i) {   // (hand-packed v_pk_fma / v_pk_mul here measured 4 % slower at N = 4097: left to the compiler)
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[i], c, nlse));
        float dpe = dp[i];
        if constexpr (DROP)   // dP = mask / (1-p) * (dO V^T)
          dpe = drop_keep(drop, ((uint64_t)b * H + head) * (uint64_t)N * (uint64_t)N, qrow, t * TILE_ROWS + kb * 32 + acc_row(i, lane >> 5), N) ? dpe * drop.inv : 0.f;
        s[i] = pv * (dpe - dlt);  // dS^T
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 dsf = acc_frag(s, ss);
#pragma unroll
        for (int db = 0; db < 2; ++db) dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.tr_frag(kimg, db, kb, ss), dsf, dqacc[db], 0, 0, 0);
      }
    }
  }
  store_lane_rows(dqacc, scale, dq + off, sn, qrow, valid, lane);
}

// ------------------------------------------------------------------------------------------
// backward, dK/dV: one wave = 32 keys, streams Q and dO tiles (+ lse, delta)
// ------------------------------------------------------------------------------------------
constexpr int DKV_STAGE = 2 * IMG_BYTES + 512;  // Q image | dO image | nlse[64] | delta[64]  (all four arrive by LDS-DMA)

// (198 VGPRs: two blocks per CU; bounding it to 168 for a third spills and measured 7 % slower; keeping the query-block loop
// rolled gives 162 VGPRs without spills and three blocks per CU: 552 vs 563 us at B = 126, N = 513, but 50.4 vs 47.0 us at B = 8
// and 1466 vs 1439 us at N = 4097 — not kept)
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                              int64_t sb, int64_t sn, const bf16* __restrict__ d_o, int64_t osb, int64_t osn,
                                                              const float* __restrict__ nlse_ws, const float* __restrict__ delta,
                                                              bf16* __restrict__ dk, bf16* __restrict__ dv, int H, int N, float scale, const DropArgs drop) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int k0 = bc.x * 128 + wave * 32;
  const int64_t off = (int64_t)b * sb + head * DH;
  const int64_t ooff = (int64_t)b * osb + head * DH;
  const int64_t stat0 = ((int64_t)b * H + head) * N;
  const int ntiles = (N + TILE_ROWS - 1) / TILE_ROWS;

#ifdef XVIT_DEBUG_ATTN_TIMES
  const uint64_t wc_entry = wall_clock64();
  uint64_t tk[4] = {0, 0, 0, 0};
#endif
  TileLoader lq, ldo;
  lq.init(q + off, sn, N, wave, lane);
  ldo.init(d_o + ooff, osn, N, wave, lane);
  // per-query statistics of a tile (64 floats each) also arrive by LDS-DMA (4 B per lane: even waves move nlse, odd waves
  // delta), so the loop contains NO ordinary global load whose compiler-inserted vmcnt(0) would drain the tile DMAs.
  // Rows past N read as 0 (buffer bounds): nlse = 0 gives P = 1 there, harmless because dO = 0 and delta = 0.
  const __amdgpu_buffer_rsrc_t rstat = make_rsrc(((wave & 1) == 0 ? nlse_ws : delta) + stat0, clamp_bytes((int64_t)N * 4));
  auto stage_stats = [&](XVIT_LDS char* st, int tile) {   // waves 2, 3 repeat waves 0, 1 (same bytes): every wave issues 5 DMAs per tile
    glds4(rstat, st + 2 * IMG_BYTES + (wave & 1) * 256, (uint32_t)(lane * 4), (uint32_t)(tile * TILE_ROWS * 4));
  };
#pragma unroll
  for (int st = 0; st < BWD_NST - 1; ++st)
    if (st < ntiles) {
      lq.issue(smem + st * DKV_STAGE, wave, st);
      ldo.issue(smem + st * DKV_STAGE + IMG_BYTES, wave, st);
      stage_stats(smem + st * DKV_STAGE, st);
    }

  bf16x8 kf[4], vf[4];
  load_lane_operand(kf, k + off, sn, k0, N, lane);
  load_lane_operand(vf, v + off, sn, k0, N, lane);
  settle(kf);
  settle(vf);
  ImgReader rd;
  rd.init(lane);
  const int h = lane >> 5;
  const float c = scale * LOG2E;
  const bool wave_active = k0 < N;   // wave-uniform

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

#ifdef XVIT_DEBUG_ATTN_TIMES
  const uint64_t wc_loop = wall_clock64();
#endif
  int stage = 0;
  for (int t = 0; t < ntiles; ++t) {
#ifdef XVIT_DEBUG_ATTN_TIMES
    if (t == 3) tk[0] = __builtin_readcyclecounter();
    if (t == 4) tk[2] = __builtin_readcyclecounter();
#endif
    ring_wait<BWD_NST, 5>(t, ntiles);
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t are in LDS; every wave is done with tile t-1
#ifdef XVIT_DEBUG_ATTN_TIMES
    if (t == 3) tk[1] = __builtin_readcyclecounter();
#endif
    if (t + BWD_NST - 1 < ntiles) {
      int ns = stage + BWD_NST - 1;
      if (ns >= BWD_NST) ns -= BWD_NST;
      XVIT_LDS char* nxt = smem + ns * DKV_STAGE;
      lq.issue(nxt, wave, t + BWD_NST - 1);
      ldo.issue(nxt + IMG_BYTES, wave, t + BWD_NST - 1);
      stage_stats(nxt, t + BWD_NST - 1);
    }
    const XVIT_LDS char* qimg = smem + stage * DKV_STAGE;
    stage = stage + 1 == BWD_NST ? 0 : stage + 1;
    const XVIT_LDS char* doimg = qimg + IMG_BYTES;
    const XVIT_LDS float* st_lse = (const XVIT_LDS float*)(qimg + 2 * IMG_BYTES);
    const XVIT_LDS float* st_dlt = st_lse + 64;
    if (!wave_active) continue;
    const int nqb = (N - t * TILE_ROWS) > 32 ? 2 : 1;   // a tail tile with <= 32 queries needs one query block only
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (qb >= nqb) break;
      f32x16 s, dp;
      // S[query][key] = Q K^T ; dP[query][key] = dO V^T   (key on the lane)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(qimg, qb, ks), kf[ks], ks == 0 ? ZERO16 : s, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(doimg, qb, ks), vf[ks], ks == 0 ? ZERO16 : dp, 0, 0, 0);
      f32x16 pr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 nl = *(const XVIT_LDS f32x4*)(st_lse + qb * 32 + 8 * g + 4 * h);
        const f32x4 dl = *(const XVIT_LDS f32x4*)(st_dlt + qb * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[g * 4 + e], c, nl[e]));
          if constexpr (DROP) {   // dV uses P_drop = mask/(1-p) P; dP = mask/(1-p) (dO V^T)
            const bool keep = drop_keep(drop, (uint64_t)stat0 * (uint64_t)N, t * TILE_ROWS + qb * 32 + 8 * g + 4 * h + e, k0 + (lane & 31), N);
            pr[g * 4 + e] = keep ? pv * drop.inv : 0.f;
            s[g * 4 + e] = pv * ((keep ? dp[g * 4 + e] * drop.inv : 0.f) - dl[e]);  // dS
          } else {
            pr[g * 4 + e] = pv;
            s[g * 4 + e] = pv * (dp[g * 4 + e] - dl[e]);  // dS
          }
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = acc_frag(pr, ss), dsf = acc_frag(s, ss);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.tr_frag(doimg, db, qb, ss), pf, dvacc[db], 0, 0, 0);
          dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.tr_frag(qimg, db, qb, ss), dsf, dkacc[db], 0, 0, 0);
        }
      }
    }
  }
#ifdef XVIT_DEBUG_ATTN_TIMES
  if (lane == 0 && wave_active) {   // borrow dk (bf16 [*, sn] rows): 8 floats per wave written at the start of this block's first dK row... use delta workspace instead
    float* dbg = const_cast<float*>(delta) + ((((int64_t)b * H + head) * gridDim.x + bc.x) * 4 + wave) * 8;
    asm volatile("s_nop 0" ::"v"(dkacc[0][0]), "v"(dkacc[1][15]), "v"(dvacc[0][0]), "v"(dvacc[1][15]));
    const uint64_t wc_end = wall_clock64();
    dbg[0] = (float)(tk[1] - tk[0]); dbg[1] = (float)(tk[2] - tk[1]); dbg[2] = 0.f; dbg[3] = 0.f;
    dbg[4] = (float)(wc_entry & 0xFFFFFF); dbg[5] = (float)(wc_loop - wc_entry); dbg[6] = (float)(wc_end - wc_loop); dbg[7] = 0.f;
  }
  asm volatile("" ::"v"(dkacc[0][0]), "v"(dkacc[1][15]), "v"(dvacc[0][0]), "v"(dvacc[1][15]));   // keep the accumulations alive
  return;
#endif
  const int krow = k0 + (lane & 31);
  const bool valid = krow < N;
  store_lane_rows(dkacc, scale, dk + off, sn, krow, valid, lane);
  store_lane_rows(dvacc, 1.0f, dv + off, sn, krow, valid, lane);
}

}  // namespace xvit

using namespace xvit;

static int attn_check(const char* who, int B, int H, int N, int dh, int64_t sb, int64_t sn, int64_t osb, int64_t osn) {
  XVIT_REQUIRE(dh == DH, "%s: head dim %d unsupported (only 64)", who, dh);
  XVIT_REQUIRE(B > 0 && H > 0 && N > 0 && B <= 65535 && H <= 65535, "%s: bad B/H/N (%d,%d,%d)", who, B, H, N);
  XVIT_REQUIRE(sn % 8 == 0 && sb % 8 == 0 && osn % 8 == 0 && osb % 8 == 0, "%s: strides must be multiples of 8 elements", who);
  XVIT_REQUIRE((int64_t)N * sn * 2 < (1ll << 31) && (int64_t)N * osn * 2 < (1ll << 31), "%s: one batch slice exceeds 2 GiB", who);
  return XVIT_OK;
}

static DropArgs drop_args(float p, uint64_t seed) { return DropArgs{(uint32_t)(p * 16777216.0f), 1.0f / (1.0f - p), seed}; }

extern "C" int xvit_attn_fwd(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, void* o, int64_t osb, int64_t osn, float* lse,
                             int B, int H, int N, int dh, float scale, float dropout_p, uint64_t dropout_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(q && k && v && o && lse, "xvit_attn_fwd: null pointer");
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "xvit_attn_fwd: dropout_p must be in [0, 1)");
  if (int e = attn_check("xvit_attn_fwd", B, H, N, dh, sb, sn, osb, osn)) return e;
  // QB = 1 (32 queries per wave).  QB = 2 was measured: identical throughput at N = 512..4097 (the loop is bound by
  // softmax VALU issue, 12.4 VALU per MFMA at d_h = 64 — not by LDS reads or per-wave ILP) and worse at small batch.
  const dim3 grid((N + 127) / 128, H, B), block(256);
  const DropArgs da = drop_args(dropout_p, dropout_seed);
  if (dropout_p > 0.f)
    hipLaunchKernelGGL((attn_fwd_kernel<1, true>), grid, block, FWD_NST * 2 * IMG_BYTES, (hipStream_t)stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                       (bf16*)o, osb, osn, lse, H, N, scale, da);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<1, false>), grid, block, FWD_NST * 2 * IMG_BYTES, (hipStream_t)stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                       (bf16*)o, osb, osn, lse, H, N, scale, da);
  return check_launch("xvit_attn_fwd");
}

extern "C" int xvit_attn_bwd(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, const void* o, const void* d_o, int64_t osb,
                             int64_t osn, const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int N, int dh, float scale,
                             float dropout_p, uint64_t dropout_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "xvit_attn_bwd: null pointer");
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "xvit_attn_bwd: dropout_p must be in [0, 1)");
  if (int e = attn_check("xvit_attn_bwd", B, H, N, dh, sb, sn, osb, osn)) return e;
  hipStream_t s = (hipStream_t)stream;
  static const bool lds_opt_in = [] {   // rings deeper than 3 stages need more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * DKV_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * DKV_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * 2 * IMG_BYTES);
    return true;
  }();
  (void)lds_opt_in;
  const int total = B * H * N;
  float* nlse = delta + total;   // workspace = [2][B,H,N]: delta | -lse*log2e
  const dim3 grid((N + 127) / 128, H, B), block(256);
  const DropArgs da = drop_args(dropout_p, dropout_seed);
  if (dropout_p > 0.f) {
    // the dQ kernel first: it also leaves delta and -lse log2(e) of every query row in the workspace for the dK/dV kernel
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, block, BWD_NST * 2 * IMG_BYTES, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn, (const bf16*)o,
                       (const bf16*)d_o, osb, osn, lse, nlse, delta, (bf16*)dq, H, N, scale, da);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, block, BWD_NST * DKV_STAGE, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn, (const bf16*)d_o, osb,
                       osn, nlse, delta, (bf16*)dk, (bf16*)dv, H, N, scale, da);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, block, BWD_NST * 2 * IMG_BYTES, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn, (const bf16*)o,
                       (const bf16*)d_o, osb, osn, lse, nlse, delta, (bf16*)dq, H, N, scale, da);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, block, BWD_NST * DKV_STAGE, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn, (const bf16*)d_o, osb,
                       osn, nlse, delta, (bf16*)dk, (bf16*)dv, H, N, scale, da);
  }
  return check_launch("xvit_attn_bwd");
}
