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
#include <atomic>

#include "xvit_common.h"

#ifndef XVIT_PEEL_DEBUG
#define XVIT_PEEL_DEBUG 0     // timing-only diagnostic builds (WRONG results): 1 = no post-loop token-0 block, 2 = no token-0 initial state, 4 = no tile rotation
#endif

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
// PEEL: workgroup x of a (b, head) walks the streamed tiles in ROTATED order, ending with the two tiles (2x, 2x+1) that hold the
// tokens with its own row indices.  Softmax statistics and the gradient sums do not depend on the order, and after the loop
// those two tiles are still in the ring, so each wave runs its one token-0 block (see "CLS peel" below) THERE: all four waves
// at once, behind the last barrier.  (Inside the loop the extra block of one wave held the other three at the next
// barrier: +15..20 % per kernel measured instead of the +6 % of work.)
struct TileOrder {
  int own, pair, ntiles;   // own = first tile of the workgroup's pair, pair = 1 or 2 tiles (0: natural order)
  __device__ __forceinline__ void init(bool peel, int x, int nt) {
    ntiles = nt;
    own = 2 * x;
    pair = (peel && !(XVIT_PEEL_DEBUG & 4)) ? min(2, nt - own) : 0;
  }
  // every other tile in natural order (the workgroups of a (b, head) keep streaming the same tiles at about the same time: L2), then the own pair
  __device__ __forceinline__ int tile(int it) const {
    if (pair == 0) return it;
    const int head = ntiles - pair;
    return it >= head ? own + (it - head) : (it < own ? it : it + pair);
  }
  // ring stage that still holds token-0 block `g`'s tile after the loop (g = 4 x + wave; its tile is 2x or 2x + 1)
  __device__ __forceinline__ int stage_of(int g, int nst) const {
    const int it = pair == 0 ? (g >> 1) : ntiles - pair + ((g >> 1) - own);
    return it % nst;
  }
};
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

// ------------------------------------------------------------------------------------------
// CLS peel (N = 64 m + 1: the reference's cls token + a whole number of patch-token tiles, model_cross.py:195-196).
// Token 0 would cost a whole extra query block per (b, head) and a whole extra key tile per block (5 x 9 block-tiles
// instead of 4 x 8 at N = 513: +47 % forward, +23 % backward measured) for one row and one column of work.  With
// PEEL the tile grid covers the patch tokens 1 .. N-1 only and token 0 rides along as rank-one side work:
//   * CLS as a KEY is one column of S: per query (lane) one 64-long dot gives its score, so it enters the forward as the
//     INITIAL online-softmax state (m = s0, l = 1, O = v0) and the backward's dQ kernel as the initial accumulator
//     dQ = dS[q, 0] k0; its own gradients dK0 = sum_q dS[q, 0] Q[q], dV0 = sum_q P[q, 0] dO[q] are column sums over queries:
//     the dK/dV kernel (Q / dO tiles in LDS) takes them as ONE extra MFMA block per wave whose key operand is k0 / v0
//     repeated in all 32 columns.
//   * CLS as a QUERY is one row of S: the dK/dV kernel (key on the lane) adds its rank-one terms dK[k] += dS[0, k] q0,
//     dV[k] += P[0, k] dO0 as the initial accumulators; its own output O[0] (forward) and gradient dQ0 (backward) are row sums
//     over keys: the forward / dQ kernels (K / V tiles in LDS) take them as ONE extra MFMA block per wave whose query
//     operand is q0 / dO0 repeated in all 32 columns, each wave over the 32 keys whose index equals its own query block.
// Every wave leaves its partial (fixed slot), a small merge kernel adds them in slot order and finishes row 0: no atomics,
// bit-reproducible.  The extra block is 1/16 of a wave's work at N = 513 and 1/128 at N = 4097.
// ------------------------------------------------------------------------------------------
constexpr int CLS_SLOT = 80;   // floats per forward partial: o[64] (unnormalised) | m (raw max) | l | pad (16-B aligned rows)

// row `rowp` (64 bf16) as an MFMA operand whose 32 columns (B) / rows (A) all equal that row
__device__ __forceinline__ void load_row_bcast(bf16x8 (&f)[4], const bf16* rowp, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *(const bf16x8*)(rowp + ks * 16 + h * 8);
}
// the same row as fp32 in accumulator order: x[db][i] = row[32 db + acc_row(i, h)]
__device__ __forceinline__ void load_row_acc(f32x16 (&x)[2], const bf16* rowp, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 t = *(const bf16x4*)(rowp + 32 * db + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[db][4 * g + e] = bf2f(t[e]);
    }
}
// Token-0 rows for the post-loop block, prefetched into LDS: one LDS-DMA per row, issued by wave 0 BEFORE its first tile DMAs (so its
// first ring wait covers them, and the loop's first barrier publishes them); the row is repeated over the instruction's 1 KiB
// (lane L fetches 16-byte chunk L & 7).  A global load after the loop would expose ~1-2 us of latency per workgroup (measured:
// 12 us of a 170 us forward, 33 us of a 540 us backward at B = 126, N = 513).
constexpr int ROW0_BYTES = 1024;
__device__ __forceinline__ void stage_row0(XVIT_LDS char* dst, const bf16* rowp, int lane) {
  glds16(make_rsrc(rowp, 128), dst, (uint32_t)((lane & 7) * 16), 0);
}
__device__ __forceinline__ void lds_row_bcast(bf16x8 (&f)[4], const XVIT_LDS char* row, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *(const XVIT_LDS bf16x8*)(row + ks * 32 + h * 16);
}
// dot over d of two operand fragment sets: each lane holds 32 of the 64 d, the lane halves complete each other
__device__ __forceinline__ float frag_dot(const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
  float p = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) p = fmaf(bf2f(a[ks][j]), bf2f(b[ks][j]), p);
  return half_sum(p);
}
// column 0 of a [64 d][32 identical columns] accumulator pair -> 64 consecutive floats
__device__ __forceinline__ void store_col0(const f32x16 (&acc)[2], float* dst, int lane) {
  if ((lane & 31) != 0) return;
  const int h = lane >> 5;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) *(f32x4*)(dst + 32 * db + 8 * g + 4 * h) = f32x4{acc[db][4 * g], acc[db][4 * g + 1], acc[db][4 * g + 2], acc[db][4 * g + 3]};
}

// Probability dropout (reference model.py:169: attention_probs = attn_dropout(softmax(scores))): element (b, h, query, key) is
// kept iff hash32(seed, ((b H + h) N + query) N + key) >= p 2^24 — the mask xvit_dropout applies to a contiguous [B, H, N, N]
// tensor with the same seed — and scaled by 1/(1-p).  The row sums (softmax normaliser) use the probabilities BEFORE the
// mask; the backward kernels regenerate the mask (nothing is stored) and, with O = P_drop V, delta = rowsum(dO O) is unchanged.
struct DropArgs { uint32_t thr; float inv; uint64_t seed; const uint64_t* epoch; };   // epoch: xvit_common.h drop_seed_at (captured steps), or nullptr
__device__ __forceinline__ DropArgs drop_at_run_time(DropArgs d) { d.seed = drop_seed_at(d.seed, d.epoch); return d; }
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

template <int QB, bool DROP, bool PEEL>
__global__ __launch_bounds__(256, QB == 1 ? 4 : 2) void attn_fwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                          int64_t sb, int64_t sn, bf16* __restrict__ o, int64_t osb, int64_t osn,
                                                          float* __restrict__ lse, int H, int N, float scale, const DropArgs drop_in, float* __restrict__ cls_ws) {
  const DropArgs drop = DROP ? drop_at_run_time(drop_in) : drop_in;   // captured steps: seed + device-side epoch
  static_assert(!PEEL || (QB == 1 && !DROP), "the CLS peel is built for 32 queries per wave, no probability dropout");
#ifdef XVIT_DEBUG_ATTN_TIMES
  const uint64_t wc_entry = wall_clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;  // [FWD_NST stages][K image | V image]
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int q0 = bc.x * (128 * QB) + wave * (32 * QB);
  const int NK = PEEL ? N - 1 : N;                                   // tokens of the tile grid (PEEL: the patch tokens, rows 1 .. N-1)
  const int64_t off0 = (int64_t)b * sb + head * DH;                  // token 0 of this (b, head)
  const int64_t off = off0 + (PEEL ? sn : 0);                        // first token of the tile grid
  const int ntiles = (NK + TILE_ROWS - 1) / TILE_ROWS;

  // K/V tiles stream through a FWD_NST-deep LDS ring.  Loads stay in flight ACROSS iterations: each wave
  // issues exactly 4 LDS-DMA instructions per tile (2 K + 2 V), so "tile t has landed, tile t+1 may still be
  // in flight" is a counted s_waitcnt vmcnt(4); barriers are raw s_barrier (a __syncthreads() would drain
  // vmcnt to 0 and serialise every iteration behind a full HBM/L2 round trip).
  TileLoader lk, lv;
  lk.init(k + off, sn, NK, wave, lane);
  lv.init(v + off, sn, NK, wave, lane);
  TileOrder ord;
  ord.init(PEEL, bc.x, ntiles);
  XVIT_LDS char* row0 = smem + FWD_NST * 2 * IMG_BYTES;   // PEEL: [q0] (ROW0_BYTES)
  if constexpr (PEEL) {
    if (wave == 0) stage_row0(row0, q + off0, lane);
  }
#pragma unroll
  for (int st = 0; st < FWD_NST - 1; ++st)
    if (st < ntiles) {
      lk.issue(smem + st * 2 * IMG_BYTES, wave, ord.tile(st));
      lv.issue(smem + st * 2 * IMG_BYTES + IMG_BYTES, wave, ord.tile(st));
    }
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    load_lane_operand(qf[qb], q + off, sn, q0 + qb * 32, NK, lane);   // one round trip together with the first ring tiles
    settle(qf[qb]);                                                  // (drains the prologue's DMAs too: fine, they are needed first)
  }
  ImgReader rd;
  rd.init(lane);

  const int h = lane >> 5;
  const float c = scale * LOG2E;
  const bool wave_active = q0 < NK;   // wave-uniform
  float m_run[QB], l_run[QB];
  f32x16 oacc[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[qb][0][i] = 0.f; oacc[qb][1][i] = 0.f; }
  }
  // PEEL: the CLS key is the initial online-softmax state of every query: m = s0 = q . k0 (raw, like the MFMA scores), p0 = 1, O = v0.
  // (l_run is a per-lane-half partial sum: the 1 goes to half 0.)  This wave's CLS-QUERY block: the 32 keys with its own queries' indices.
  const int cls_g = bc.x * 4 + wave, cls_kb = cls_g & 1;
  if constexpr (PEEL && !(XVIT_PEEL_DEBUG & 2)) {
    bf16x8 k0f[4];
    load_row_bcast(k0f, k + off0, lane);
    m_run[0] = frag_dot(qf[0], k0f);
    l_run[0] = h == 0 ? 1.f : 0.f;
    load_row_acc(oacc[0], v + off0, lane);
    asm volatile("" : "+v"(m_run[0]));   // settle these loads before the tile loop: see settle()
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
      lk.issue(smem + ns * 2 * IMG_BYTES, wave, ord.tile(t + FWD_NST - 1));
      lv.issue(smem + ns * 2 * IMG_BYTES + IMG_BYTES, wave, ord.tile(t + FWD_NST - 1));
    }
    const XVIT_LDS char* kimg = smem + stage * 2 * IMG_BYTES;
    const XVIT_LDS char* vimg = kimg + IMG_BYTES;
    stage = stage + 1 == FWD_NST ? 0 : stage + 1;

    if (!wave_active) continue;   // this wave's queries are all past N: keep moving tiles and barriers, skip the math
    // keys of this tile past N: with <= 32 valid keys the second 32-key block is skipped entirely
    const int valid = PEEL ? TILE_ROWS : NK - t * TILE_ROWS;   // PEEL: 64 | NK, every tile is full
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
    const bool valid = qrow < NK;
    if (valid && h == 0) lse[((int64_t)b * H + head) * N + (PEEL ? 1 : 0) + qrow] = m_run[qb] * scale + __logf(l_tot);
    store_lane_rows(oacc[qb], 1.0f / l_tot, o + (int64_t)b * osb + head * DH + (PEEL ? osn : 0), osn, qrow, valid, lane);
  }
  if constexpr (PEEL && !(XVIT_PEEL_DEBUG & 1)) {
    if (wave_active) {   // once per wave, after the loop: the CLS query against key block cls_kb of its own tile (still in the ring), q0 in all 32 columns
      const XVIT_LDS char* kimg = smem + ord.stage_of(cls_g, FWD_NST) * 2 * IMG_BYTES;
      const XVIT_LDS char* vimg = kimg + IMG_BYTES;
      bf16x8 q0f[4];
      lds_row_bcast(q0f, row0, lane);
      f32x16 sc;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(kimg, cls_kb, ks), q0f[ks], ks == 0 ? ZERO16 : sc, 0, 0, 0);
      float mx = sc[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sc[i]);
      mx = half_max(mx);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sc[i] = __builtin_amdgcn_exp2f((sc[i] - mx) * c); ps += sc[i]; }
      ps = half_sum(ps);
      f32x16 oc[2];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = acc_frag(sc, ss);
#pragma unroll
        for (int db = 0; db < 2; ++db) oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.tr_frag(vimg, db, cls_kb, ss), pf, ss == 0 ? ZERO16 : oc[db], 0, 0, 0);
      }
      float* slot = cls_ws + ((((int64_t)b * H + head) * (NK / 32)) + cls_g) * CLS_SLOT;
      store_col0(oc, slot, lane);
      if (lane == 0) { slot[64] = mx; slot[65] = ps; }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, dQ: one wave = 32 queries, streams K and V tiles
// ------------------------------------------------------------------------------------------
// (133 VGPRs: three blocks per CU; squeezing it to 128 for a fourth spills and measured 6 % slower)
// one 32-key block of the dQ recomputation: S^T = K Q^T, dP^T = V dO^T -> dS^T = P^T (dP^T - delta) -> dQ^T += K^T dS^T.  qf / dof: the
// queries on the lanes (or, for the CLS row, q0 / dO0 repeated in every column); nlse / dlt: their row statistics
__device__ __forceinline__ void dq_block(const ImgReader& rd, const XVIT_LDS char* kimg, const XVIT_LDS char* vimg, int kb, const bf16x8 (&qf)[4],
                                         const bf16x8 (&dof)[4], float c, float nlse, float dlt, f32x16 (&dqacc)[2]) {
  f32x16 s, dp;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(kimg, kb, ks), qf[ks], ks == 0 ? ZERO16 : s, 0, 0, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(vimg, kb, ks), dof[ks], ks == 0 ? ZERO16 : dp, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(fmaf(s[i], c, nlse)) * (dp[i] - dlt);
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) {
    const bf16x8 dsf = acc_frag(s, ss);
#pragma unroll
    for (int db = 0; db < 2; ++db) dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.tr_frag(kimg, db, kb, ss), dsf, dqacc[db], 0, 0, 0);
  }
}

// PEEL workspace (after delta | nlse): per (b, head) and wave slot g, 64 floats each of dQ0, dK0, dV0 partials
#ifndef XVIT_DQ_PEEL_WAVES
#define XVIT_DQ_PEEL_WAVES 4
#endif
#ifndef XVIT_DKV_PEEL_WAVES
#define XVIT_DKV_PEEL_WAVES 3
#endif
template <bool DROP, bool PEEL>
__global__ __launch_bounds__(256, PEEL ? XVIT_DQ_PEEL_WAVES : 2) void attn_bwd_dq_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                             int64_t sb, int64_t sn, const bf16* __restrict__ o, const bf16* __restrict__ d_o, int64_t osb,
                                                             int64_t osn, const float* __restrict__ lse, float* __restrict__ nlse_ws,
                                                             float* __restrict__ delta, bf16* __restrict__ dq, int H, int N, float scale, const DropArgs drop_in,
                                                             float* __restrict__ pdq) {
  const DropArgs drop = DROP ? drop_at_run_time(drop_in) : drop_in;   // captured steps: seed + device-side epoch
  static_assert(!PEEL || !DROP, "the CLS peel is built without probability dropout");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int q0 = bc.x * 128 + wave * 32;
  const int NK = PEEL ? N - 1 : N;
  const int64_t off0 = (int64_t)b * sb + head * DH, off = off0 + (PEEL ? sn : 0);
  const int64_t ooff0 = (int64_t)b * osb + head * DH, ooff = ooff0 + (PEEL ? osn : 0);
  const int ntiles = (NK + TILE_ROWS - 1) / TILE_ROWS;

  TileLoader lk, lv;
  lk.init(k + off, sn, NK, wave, lane);
  lv.init(v + off, sn, NK, wave, lane);
  TileOrder ord;
  ord.init(PEEL, bc.x, ntiles);
  XVIT_LDS char* row0 = smem + BWD_NST * 2 * IMG_BYTES;   // PEEL: [q0 | dO0] (2 ROW0_BYTES)
  if constexpr (PEEL) {
    if (wave == 0) {
      stage_row0(row0, q + off0, lane);
      stage_row0(row0 + ROW0_BYTES, d_o + ooff0, lane);
    }
  }
#pragma unroll
  for (int st = 0; st < BWD_NST - 1; ++st)
    if (st < ntiles) {
      lk.issue(smem + st * 2 * IMG_BYTES, wave, ord.tile(st));
      lv.issue(smem + st * 2 * IMG_BYTES + IMG_BYTES, wave, ord.tile(st));
    }

  bf16x8 qf[4], dof[4];
  load_lane_operand(qf, q + off, sn, q0, NK, lane);
  load_lane_operand(dof, d_o + ooff, osn, q0, NK, lane);
  settle(qf);
  settle(dof);
  ImgReader rd;
  rd.init(lane);

  const int qrow = q0 + (lane & 31);
  const bool valid = qrow < NK;
  const bool wave_active = q0 < NK;   // wave-uniform
  const int64_t stat = ((int64_t)b * H + head) * N + (PEEL ? 1 : 0) + qrow;
  const float c = scale * LOG2E;
  // Row statistics of the backward, computed HERE from the query rows this wave already holds (dO in registers, O loaded the
  // same way) and left in the workspace for the dK/dV kernel, which is launched after this one: delta = rowsum(dO . O) and
  // nlse = -lse log2(e) (P = exp2(s c + nlse): no per-tile transform).  A separate pass over O and dO (198 MB at configs[1])
  // did this before.  Invalid query rows: nlse = -big -> P = 0, nothing is accumulated for them.
  float nlse, dlt;
  {
    bf16x8 of[4];
    load_lane_operand(of, o + ooff, osn, q0, NK, lane);
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
  // PEEL: the CLS key's column of dS is one scalar per query (lane): dS[q, 0] = P[q, 0] (dO[q] . v0 - delta[q]), and its term of
  // dQ[q] = sum_k dS[q, k] K[k] is the initial accumulator dS[q, 0] k0.  This wave's CLS-QUERY block: the keys with its own indices.
  const int cls_g = bc.x * 4 + wave, cls_kb = cls_g & 1;
  float nlse0 = 0.f, dlt0 = 0.f;   // the CLS query's row statistics (post-loop block)
  if constexpr (PEEL) {
    bf16x8 do0f[4], o0f[4];
    load_row_bcast(do0f, d_o + ooff0, lane);
    load_row_bcast(o0f, o + ooff0, lane);
    nlse0 = -lse[((int64_t)b * H + head) * N] * LOG2E;
    dlt0 = frag_dot(o0f, do0f);
  }
  if constexpr (PEEL && !(XVIT_PEEL_DEBUG & 2)) {
    bf16x8 k0f[4], v0f[4];
    load_row_bcast(k0f, k + off0, lane);
    load_row_bcast(v0f, v + off0, lane);
    const float p0 = __builtin_amdgcn_exp2f(fmaf(frag_dot(qf, k0f), c, nlse));
    float ds0 = p0 * (frag_dot(dof, v0f) - dlt);
    f32x16 k0a[2];
    load_row_acc(k0a, k + off0, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqacc[0][i] = ds0 * k0a[0][i]; dqacc[1][i] = ds0 * k0a[1][i]; }
    asm volatile("" : "+v"(ds0));   // settle these loads before the tile loop: see settle()
  }
  asm volatile("" : "+v"(nlse0), "+v"(dlt0));

  int stage = 0;
  for (int t = 0; t < ntiles; ++t) {
    ring_wait<BWD_NST, 4>(t, ntiles);
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t are in LDS; every wave is done with tile t-1
    if (t + BWD_NST - 1 < ntiles) {
      int ns = stage + BWD_NST - 1;
      if (ns >= BWD_NST) ns -= BWD_NST;
      lk.issue(smem + ns * 2 * IMG_BYTES, wave, ord.tile(t + BWD_NST - 1));
      lv.issue(smem + ns * 2 * IMG_BYTES + IMG_BYTES, wave, ord.tile(t + BWD_NST - 1));
    }
    const XVIT_LDS char* kimg = smem + stage * 2 * IMG_BYTES;
    const XVIT_LDS char* vimg = kimg + IMG_BYTES;
    stage = stage + 1 == BWD_NST ? 0 : stage + 1;
    if (!wave_active) continue;
    const int nkb = PEEL || (N - t * TILE_ROWS) > 32 ? 2 : 1;   // a tail tile with <= 32 keys needs one key block only (PEEL: 64 | NK, no tail)
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
  store_lane_rows(dqacc, scale, dq + off, sn, qrow, valid, lane);   // dq has q's strides
  if constexpr (PEEL && !(XVIT_PEEL_DEBUG & 1)) {
    if (wave_active) {   // once per wave, after the loop: the CLS query (q0, dO0 in every column) against key block cls_kb of its own tile -> a partial of dQ0
      const XVIT_LDS char* kimg = smem + ord.stage_of(cls_g, BWD_NST) * 2 * IMG_BYTES;
      bf16x8 q0f[4], do0f[4];
      lds_row_bcast(q0f, row0, lane);
      lds_row_bcast(do0f, row0 + ROW0_BYTES, lane);
      f32x16 dq0[2] = {ZERO16, ZERO16};
      dq_block(rd, kimg, kimg + IMG_BYTES, cls_kb, q0f, do0f, c, nlse0, dlt0, dq0);
      store_col0(dq0, pdq + ((((int64_t)b * H + head) * (NK / 32)) + cls_g) * 64, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, dK/dV: one wave = 32 keys, streams Q and dO tiles (+ lse, delta)
// ------------------------------------------------------------------------------------------
constexpr int DKV_STAGE = 2 * IMG_BYTES + 512;  // Q image | dO image | nlse[64] | delta[64]  (all four arrive by LDS-DMA)

// (198 VGPRs: two blocks per CU; bounding it to 168 for a third spills and measured 7 % slower; keeping the query-block loop
// rolled gives 162 VGPRs without spills and three blocks per CU: 552 vs 563 us at B = 126, N = 513, but 50.4 vs 47.0 us at B = 8
// and 1466 vs 1439 us at N = 4097 — not kept)
// one 32-query block of the dK/dV recomputation (keys on the lanes; or, for the CLS key, k0 / v0 repeated in every column):
// S = Q K^T, dP = dO V^T -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS.  st_lse / st_dlt: the tile's 64 query statistics in LDS
__device__ __forceinline__ void dkv_block(const ImgReader& rd, const XVIT_LDS char* qimg, const XVIT_LDS char* doimg, const XVIT_LDS float* st_lse,
                                          const XVIT_LDS float* st_dlt, int qb, const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], float c, int h,
                                          f32x16 (&dkacc)[2], f32x16 (&dvacc)[2]) {
  f32x16 s, dp, pr;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(qimg, qb, ks), kf[ks], ks == 0 ? ZERO16 : s, 0, 0, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd.row_frag(doimg, qb, ks), vf[ks], ks == 0 ? ZERO16 : dp, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 nl = *(const XVIT_LDS f32x4*)(st_lse + qb * 32 + 8 * g + 4 * h);
    const f32x4 dl = *(const XVIT_LDS f32x4*)(st_dlt + qb * 32 + 8 * g + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pv = __builtin_amdgcn_exp2f(fmaf(s[g * 4 + e], c, nl[e]));
      pr[g * 4 + e] = pv;
      s[g * 4 + e] = pv * (dp[g * 4 + e] - dl[e]);  // dS
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

template <bool DROP, bool PEEL>
__global__ __launch_bounds__(256, PEEL ? XVIT_DKV_PEEL_WAVES : 2) void attn_bwd_dkv_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                              int64_t sb, int64_t sn, const bf16* __restrict__ o, const bf16* __restrict__ d_o, int64_t osb, int64_t osn,
                                                              const float* __restrict__ lse, const float* __restrict__ nlse_ws, const float* __restrict__ delta,
                                                              bf16* __restrict__ dk, bf16* __restrict__ dv, int H, int N, float scale, const DropArgs drop_in,
                                                              float* __restrict__ pdk, float* __restrict__ pdv) {
  const DropArgs drop = DROP ? drop_at_run_time(drop_in) : drop_in;   // captured steps: seed + device-side epoch
  static_assert(!PEEL || !DROP, "the CLS peel is built without probability dropout");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BlockCoord bc = xcd_block_coord();
  const int b = bc.b, head = bc.head;
  const int k0 = bc.x * 128 + wave * 32;
  const int NK = PEEL ? N - 1 : N;
  const int64_t off0 = (int64_t)b * sb + head * DH, off = off0 + (PEEL ? sn : 0);
  const int64_t ooff0 = (int64_t)b * osb + head * DH, ooff = ooff0 + (PEEL ? osn : 0);
  const int64_t stat0 = ((int64_t)b * H + head) * N + (PEEL ? 1 : 0);
  const int ntiles = (NK + TILE_ROWS - 1) / TILE_ROWS;

#ifdef XVIT_DEBUG_ATTN_TIMES
  const uint64_t wc_entry = wall_clock64();
  uint64_t tk[4] = {0, 0, 0, 0};
#endif
  TileLoader lq, ldo;
  lq.init(q + off, sn, NK, wave, lane);
  ldo.init(d_o + ooff, osn, NK, wave, lane);
  // per-query statistics of a tile (64 floats each) also arrive by LDS-DMA (4 B per lane: even waves move nlse, odd waves
  // delta), so the loop contains NO ordinary global load whose compiler-inserted vmcnt(0) would drain the tile DMAs.
  // Rows past N read as 0 (buffer bounds): nlse = 0 gives P = 1 there, harmless because dO = 0 and delta = 0.
  const __amdgpu_buffer_rsrc_t rstat = make_rsrc(((wave & 1) == 0 ? nlse_ws : delta) + stat0, clamp_bytes((int64_t)NK * 4));
  auto stage_stats = [&](XVIT_LDS char* st, int tile) {   // waves 2, 3 repeat waves 0, 1 (same bytes): every wave issues 5 DMAs per tile
    glds4(rstat, st + 2 * IMG_BYTES + (wave & 1) * 256, (uint32_t)(lane * 4), (uint32_t)(tile * TILE_ROWS * 4));
  };
  TileOrder ord;
  ord.init(PEEL, bc.x, ntiles);
  XVIT_LDS char* row0 = smem + BWD_NST * DKV_STAGE;   // PEEL: [k0 | v0] (2 ROW0_BYTES)
  if constexpr (PEEL) {
    if (wave == 0) {
      stage_row0(row0, k + off0, lane);
      stage_row0(row0 + ROW0_BYTES, v + off0, lane);
    }
  }
#pragma unroll
  for (int st = 0; st < BWD_NST - 1; ++st)
    if (st < ntiles) {
      lq.issue(smem + st * DKV_STAGE, wave, ord.tile(st));
      ldo.issue(smem + st * DKV_STAGE + IMG_BYTES, wave, ord.tile(st));
      stage_stats(smem + st * DKV_STAGE, ord.tile(st));
    }

  bf16x8 kf[4], vf[4];
  load_lane_operand(kf, k + off, sn, k0, NK, lane);
  load_lane_operand(vf, v + off, sn, k0, NK, lane);
  settle(kf);
  settle(vf);
  ImgReader rd;
  rd.init(lane);
  const int h = lane >> 5;
  const float c = scale * LOG2E;
  const bool wave_active = k0 < NK;   // wave-uniform

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }
  // PEEL: the CLS query's row of P / dS is one scalar per key (lane), and its terms of dK[k] = sum_q dS[q, k] Q[q], dV[k] = sum_q P[q, k] dO[q]
  // are the initial accumulators dS[0, k] q0 and P[0, k] dO0.  This wave's CLS-KEY block: the queries with its own keys' indices
  // (k0 / v0 in every column -> partials of dK0, dV0).
  const int cls_g = bc.x * 4 + wave, cls_qb = cls_g & 1;
  if constexpr (PEEL && !(XVIT_PEEL_DEBUG & 2)) {
    bf16x8 q0f[4], do0f[4], o0f[4];
    load_row_bcast(q0f, q + off0, lane);
    load_row_bcast(do0f, d_o + ooff0, lane);
    load_row_bcast(o0f, o + ooff0, lane);
    const float nlse0 = -lse[((int64_t)b * H + head) * N] * LOG2E, dlt0 = frag_dot(o0f, do0f);
    const float p0 = __builtin_amdgcn_exp2f(fmaf(frag_dot(kf, q0f), c, nlse0));
    float ds0 = p0 * (frag_dot(vf, do0f) - dlt0);
    f32x16 ra[2];
    load_row_acc(ra, q + off0, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkacc[0][i] = ds0 * ra[0][i]; dkacc[1][i] = ds0 * ra[1][i]; }
    load_row_acc(ra, d_o + ooff0, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) { dvacc[0][i] = p0 * ra[0][i]; dvacc[1][i] = p0 * ra[1][i]; }
    asm volatile("" : "+v"(ds0));   // settle these loads before the tile loop: see settle()
  }

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
      lq.issue(nxt, wave, ord.tile(t + BWD_NST - 1));
      ldo.issue(nxt + IMG_BYTES, wave, ord.tile(t + BWD_NST - 1));
      stage_stats(nxt, ord.tile(t + BWD_NST - 1));
    }
    const XVIT_LDS char* qimg = smem + stage * DKV_STAGE;
    stage = stage + 1 == BWD_NST ? 0 : stage + 1;
    const XVIT_LDS char* doimg = qimg + IMG_BYTES;
    const XVIT_LDS float* st_lse = (const XVIT_LDS float*)(qimg + 2 * IMG_BYTES);
    const XVIT_LDS float* st_dlt = st_lse + 64;
    if (!wave_active) continue;
    const int nqb = PEEL || (N - t * TILE_ROWS) > 32 ? 2 : 1;   // a tail tile with <= 32 queries needs one query block only (PEEL: no tail)
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
  const bool valid = krow < NK;
  store_lane_rows(dkacc, scale, dk + off, sn, krow, valid, lane);
  store_lane_rows(dvacc, 1.0f, dv + off, sn, krow, valid, lane);
  if constexpr (PEEL && !(XVIT_PEEL_DEBUG & 1)) {
    if (wave_active) {   // once per wave, after the loop: the CLS key (k0, v0 in every column) against query block cls_qb of its own tile -> partials of dK0, dV0
      const XVIT_LDS char* qimg = smem + ord.stage_of(cls_g, BWD_NST) * DKV_STAGE;
      const XVIT_LDS float* st_lse = (const XVIT_LDS float*)(qimg + 2 * IMG_BYTES);
      bf16x8 k0f[4], v0f[4];
      lds_row_bcast(k0f, row0, lane);
      lds_row_bcast(v0f, row0 + ROW0_BYTES, lane);
      f32x16 dk0[2] = {ZERO16, ZERO16}, dv0[2] = {ZERO16, ZERO16};
      dkv_block(rd, qimg, qimg + IMG_BYTES, st_lse, st_lse + 64, cls_qb, k0f, v0f, c, h, dk0, dv0);
      const int64_t slot = ((((int64_t)b * H + head) * (NK / 32)) + cls_g) * 64;
      store_col0(dk0, pdk + slot, lane);
      store_col0(dv0, pdv + slot, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------
// CLS peel, row 0 of the outputs: one wave per (b, head), lane = d.  The waves' partials are added in slot order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_cls_fwd_merge_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v, int64_t sb,
                                                                 bf16* __restrict__ o, int64_t osb, float* __restrict__ lse, const float* __restrict__ cls_ws,
                                                                 int BH, int H, int N, float scale) {
  const int lane = threadIdx.x & 63, bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= BH) return;
  const int b = bh / H, head = bh - b * H, G = (N - 1) / 32;
  const int64_t off0 = (int64_t)b * sb + head * DH;
  const float c = scale * LOG2E;
  const float* ws = cls_ws + (int64_t)bh * G * CLS_SLOT;
  // the (CLS, CLS) score, raw like the partial maxima, is the starting state; the partials follow in slot order, eight loads deep
  float m = wave_sum(bf2f(q[off0 + lane]) * bf2f(k[off0 + lane]));
  float acc = bf2f(v[off0 + lane]), l = 1.f;
  for (int g0 = 0; g0 < G; g0 += 8) {
    float mg[8], lg[8], og[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* r = ws + (int64_t)min(g0 + j, G - 1) * CLS_SLOT;
      mg[j] = r[64]; lg[j] = r[65]; og[j] = r[lane];
    }
    float mn = m;
#pragma unroll
    for (int j = 0; j < 8; ++j) mn = fmaxf(mn, mg[j]);   // (a repeated last slot cannot raise the maximum)
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
    acc *= alpha; l *= alpha; m = mn;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = g0 + j < G ? __builtin_amdgcn_exp2f((mg[j] - m) * c) : 0.f;
      acc = fmaf(w, og[j], acc);
      l = fmaf(w, lg[j], l);
    }
  }
  o[(int64_t)b * osb + head * DH + lane] = f2bf(acc / l);
  if (lane == 0) lse[(int64_t)bh * N] = m * scale + __logf(l);
}

__global__ __launch_bounds__(256) void attn_cls_bwd_merge_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v, int64_t sb,
                                                                 const bf16* __restrict__ o, const bf16* __restrict__ d_o, int64_t osb, const float* __restrict__ lse,
                                                                 const float* __restrict__ pdq, const float* __restrict__ pdk, const float* __restrict__ pdv,
                                                                 bf16* __restrict__ dq, bf16* __restrict__ dk, bf16* __restrict__ dv, int BH, int H, int N, float scale) {
  const int lane = threadIdx.x & 63, bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= BH) return;
  const int b = bh / H, head = bh - b * H, G = (N - 1) / 32;
  const int64_t off0 = (int64_t)b * sb + head * DH, ooff0 = (int64_t)b * osb + head * DH;
  const float c = scale * LOG2E;
  const float q0 = bf2f(q[off0 + lane]), k0 = bf2f(k[off0 + lane]), v0 = bf2f(v[off0 + lane]), o0 = bf2f(o[ooff0 + lane]), g0 = bf2f(d_o[ooff0 + lane]);
  float sq = 0.f, sk = 0.f, sv = 0.f;
  const int64_t base = (int64_t)bh * G * 64 + lane;
  for (int gg = 0; gg < G; gg += 8) {   // slot order, eight slots (24 loads) in flight
    float a[8], bb[8], cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t at = base + (int64_t)min(gg + j, G - 1) * 64;
      a[j] = pdq[at]; bb[j] = pdk[at]; cc[j] = pdv[at];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (gg + j < G) { sq += a[j]; sk += bb[j]; sv += cc[j]; }
  }
  // the (CLS, CLS) element: P00 = exp(s00 scale - lse0), dS00 = P00 (dO0 . v0 - dO0 . O0)
  const float p00 = __builtin_amdgcn_exp2f(fmaf(wave_sum(q0 * k0), c, -lse[(int64_t)bh * N] * LOG2E));
  const float ds00 = p00 * (wave_sum(g0 * v0) - wave_sum(g0 * o0));
  dq[off0 + lane] = f2bf(scale * fmaf(ds00, k0, sq));
  dk[off0 + lane] = f2bf(scale * fmaf(ds00, q0, sk));
  dv[off0 + lane] = f2bf(fmaf(p00, g0, sv));
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

static DropArgs drop_args(float p, uint64_t seed) { return DropArgs{(uint32_t)(p * 16777216.0f), 1.0f / (1.0f - p), seed, p > 0.f ? drop_epoch_ptr() : nullptr}; }

// xvit_set_option("attn_peel"): 0 = never; 2 = token 0 off the tile grid whenever N = 64 m + 1 and no probability dropout; 1 (default) = that,
// on grids of >= 768 workgroups only.  On a grid the chip holds in a single round a launch takes as long as its slowest workgroup,
// and the peeled workgroup (8 tiles + token-0 prologue and post-loop block, then the merge launch) is not shorter than the grid
// form's (9 tiles).  tools/attn_peel_bench.py, one box, interleaved, forward / backward us, peel vs grid:
//   N = 513:  B = 8 (384 workgroups) 21.4 vs 17.5 / 53 vs 45;  B = 16 (768) 29.6 vs 31.0 / 78 vs 81;  B = 24 39.4 vs 39.8 / 112 vs 108;
//             B = 32 46 vs 52 / 134 vs 143;  B = 48 64 vs 72 / 193 vs 203;  B = 126 167 vs 180 / 508 vs 535
//   N = 4097: B = 4 (1536 workgroups) 252 vs 284 / 716 vs 764;  B = 8 474 vs 511 / 1384 vs 1422
static std::atomic<int> g_attn_peel{1};
namespace xvit { void set_attn_peel(int v) { g_attn_peel.store(v, std::memory_order_relaxed); } }
static bool peel_shape(int B, int H, int N, float dropout_p) {
  const int mode = g_attn_peel.load(std::memory_order_relaxed);
  if (mode == 0 || N <= 1 || (N - 1) % 64 != 0 || dropout_p != 0.f) return false;
  return mode == 2 || (int64_t)B * H * ((N - 1 + 127) / 128) >= 768;
}

extern "C" int64_t xvit_attn_fwd_workspace_bytes(int B, int H, int N) {
  if (B <= 0 || H <= 0 || !peel_shape(B, H, N, 0.f)) return 0;
  return (int64_t)B * H * ((N - 1) / 32) * CLS_SLOT * 4;
}
extern "C" int64_t xvit_attn_bwd_workspace_bytes(int B, int H, int N) {
  if (B <= 0 || H <= 0 || N <= 0) return 0;
  return ((int64_t)2 * B * H * N + (peel_shape(B, H, N, 0.f) ? (int64_t)3 * B * H * ((N - 1) / 32) * 64 : 0)) * 4;
}

extern "C" int xvit_attn_fwd(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, void* o, int64_t osb, int64_t osn, float* lse,
                             int B, int H, int N, int dh, float scale, float dropout_p, uint64_t dropout_seed, float* workspace, int64_t workspace_bytes,
                             xvit_stream_t stream) {
  XVIT_REQUIRE(q && k && v && o && lse, "xvit_attn_fwd: null pointer");
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "xvit_attn_fwd: dropout_p must be in [0, 1)");
  if (int e = attn_check("xvit_attn_fwd", B, H, N, dh, sb, sn, osb, osn)) return e;
  // QB = 1 (32 queries per wave).  QB = 2 was measured: identical throughput at N = 512..4097 (the loop is bound by
  // softmax VALU issue, 12.4 VALU per MFMA at d_h = 64 — not by LDS reads or per-wave ILP) and worse at small batch.
  const dim3 block(256);
  const DropArgs da = drop_args(dropout_p, dropout_seed);
  hipStream_t s = (hipStream_t)stream;
  if (workspace && peel_shape(B, H, N, dropout_p)) {   // token 0 off the tile grid (see "CLS peel" above); without a workspace: the general kernel
    XVIT_REQUIRE(workspace_bytes >= xvit_attn_fwd_workspace_bytes(B, H, N) && ((uintptr_t)workspace & 15) == 0,
                 "xvit_attn_fwd: workspace of %lld bytes, need %lld (xvit_attn_fwd_workspace_bytes), 16-byte aligned", (long long)workspace_bytes,
                 (long long)xvit_attn_fwd_workspace_bytes(B, H, N));
    const dim3 grid((N - 1 + 127) / 128, H, B);
    hipLaunchKernelGGL((attn_fwd_kernel<1, false, true>), grid, block, FWD_NST * 2 * IMG_BYTES + ROW0_BYTES, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                       (bf16*)o, osb, osn, lse, H, N, scale, da, workspace);
    hipLaunchKernelGGL(attn_cls_fwd_merge_kernel, dim3((B * H + 3) / 4), block, 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, (bf16*)o, osb, lse,
                       (const float*)workspace, B * H, H, N, scale);
    return check_launch("xvit_attn_fwd");
  }
  const dim3 grid((N + 127) / 128, H, B);
  if (dropout_p > 0.f)
    hipLaunchKernelGGL((attn_fwd_kernel<1, true, false>), grid, block, FWD_NST * 2 * IMG_BYTES, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                       (bf16*)o, osb, osn, lse, H, N, scale, da, (float*)nullptr);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<1, false, false>), grid, block, FWD_NST * 2 * IMG_BYTES, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                       (bf16*)o, osb, osn, lse, H, N, scale, da, (float*)nullptr);
  return check_launch("xvit_attn_fwd");
}

template <bool DROP, bool PEEL>
static void launch_attn_bwd(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, const void* o, const void* d_o, int64_t osb, int64_t osn,
                            const float* lse, float* ws, void* dq, void* dk, void* dv, int B, int H, int N, float scale, DropArgs da, hipStream_t s) {
  const int64_t total = (int64_t)B * H * N;
  float* delta = ws;            // workspace = [2][B,H,N]: delta | -lse*log2e, then (PEEL) the dQ0 | dK0 | dV0 partials, [B,H,(N-1)/32,64] each
  float* nlse = ws + total;
  const int64_t slots = PEEL ? (int64_t)B * H * ((N - 1) / 32) * 64 : 0;
  float *pdq = ws + 2 * total, *pdk = pdq + slots, *pdv = pdk + slots;
  const int NK = PEEL ? N - 1 : N;
  const dim3 grid((NK + 127) / 128, H, B), block(256);
  // the dQ kernel first: it also leaves delta and -lse log2(e) of every query row in the workspace for the dK/dV kernel
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DROP, PEEL>), grid, block, BWD_NST * 2 * IMG_BYTES + (PEEL ? 2 * ROW0_BYTES : 0), s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                     (const bf16*)o, (const bf16*)d_o, osb, osn, lse, nlse, delta, (bf16*)dq, H, N, scale, da, pdq);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DROP, PEEL>), grid, block, BWD_NST * DKV_STAGE + (PEEL ? 2 * ROW0_BYTES : 0), s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, sn,
                     (const bf16*)o, (const bf16*)d_o, osb, osn, lse, nlse, delta, (bf16*)dk, (bf16*)dv, H, N, scale, da, pdk, pdv);
  if (PEEL)
    hipLaunchKernelGGL(attn_cls_bwd_merge_kernel, dim3((B * H + 3) / 4), block, 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, sb, (const bf16*)o,
                       (const bf16*)d_o, osb, lse, pdq, pdk, pdv, (bf16*)dq, (bf16*)dk, (bf16*)dv, B * H, H, N, scale);
}

extern "C" int xvit_attn_bwd(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, const void* o, const void* d_o, int64_t osb,
                             int64_t osn, const float* lse, float* workspace, int64_t workspace_bytes, void* dq, void* dk, void* dv, int B, int H, int N, int dh,
                             float scale, float dropout_p, uint64_t dropout_seed, xvit_stream_t stream) {
  XVIT_REQUIRE(q && k && v && o && d_o && lse && workspace && dq && dk && dv, "xvit_attn_bwd: null pointer");
  XVIT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "xvit_attn_bwd: dropout_p must be in [0, 1)");
  if (int e = attn_check("xvit_attn_bwd", B, H, N, dh, sb, sn, osb, osn)) return e;
  const bool peel = peel_shape(B, H, N, dropout_p);
  const int64_t need = peel ? xvit_attn_bwd_workspace_bytes(B, H, N) : (int64_t)2 * B * H * N * 4;
  XVIT_REQUIRE(workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, "xvit_attn_bwd: workspace of %lld bytes, need %lld (xvit_attn_bwd_workspace_bytes), 16-byte aligned",
               (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  static const bool lds_opt_in = [] {   // rings deeper than 3 stages need more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * DKV_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * DKV_STAGE);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * DKV_STAGE + 2 * ROW0_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_NST * 2 * IMG_BYTES + 2 * ROW0_BYTES);
    return true;
  }();
  (void)lds_opt_in;
  const DropArgs da = drop_args(dropout_p, dropout_seed);
  if (peel) launch_attn_bwd<false, true>(q, k, v, sb, sn, o, d_o, osb, osn, lse, workspace, dq, dk, dv, B, H, N, scale, da, s);
  else if (dropout_p > 0.f) launch_attn_bwd<true, false>(q, k, v, sb, sn, o, d_o, osb, osn, lse, workspace, dq, dk, dv, B, H, N, scale, da, s);
  else launch_attn_bwd<false, false>(q, k, v, sb, sn, o, d_o, osb, osn, lse, workspace, dq, dk, dv, B, H, N, scale, da, s);
  return check_launch("xvit_attn_bwd");
}
