// bf16 MFMA GEMM with fused epilogue for gfx950 (MI355X).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Operands go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds,
// 1 KiB per wave-instruction, out-of-range -> 0), double buffered, one barrier per K-step.
// An operand is either K-CONTIGUOUS (tile image [128 rows][64 k], 128-B rows, fragments by
// ds_read_b128) or K-STRIDED (tile image [64 k][128 cols], 256-B rows, fragments by
// ds_read_b64_tr_b16), so NT / NN / TN all run without a transpose pass over HBM.
// LDS-DMA writes lane-linear, so both images are XOR-swizzled on the per-lane SOURCE address
// and un-swizzled on the read (bank-conflict-free for both read kinds).
// Epilogue: accumulators -> wave-private LDS slab -> row-contiguous 16-B accesses.
#include "xvit_common.h"

#include <atomic>
#include <mutex>
#include <algorithm>
#include <string>

namespace xvit {

struct GemmParams {
  const bf16* A; const bf16* B; void* C; const float* bias; const float* res; bf16* aux;
  int64_t lda, ldb, ldc, ldr, ldaux;
  int64_t sA, sB, sC, sBias, sR, sAux;
  int M, N, K, k_per_split, split_k, ntm, ntn;
  int c_f32, act, accumulate;
  int res_row_mod, res_row_off, seg_rows, seg_skip, row_off;
  float* slab;          // split-K partial sums [split][batch][M][N], else nullptr
  float* colsum;        // optional [N] fp32: += column sums of the stored C
  float drop_p, drop_inv; uint64_t drop_seed;   // dropout after the activation, before the residual (p == 0: off)
  const uint64_t* drop_epoch;                   // device-side epoch added to the seed at run time (captured steps), or nullptr
  int aux_deriv;                 // aux holds GELU'(z) instead of z: ACT_GELU writes the derivative, ACT_DGELU multiplies by it
  int narrow_epi;                // force the 8-byte-per-lane epilogue (A/B measurements)
  int ncg;                       // gemm_big_kernel: column tiles per super-column of the tile walk
  // Patch rows gathered straight from the volume (xvit_patch_embed_*; reference model_cross.py:193, the einops rearrange
  // 'b c (d p1) (h p2) (w p3) -> b (h w d) (p1 p2 p3 c)' in front of patch_to_embedding): the [rows, dp hp wp] patch matrix is
  // never stored.  mode 1: the A rows of the forward NT product; mode 2: the weight gradient (TN), whose contraction runs over
  // the patch rows in token order with the CLS rows skipped, so both operands take their K-step offset from (sample, 64-token group).
  struct PatchGather {
    int mode;
    int dp, hp, wp;        // patch extents along the volume's first / middle / contiguous axis
    int Dn, Wn;            // patches along the first and the contiguous axis: token t = (h Wn + w) Dn + d
    int Sy, Sz;            // element strides of the middle and the first axis (W, H W)
    int pcount, ntok, cls; // patches per sample, rows per sample in the token matrix (cls + pcount), leading CLS rows (0 | 1)
    int nb;                // samples are ordered [modality][batch]: s -> modality s / nb, batch s % nb
    int64_t sBt, sMd;      // element strides of batch and modality in the volume tensor
    uint32_t vol_bytes;
    uint32_t m_pcount, m_dn, m_wn, m_nb;   // floor(2^32 / divisor): mode 3 divides by these every K-step (fast_div)
  } g;
};

// ---- patch-gather address arithmetic (elements; every volume stays below 2 GiB, so byte offsets fit 31 bits) -------------
constexpr uint32_t GATHER_OOB = 0x80000000u;   // + any K-step offset < 2^31 neither wraps nor falls below num_records
__device__ __forceinline__ uint32_t gather_elem_off(const GemmParams::PatchGather& g, int e) {   // feature f = (p1 hp + p2) wp + p3
  const int hw = g.hp * g.wp, p1 = e / hw, r = e - p1 * hw, p2 = r / g.wp, p3 = r - p2 * g.wp;
  return (uint32_t)(p1 * g.Sz + p2 * g.Sy + p3);
}
__device__ __forceinline__ uint32_t gather_patch_origin(const GemmParams::PatchGather& g, int q) {   // token q of a sample
  const int d = q % g.Dn, t = q / g.Dn, w = t % g.Wn, h = t / g.Wn;
  return (uint32_t)(d * g.dp * g.Sz + h * g.hp * g.Sy + w * g.wp);
}
__device__ __forceinline__ uint32_t gather_sample_origin(const GemmParams::PatchGather& g, int s) {
  const int mod = s / g.nb, b = s - mod * g.nb;
  return (uint32_t)(b * g.sBt + mod * g.sMd);
}

// n / d for any 32-bit n with m = floor(2^32 / d): the estimate is at most one short (n m / 2^32 > n / d - 1), one correction step makes it exact
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t d, uint32_t m, uint32_t& rem) {
  uint32_t q = __umulhi(n, m);
  rem = n - q * d;
  if (rem >= d) { ++q; rem -= d; }
  return q;
}
// mode 3 (weight gradient, ANY patch grid — e.g. 15 patches per axis at configs[2]'s 240^3 volumes, where 64 consecutive tokens are
// neither whole d-columns nor inside one sample): the contraction index is the running patch-token count T over all samples; every
// K-step each lane places its k-rows itself.  T -> (sample, token q) -> dx row (CLS rows skipped) and volume offset of the patch.
struct GatherTok { uint32_t row, vol; bool ok; };
__device__ __forceinline__ GatherTok gather_token(const GemmParams::PatchGather& g, uint32_t T, uint32_t total) {
  GatherTok o;
  o.ok = T < total;
  uint32_t q, b, d, w;
  const uint32_t smp = fast_div(o.ok ? T : 0u, (uint32_t)g.pcount, g.m_pcount, q);
  const uint32_t mod = fast_div(smp, (uint32_t)g.nb, g.m_nb, b);
  const uint32_t t = fast_div(q, (uint32_t)g.Dn, g.m_dn, d);
  const uint32_t h = fast_div(t, (uint32_t)g.Wn, g.m_wn, w);
  o.row = smp * (uint32_t)g.ntok + (uint32_t)g.cls + q;
  o.vol = (uint32_t)(b * g.sBt + mod * g.sMd) + d * (uint32_t)(g.dp * g.Sz) + h * (uint32_t)(g.hp * g.Sy) + w * (uint32_t)g.wp;
  return o;
}

// compile-time activation codes of the wide epilogue kernels for aux_mode 1 (the public codes + p.aux_deriv everywhere else)
constexpr int ACT_GELU_D = 3;    // C = gelu(z), aux <- gelu'(z)
constexpr int ACT_MULAUX = 4;    // C = acc * aux

// ---- the fused epilogue, shared by both tile kernels and the split-K reduce kernel -------------
// v: 4 consecutive output columns of one row (fp32 accumulators).  Returns the value stored.
template <int ACT, bool DROP = false>
__device__ __forceinline__ f32x4 epilogue_apply(const GemmParams& p, f32x4 v, int row, int col, int64_t cb, const float* bias, const float* res,
                                                bf16* aux) {
  if (bias) v += *(const f32x4*)(bias + col);
  if (ACT == XVIT_ACT_GELU) {
    f32x4 d;
    const f32x4 zv = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float a_, d_; gelu_and_grad(zv[e], a_, d_); v[e] = a_; d[e] = d_; }
    if (aux) {   // the pre-activation, or (aux_mode 1) the derivative the backward will multiply by
      const f32x4 s = p.aux_deriv ? d : zv;
      bf16x4 z = {f2bf(s[0]), f2bf(s[1]), f2bf(s[2]), f2bf(s[3])};
      *(bf16x4*)(aux + (int64_t)row * p.ldaux + col) = z;
    }
  } else if (ACT == XVIT_ACT_DGELU) {
    const bf16x4 z = *(const bf16x4*)(aux + (int64_t)row * p.ldaux + col);
    if (p.aux_deriv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= bf2f(z[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(bf2f(z[e]));
    }
  }
  if (DROP) {   // mask keyed by (seed, batch-local element index): regenerated, never stored
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
    const uint64_t idx = (uint64_t)(cb / (p.sC ? p.sC : 1)) * ((uint64_t)p.M * p.N) + (uint64_t)row * p.N + col;
    const uint64_t seed = drop_seed_at(p.drop_seed, p.drop_epoch);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (hash32(seed, idx + e) & 0xFFFFFFu) >= thr ? v[e] * p.drop_inv : 0.f;
  }
  if (res) {
    const int rr = p.res_row_mod > 0 ? p.res_row_off + (row % p.res_row_mod) : row;
    v += *(const f32x4*)(res + (int64_t)rr * p.ldr + col);
  }
  const int64_t orow = p.seg_rows > 0 ? (int64_t)row + (row / p.seg_rows) * p.seg_skip + p.row_off : row;
  if (p.c_f32) {
    float* dst = (float*)p.C + cb + orow * p.ldc + col;
    if (p.accumulate) v += *(const f32x4*)dst;
    *(f32x4*)dst = v;
  } else {
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *(bf16x4*)((bf16*)p.C + cb + orow * p.ldc + col) = o;
  }
  return v;
}
__device__ __forceinline__ f32x4 epilogue_apply_rt(const GemmParams& p, f32x4 v, int row, int col, int64_t cb, const float* bias, const float* res,
                                                   bf16* aux) {
  if (p.drop_p > 0.f) {
    if (p.act == XVIT_ACT_GELU) return epilogue_apply<XVIT_ACT_GELU, true>(p, v, row, col, cb, bias, res, aux);
    if (p.act == XVIT_ACT_DGELU) return epilogue_apply<XVIT_ACT_DGELU, true>(p, v, row, col, cb, bias, res, aux);
    return epilogue_apply<XVIT_ACT_NONE, true>(p, v, row, col, cb, bias, res, aux);
  }
  if (p.act == XVIT_ACT_GELU) return epilogue_apply<XVIT_ACT_GELU>(p, v, row, col, cb, bias, res, aux);
  if (p.act == XVIT_ACT_DGELU) return epilogue_apply<XVIT_ACT_DGELU>(p, v, row, col, cb, bias, res, aux);
  return epilogue_apply<XVIT_ACT_NONE>(p, v, row, col, cb, bias, res, aux);
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int OPER_BYTES = BM * BK * 2;        // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OPER_BYTES;    // 32 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // 64 KiB
constexpr int EPI_LD = 68;                     // floats per row of the epilogue slab (64 + 4 pad)

// swizzle of the 16-B chunk index inside a 256-B row of a K-strided image (serves tr reads)
__device__ __forceinline__ int swz_ks(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }
// swizzle of the 16-B chunk index inside a 128-B row of a K-contiguous image (serves b128 reads)
__device__ __forceinline__ int swz_kc(int row) { return (row >> 1) & 7; }

template <bool KS>
struct OperandLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[4];
  uint32_t kstep;  // soffset increment per K-tile, bytes
  // ld in elements; `wave` handles pieces 4*wave .. 4*wave+3 of the 16 KiB image
  __device__ __forceinline__ void init(const bf16* tile_base, int64_t bytes_avail, int64_t ld, int wave, int lane) {
    rsrc = make_rsrc(tile_base, clamp_bytes(bytes_avail));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;
      if (KS) {
        const int krow = piece * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ swz_ks(krow);
        voff[j] = (uint32_t)(krow * ld * 2 + chunk * 16);
      } else {
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz_kc(row);
        voff[j] = (uint32_t)(row * ld * 2 + chunk * 16);
      }
    }
    kstep = KS ? (uint32_t)(BK * ld * 2) : (uint32_t)(BK * 2);
  }
  __device__ __forceinline__ void issue(XVIT_LDS char* image, int wave, int kt) const {
    const uint32_t soff = (uint32_t)kt * kstep;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(rsrc, image + (wave * 4 + j) * 1024, voff[j], soff);
  }
};

// Fragment addressing.  `sub` = this wave's 64-wide slice (0/1) of the tile's 128 rows/cols.
template <bool KS>
struct FragReader {
  uint32_t off[KS ? 8 : 2];
  __device__ __forceinline__ void init(int sub, int lane) {
    if (KS) {
      const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int krow = 8 * g + 4 * s + q;
          const int ch = sub * 8 + t * 2 + (p >> 1);
          off[t * 2 + s] = (uint32_t)(256 * krow + 16 * (ch ^ swz_ks(krow)) + 8 * (p & 1));
        }
    } else {
      const int row = sub * 64 + (lane & 15);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) off[kk] = (uint32_t)(row * 128 + (((kk * 4 + (lane >> 4)) ^ swz_kc(row)) << 4));
    }
  }
  // fragment for 16-wide tile t (0..3), k-step kk (0..1) of the staged 64-deep K tile
  __device__ __forceinline__ bf16x8 read(const XVIT_LDS char* image, int t, int kk) const {
    if (KS) {
      const s16x4 lo = lds_read_tr16(image + off[t * 2 + 0] + kk * 32 * 256);
      const s16x4 hi = lds_read_tr16(image + off[t * 2 + 1] + kk * 32 * 256);
      s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    } else {
      return *(const XVIT_LDS bf16x8*)(image + off[kk] + t * 16 * 128);
    }
  }
};

template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (its L2); give each XCD a contiguous
  // run of tiles, n fastest, so an A row-panel is fetched by one XCD and B stays L2-resident.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = logical / p.ntn, tn = logical - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int batch = blockIdx.z / p.split_k, split = blockIdx.z - batch * p.split_k;
  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = (k_end - k_begin + BK - 1) / BK;
  OperandLoader<A_KS> la;   // nk may be 0 for a trailing split: it still writes its (zero) partial tile
  OperandLoader<B_KS> lb;
  {
    const bf16* Ab = p.A + batch * p.sA;
    if (A_KS) {  // stored [K, M]
      const bf16* base = Ab + (int64_t)k_begin * p.lda + m0;
      la.init(base, ((int64_t)(k_end - 1 - k_begin) * p.lda + (p.M - m0)) * 2, p.lda, wave, lane);
    } else {  // stored [M, K]
      const bf16* base = Ab + (int64_t)m0 * p.lda + k_begin;
      la.init(base, ((int64_t)(p.M - 1 - m0) * p.lda + (k_end - k_begin)) * 2, p.lda, wave, lane);
    }
    const bf16* Bb = p.B + batch * p.sB;
    if (B_KS) {  // stored [K, N]
      const bf16* base = Bb + (int64_t)k_begin * p.ldb + n0;
      lb.init(base, ((int64_t)(k_end - 1 - k_begin) * p.ldb + (p.N - n0)) * 2, p.ldb, wave, lane);
    } else {  // stored [N, K]
      const bf16* base = Bb + (int64_t)n0 * p.ldb + k_begin;
      lb.init(base, ((int64_t)(p.N - 1 - n0) * p.ldb + (k_end - k_begin)) * 2, p.ldb, wave, lane);
    }
  }
  FragReader<A_KS> fa;
  FragReader<B_KS> fb;
  fa.init(wr, lane);
  fb.init(wc, lane);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    la.issue(smem, wave, 0);
    lb.issue(smem + OPER_BYTES, wave, 0);
  }

  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed for every wave; every wave is done reading tile kt-1
    if (kt + 1 < nk) {
      XVIT_LDS char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
      la.issue(nxt, wave, kt + 1);
      lb.issue(nxt + OPER_BYTES, wave, kt + 1);
    }
    const XVIT_LDS char* sa = smem + (kt & 1) * STAGE_BYTES;
    const XVIT_LDS char* sb = sa + OPER_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = fa.read(sa, t, kk);
#pragma unroll
      for (int t = 0; t < 4; ++t) bfr[t] = fb.read(sb, t, kk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---------------- epilogue: accumulators -> wave-private LDS slab -> coalesced rows -------------
  __syncthreads();  // every wave has finished reading the staging buffers
  XVIT_LDS float* slab = (XVIT_LDS float*)(smem + wave * 16384);
  const int64_t cb = batch * p.sC;
  const float* bias = p.bias ? p.bias + batch * p.sBias : nullptr;
  const float* res = p.res ? p.res + batch * p.sR : nullptr;
  bf16* aux = p.aux ? p.aux + batch * p.sAux : nullptr;
  float* csum = p.colsum ? p.colsum + batch * p.sBias : nullptr;
  const int nbatch = gridDim.z / p.split_k;
  float* part = p.slab ? p.slab + ((int64_t)split * nbatch + batch) * (int64_t)p.M * p.N : nullptr;

  // column sums of the stored values: a lane's 4 columns are the same in every row it visits, so they are summed in registers,
  // then over the 4 lanes that share the columns, and reach memory as ONE atomic per column and wave (per-element atomics made a
  // 4104 x 3072 GELU' dgrad take 476 us instead of 34)
  f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};
  const int col = n0 + wc * 64 + (lane & 15) * 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          slab[(mi * 16 + (lane >> 4) * 4 + r) * EPI_LD + j * 16 + (lane & 15)] = acc[pass * 2 + mi][j][r];
    const int row_base = m0 + wr * 64 + pass * 32;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
      const int rl = it * 4 + (lane >> 4);
      const int row = row_base + rl;
      const f32x4 v = *(const XVIT_LDS f32x4*)(slab + rl * EPI_LD + (lane & 15) * 4);
      if (row < p.M && col < p.N) {
        if (part) {
          *(f32x4*)(part + (int64_t)row * p.N + col) = v;   // split-K partial: epilogue runs in the reduce kernel
        } else {
          cs4 += epilogue_apply_rt(p, v, row, col, cb, bias, res, aux);
        }
      }
    }
  }
  if (csum && !part) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = cs4[e];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      cs4[e] = t;
    }
    if (lane < 16 && col < p.N) {
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(csum + col + e, cs4[e]);
    }
  }
}

// ==========================================================================================
// Big-tile kernel: 256x256x64 block tile, 512 threads = 8 waves (2 along M x 4 along N), each wave
// 128x64 = 8x4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator registers).  128 FLOP per byte
// staged into LDS (the 128^2 tile's 64 FLOP/B is bound by the L2->LDS rate, not by MFMA).  One
// block per CU (128 KiB LDS, 2 stages).  The MFMA operand roles are swapped (weights as "A",
// activations as "B"), so a lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one
// row: the epilogue runs straight from registers with 8/16-byte stores — no LDS round trip.
// Split-K writes fp32 partial tiles to a slab (plain stores); splitk_reduce_kernel sums them in a
// fixed order (bit-reproducible, no atomics).
// ==========================================================================================
constexpr int TBM = 256, TBN = 256;
constexpr int T_OPER = TBM * BK * 2;     // 32 KiB per operand per stage
constexpr int T_STAGE = 2 * T_OPER;      // 64 KiB
constexpr int T_LDS = 2 * T_STAGE;       // 128 KiB

// Weight-gradient K-step: LDS k-row kr of both operands holds token gather_krow_token(kr) of the 64-token group (the
// contraction order is free as long as dx and the volume agree).  One LDS-DMA instruction fills two k-rows; pairing the
// tokens (w, d) and (w + 1, d) puts their 16-voxel runs side by side: 64 contiguous bytes per (p1, p2) instead of two
// 32-byte pieces 512 KiB apart.
__device__ __forceinline__ int gather_krow_token(const GemmParams::PatchGather& g, int kr) {
  const int nw = 64 / g.Dn;
  if (nw & 1) return kr;
  const int a = kr >> 1, b = kr & 1, wa = a / g.Dn, d = a - wa * g.Dn;
  return d + g.Dn * (2 * wa + b);
}

template <bool KS>
struct BigLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[4];
  uint32_t kstep;
  // 32 KiB image = 32 pieces of 1 KiB; wave w moves pieces 4w .. 4w+3
  // perm != nullptr (weight gradient of the gathered patch embedding): LDS k-row kr is source row gather_krow_token(kr)
  __device__ __forceinline__ void init(const bf16* tile_base, int64_t bytes_avail, int64_t ld, int wave, int lane, const GemmParams::PatchGather* perm = nullptr) {
    rsrc = make_rsrc(tile_base, clamp_bytes(bytes_avail));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;
      if (KS) {  // image [64 k][256 cols], 512-B rows: a piece is 2 k-rows
        const int krow = piece * 2 + (lane >> 5);
        const int chunk = (lane & 31) ^ swz_ks(krow);
        const int srow = perm ? gather_krow_token(*perm, krow) : krow;
        voff[j] = (uint32_t)(srow * ld * 2 + chunk * 16);
      } else {   // image [256 rows][64 k], 128-B rows: a piece is 8 rows
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz_kc(row);
        voff[j] = (uint32_t)(row * ld * 2 + chunk * 16);
      }
    }
    kstep = KS ? (uint32_t)(BK * ld * 2) : (uint32_t)(BK * 2);
  }
  // mode 1 (KS = false): row m0 + r of the token matrix = patch (sample, token) or a CLS / out-of-range row (zeros)
  __device__ __forceinline__ void init_gather_rows(const GemmParams::PatchGather& g, const bf16* vol, int m0, int M, int wave, int lane) {
    rsrc = make_rsrc(vol, g.vol_bytes);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // LDS row rl = 32 wave + 8 j + ii holds tile row 32 wave + (ii % 4) 8 + 2 j + ii / 4: the 8 rows of one DMA instruction are
      // two groups of 4 tokens that are neighbours along W, so each of its 4 x 2 runs covers a whole 128-byte line (4 x 32 B)
      // instead of 32 lines of 32 bytes; the epilogue un-permutes in its LDS transpose (wave_tile_epilogue<PERM>)
      const int ii = lane >> 3, rl = (wave * 4 + j) * 8 + ii, row = m0 + wave * 32 + (ii & 3) * 8 + 2 * j + (ii >> 2);
      const int chunk = (lane & 7) ^ swz_kc(rl);
      const int smp = row / g.ntok, n = row - smp * g.ntok;
      voff[j] = (row < M && n >= g.cls) ? 2u * (gather_sample_origin(g, smp) + gather_patch_origin(g, n - g.cls) + gather_elem_off(g, chunk * 8)) : GATHER_OOB;
    }
    kstep = 0;
  }
  // mode 2 (KS = true): k-row = token (64 consecutive tokens of one sample per K-step), column = feature n0 + c
  __device__ __forceinline__ void init_gather_ks(const GemmParams::PatchGather& g, const bf16* vol, int n0, int N, int wave, int lane) {
    rsrc = make_rsrc(vol, g.vol_bytes);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int krow = (wave * 4 + j) * 2 + (lane >> 5);
      const int e = n0 + (((lane & 31) ^ swz_ks(krow)) << 3);
      const int tok = gather_krow_token(g, krow);
      voff[j] = e < N ? 2u * ((uint32_t)((tok % g.Dn) * g.dp * g.Sz + (tok / g.Dn) * g.wp) + gather_elem_off(g, e)) : GATHER_OOB;
    }
    kstep = 0;
  }
  // mode 3: k-row kr of K-step ktg is patch token T = 64 ktg + kr.  KS operand A = dx (row = token row, this lane's 16-byte column chunk),
  // KS operand B = the volume (this lane's feature chunk of the token's patch).  `fixed` = the lane's part that does not move.
  __device__ __forceinline__ void place_tokens(const GemmParams::PatchGather& g, int ktg, uint32_t total, int64_t ld, bool volume, const uint32_t (&fixed)[4], int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int krow = (wave * 4 + j) * 2 + (lane >> 5);
      const GatherTok t = gather_token(g, (uint32_t)ktg * 64u + (uint32_t)krow, total);
      voff[j] = !t.ok || fixed[j] == GATHER_OOB ? GATHER_OOB : (volume ? 2u * t.vol + fixed[j] : t.row * (uint32_t)(ld * 2) + fixed[j]);
    }
  }
  __device__ __forceinline__ void issue_at(XVIT_LDS char* image, int wave, uint32_t soff) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(rsrc, image + (wave * 4 + j) * 1024, voff[j], soff);
  }
  __device__ __forceinline__ void issue(XVIT_LDS char* image, int wave, int kt) const { issue_at(image, wave, (uint32_t)kt * kstep); }
};

// K-step offsets (bytes) of the gathered operands; ktg = K-step index counted from k = 0
__device__ __forceinline__ uint32_t gather_soff_rows(const GemmParams::PatchGather& g, int ktg) { return 2u * gather_elem_off(g, ktg * 64); }
struct GatherStep { uint32_t a, b; };
__device__ __forceinline__ GatherStep gather_soff_wgrad(const GemmParams::PatchGather& g, int ktg, int64_t lda) {
  const int spk = g.pcount >> 6, smp = ktg / spk, r = ktg - smp * spk;     // 64-token group r of sample smp
  const int wsteps = (g.Dn * g.Wn) >> 6, h = r / wsteps, wpart = (r - h * wsteps) * (64 / g.Dn);
  GatherStep o;
  o.a = (uint32_t)(((int64_t)smp * g.ntok + g.cls + r * 64) * lda * 2);
  o.b = 2u * (gather_sample_origin(g, smp) + (uint32_t)(h * g.hp * g.Sy + wpart * g.wp));
  return o;
}

// NT = number of 16-wide tiles this wave reads from the image (8 along M, 4 along N);
// `first` = index of the wave's first 16-wide tile inside the 256-wide image.
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_dbg;
template <bool KS, int NT_>
struct BigFrag {
  uint32_t off[KS ? 2 * NT_ : 2];
  __device__ __forceinline__ void init(int first, int lane) {
    if (KS) {
      const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
#pragma unroll
      for (int t = 0; t < NT_; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int krow = 8 * g + 4 * s + q;
          const int ch = (first + t) * 2 + (p >> 1);
          off[t * 2 + s] = (uint32_t)(512 * krow + 16 * (ch ^ swz_ks(krow)) + 8 * (p & 1));
        }
    } else {
      const int row = first * 16 + (lane & 15);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) off[kk] = (uint32_t)(row * 128 + (((kk * 4 + (lane >> 4)) ^ swz_kc(row)) << 4));
    }
  }
  __device__ __forceinline__ bf16x8 read(const XVIT_LDS char* image, int t, int kk) const {
#ifdef XVIT_DEBUG_NO_LDSREAD   // energy decomposition: MFMAs on register-resident pseudo-random operands, no fragment reads
    u32x4_dbg v = {off[0] * 2654435761u, off[0] * 40503u + 12345u, (uint32_t)t * 0x9E3779B9u ^ off[0], 0x3F803F80u ^ (off[0] << 3)};
    v &= 0xBFFFBFFFu;            // keep the bf16 exponents below 2^1: finite products
    asm volatile("" : "+v"(v));
    return __builtin_bit_cast(bf16x8, v);
#endif
    if (KS) {
      const s16x4 lo = lds_read_tr16(image + off[t * 2 + 0] + kk * 32 * 512);
      const s16x4 hi = lds_read_tr16(image + off[t * 2 + 1] + kk * 32 * 512);
      s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    } else {
      return *(const XVIT_LDS bf16x8*)(image + off[kk] + t * 16 * 128);
    }
  }
};

// scheduling pattern of big_tile_mma: per step, the LDS reads issued in it, then its 4 MFMAs (the builtin wants literals)
template <int S, int PF, int NA, int NB>
__device__ __forceinline__ void mma_sched_steps() {
  constexpr int reads = (S + PF < 16 ? NA : 0) + (S >= 4 && S < 8 ? NB : 0);
  if constexpr (reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  if constexpr (S + 1 < 16) mma_sched_steps<S + 1, PF, NA, NB>();
}

// One 256 x 256 x 64 K-tile for a wave's 128 x 64 sub-tile: 16 steps (2 k-halves x 8 row tiles) of 4 MFMAs.
// The fragment reads are software-pipelined by hand: A fragments run PF steps ahead of their MFMAs, the
// second k-half's B fragments are slipped in during steps 4..7, and sched_group_barrier pins that order.
// (Left alone, hipcc folds all A fragments into ONE register and emits ds_read -> lgkmcnt(0) -> 4 MFMA
// sixteen times per tile, exposing every LDS round trip; the main loop then sits at 49 % of the MFMA peak.)
// The 8 LDS-DMA instructions that prefetch the next K-tile are issued one per step in steps 0..7 (`ra` / `rb`
// are zero-size descriptors in the last iteration): issued back to back at the top of the iteration they
// stall the wave ~670 cycles in the in-order vector-memory issue queue before its first MFMA (same-box A/B: one
// per step -3 % on the model's 15 GEMM shapes; two per step or every other step are both slower than that).
template <bool A_KS, bool B_KS>
__device__ __forceinline__ void big_tile_mma(const XVIT_LDS char* sa, const XVIT_LDS char* sb, const BigFrag<A_KS, 8>& fa, const BigFrag<B_KS, 4>& fb,
                                             f32x4 (&acc)[8][4], const BigLoader<A_KS>& la, const BigLoader<B_KS>& lb, __amdgpu_buffer_rsrc_t ra,
                                             __amdgpu_buffer_rsrc_t rb, XVIT_LDS char* nxt, int wave, uint32_t soff_a, uint32_t soff_b) {
  constexpr int PF = 2;                       // prefetch distance in steps
  constexpr int NA = A_KS ? 2 : 1, NB = B_KS ? 2 : 1;   // LDS instructions per fragment
  bf16x8 b0[4], b1[4], a[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) b0[t] = fb.read(sb, t, 0);
#pragma unroll
  for (int s = 0; s < PF; ++s) a[s] = fa.read(sa, s & 7, s >> 3);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
#ifndef XVIT_DEBUG_NO_DMA
    if (s < 4) glds16(ra, nxt + (wave * 4 + s) * 1024, la.voff[s], soff_a);
    else if (s < 8) glds16(rb, nxt + T_OPER + (wave * 4 + s - 4) * 1024, lb.voff[s - 4], soff_b);
#endif
    if (s + PF < 16) a[s + PF] = fa.read(sa, (s + PF) & 7, (s + PF) >> 3);
    if (s >= 4 && s < 8) b1[s - 4] = fb.read(sb, s - 4, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[s & 7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s < 8 ? b0[j] : b1[j], a[s], acc[s & 7][j], 0, 0, 0);  // D'[n][m]
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 4 * NB + PF * NA, 0);
  mma_sched_steps<0, PF, NA, NB>();
}

// ---- big-tile epilogue: accumulators -> per-wave LDS transpose -> row-contiguous global accesses ----------
// In the accumulators a lane owns 4 consecutive columns of rows (lane & 15) + 16 I: neighbouring lanes sit in
// DIFFERENT rows, so a store straight from registers scatters 64 separate 8/16-byte pieces over 16 cache lines
// (and the residual / aux loads likewise).  Each wave therefore bounces its 128 x 64 fp32 tile, 64 rows at a
// time, through its own 16 KiB slice of the (now idle) stage buffers: written as [64 rows][16 chunks of 16 B]
// with chunk ^= row & 15 (conflict-free for the column-wise writes and the row-wise reads), read back with 16
// lanes per row.  One global instruction then covers 4 rows x 256 contiguous bytes (fp32) or 4 x 128 (bf16),
// the bias is a per-lane constant, and a lane's rows advance by 4 per body, so the row -> offset maps
// (residual row modulo, output segment gaps) are stepped incrementally instead of divided per body.
// Every global access goes through a buffer resource with the out-of-range offset trick instead of a branch
// (loads return 0, stores are dropped), so the bodies form straight-line code and their loads batch.
// Cache policy of the epilogue's global traffic (the `aux` operand of the buffer instructions: 0 = default, 2 = nt, 16 = sc1).
// Outputs and the z operand of GELU' are streamed once; written through / read non-temporally they stop evicting the A / W
// panels the tiles of an XCD share through its 4 MiB L2.
// Measured at configs[1] (tools/gemm_model_bench.py, same box, ms of GEMM time per step): default policy 25.0, sc1 stores 24.8,
// nt stores + nt z loads 24.2, + nt residual loads 24.0 (FFN1 with its two bf16 outputs 411 -> 354 us, GELU' dgrad 413 -> 381,
// out-proj 156 -> 146).
#ifndef XVIT_EPI_STORE_AUX
#define XVIT_EPI_STORE_AUX 2
#endif
#ifndef XVIT_EPI_LOAD_AUX
#define XVIT_EPI_LOAD_AUX 2
#endif
#ifndef XVIT_EPI_RES_AUX
#define XVIT_EPI_RES_AUX 2
#endif
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
constexpr uint32_t OOB = 0xFFFFFFF0u;
constexpr int EPI_WAVE_BYTES = 64 * 256;   // 64 rows x 64 fp32

struct BigEpi {
  __amdgpu_buffer_rsrc_t rc, raux, rres, rslab;
  bool has_res, has_aux, to_slab;
  uint32_t col;            // this lane's first output column (4 consecutive)
  bool col_ok;
  f32x4 bias;
  f32x4 csum;              // running column sums of the values this lane stored (for p.colsum)
  float* colsum_dst;       // p.colsum + batch offset + this lane's first column
  uint64_t drop_base;
  // state of the lane's current row, advanced by 4 rows per body (32-bit byte offsets: the host guarantees
  // every addressed tensor stays below 2 GiB per batch)
  uint32_t row, c, aux, res, slab;
  uint32_t rrem, rmod, srem, smod;         // row % res_row_mod, row % seg_rows and their moduli (huge when unused)
  uint32_t c_step, aux_step, res_step, slab_step, c_gap, res_wrap;
};

// The z loads of the GELU' epilogue are software-pipelined one region (4 bodies) ahead of the stores (two regions ahead measured the same): on gfx9 loads and
// stores retire through ONE in-order vmcnt, so a load issued right after a region's stores can only be waited for once
// those stores have been acknowledged by memory.  Loading z inside each body cost the GELU' dgrad ~15 us per tile round
// (505 TFLOP/s where the same shape without an epilogue load reaches 860).  LoadCursor is a second copy of the row state
// that runs one region ahead of the store cursor in BigEpi.  (The fp32 residual loads of the out-proj / FFN2 epilogues
// stay in the body: those epilogues are bound by their 8 B/element of HBM traffic, and prefetching them as well doubled
// the number of inlined epilogue variants — minutes of compile time and register spills.)
struct LoadCursor { uint32_t row, aux; };
struct EpiLoads { u32x2_t aux[4]; };

__device__ __forceinline__ void big_epi_issue_aux(const GemmParams& p, const BigEpi& e, LoadCursor& lc, EpiLoads& L) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const bool ok = e.col_ok && lc.row < (uint32_t)p.M;
    L.aux[b] = __builtin_amdgcn_raw_buffer_load_b64(e.raux, ok ? lc.aux : OOB, 0, XVIT_EPI_LOAD_AUX);
    lc.row += 4; lc.aux += e.aux_step;
  }
}

template <int ACT, bool DROP>
__device__ __forceinline__ void big_epi_body(const GemmParams& p, BigEpi& e, f32x4 v, u32x2_t auxv) {
  const bool ok = e.col_ok && e.row < (uint32_t)p.M;
  if (e.to_slab) {   // split-K partial sums, [M][N] fp32
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), e.rslab, ok ? e.slab : OOB, 0, XVIT_EPI_STORE_AUX);
  } else {
    v += e.bias;
    if (ACT == XVIT_ACT_GELU) {
      f32x4 d;
      const f32x4 zv = v;
#pragma unroll
      for (int c = 0; c < 4; ++c) { float a_, d_; gelu_and_grad(zv[c], a_, d_); v[c] = a_; d[c] = d_; }
      if (e.has_aux) {
        const f32x4 sv = p.aux_deriv ? d : zv;
        bf16x4 z = {f2bf(sv[0]), f2bf(sv[1]), f2bf(sv[2]), f2bf(sv[3])};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, z), e.raux, ok ? e.aux : OOB, 0, XVIT_EPI_STORE_AUX);
      }
    } else if (ACT == XVIT_ACT_DGELU) {
      const bf16x4 z = __builtin_bit_cast(bf16x4, auxv);
      if (p.aux_deriv) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] *= bf2f(z[c]);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] *= dgelu_f(bf2f(z[c]));
      }
    }
    if (DROP) {
      const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
      const uint64_t idx = e.drop_base + (uint64_t)e.row * p.N + e.col;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (hash32(drop_seed_at(p.drop_seed, p.drop_epoch), idx + c) & 0xFFFFFFu) >= thr ? v[c] * p.drop_inv : 0.f;
    }
    if (e.has_res) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(e.rres, ok ? e.res : OOB, 0, XVIT_EPI_RES_AUX));
    if (p.c_f32) {
      const uint32_t off = ok ? e.c : OOB;
      if (p.accumulate) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(e.rc, off, 0, 0));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), e.rc, off, 0, XVIT_EPI_STORE_AUX);
    } else {
      bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o), e.rc, ok ? e.c : OOB, 0, XVIT_EPI_STORE_AUX);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) e.csum[c] += ok ? v[c] : 0.f;   // the lane's 4 columns are the same in every body
  }
  // next body: 4 rows further down
  e.row += 4; e.c += e.c_step; e.aux += e.aux_step; e.res += e.res_step; e.slab += e.slab_step;
  e.rrem += 4; e.srem += 4;
  if (e.rrem >= e.rmod) { e.rrem -= e.rmod; e.res -= e.res_wrap; }
  if (e.srem >= e.smod) { e.srem -= e.smod; e.c += e.c_gap; }
}

// Region R (0..7) of the wave's 128-row tile = 4 bodies = rows 16 R .. 16 R + 15; regions 0-3 and 4-7 share one LDS
// transpose pass each (64 rows: accumulators -> LDS column-wise, read back row-wise).  `cur` holds this region's
// pre-issued z loads (GELU' only); the next region's are issued before this region's stores.
template <int ACT, bool DROP, int R, bool PERM = false>
__device__ __forceinline__ void big_epi_regions(const GemmParams& p, BigEpi& e, LoadCursor& lc, EpiLoads& cur, const f32x4 (&acc)[8][4], XVIT_LDS char* slice,
                                                const uint32_t (&woff)[8], const uint32_t (&roff)[4]) {
  if constexpr ((R & 3) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // PERM (gathered patch rows): accumulator row 16 i + r of the pass is output row 32 (i / 2) + perm32(16 (i % 2) + r)
        if constexpr (PERM) *(XVIT_LDS f32x4*)(slice + (i >> 1) * 8192 + woff[(i & 1) * 4 + j]) = acc[(R >> 2) * 4 + i][j];
        else *(XVIT_LDS f32x4*)(slice + i * 4096 + woff[j]) = acc[(R >> 2) * 4 + i][j];
      }
  }
  EpiLoads nxt;
  if constexpr (ACT == XVIT_ACT_DGELU && R < 7) big_epi_issue_aux(p, e, lc, nxt);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int it = (R & 3) * 4 + b;
    const f32x4 v = *(const XVIT_LDS f32x4*)(slice + it * 1024 + roff[it & 3]);
    big_epi_body<ACT, DROP>(p, e, v, cur.aux[b]);
  }
  __builtin_amdgcn_sched_barrier(0);   // one region per scheduling window
  if constexpr (R < 7) big_epi_regions<ACT, DROP, R + 1, PERM>(p, e, lc, nxt, acc, slice, woff, roff);
}

template <int ACT, bool DROP, bool PERM = false>
__device__ __forceinline__ void big_epilogue(const GemmParams& p, BigEpi& e, const f32x4 (&acc)[8][4], XVIT_LDS char* slice, const uint32_t (&woff)[8],
                                             const uint32_t (&roff)[4]) {
  LoadCursor lc = {e.row, e.aux};
  EpiLoads first;
  if constexpr (ACT == XVIT_ACT_DGELU) big_epi_issue_aux(p, e, lc, first);
  big_epi_regions<ACT, DROP, 0, PERM>(p, e, lc, first, acc, slice, woff, roff);
  if (p.colsum && !e.to_slab) {   // += column sums of the stored tile: lanes l, l+16, l+32, l+48 share their 4 columns
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t = e.csum[c];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      e.csum[c] = t;
    }
    if ((threadIdx.x & 63) < 16 && e.col_ok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) unsafeAtomicAdd(e.colsum_dst + c, e.csum[c]);
    }
  }
}

// ---- wide epilogue for bf16 outputs: 16 bytes per lane -----------------------------------------------------------
// With 4 columns per lane a bf16 store (or z load) moves 8 bytes per lane: 64 store instructions per wave tile (128 for
// the GELU epilogue's z + a), and the epilogue is bound by the ISSUE of those narrow vector-memory instructions, not by
// HBM (de-phasing the CUs changed nothing; 128 KB per tile in ~6 us = 21 B/clk/CU).  Here a lane owns 8 consecutive
// columns of one row: 8 lanes cover a 64-column row (128 contiguous bytes), one instruction covers 8 rows, and a wave
// tile needs 16 stores per bf16 tensor.  Same LDS transpose; the read-back takes two swizzled 16-byte chunks per lane
// (conflict-free: the XOR keeps an aligned chunk pair together).  Used when C is bf16 with no residual / row remap /
// split-K slab and N, ldc, ldaux are multiples of 8.
typedef __attribute__((ext_vector_type(8))) float f32x8;
struct WideEpi {
  __amdgpu_buffer_rsrc_t rc, raux;
  bool has_aux, col_ok;
  uint32_t col;                  // first of this lane's 8 consecutive output columns
  f32x4 bias0, bias1, csum0, csum1;
  uint64_t drop_base;
  uint32_t row, c, aux, c_step, aux_step;   // the lane's current row, advanced by 8 rows per body
};
struct WideCursor { uint32_t row, aux; };
struct WideLoads { u32x4_t aux[2]; };

__device__ __forceinline__ void wide_issue_aux(const GemmParams& p, const WideEpi& e, WideCursor& lc, WideLoads& L) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const bool ok = e.col_ok && lc.row < (uint32_t)p.M;
    L.aux[b] = __builtin_amdgcn_raw_buffer_load_b128(e.raux, ok ? lc.aux : OOB, 0, XVIT_EPI_LOAD_AUX);
    lc.row += 8; lc.aux += e.aux_step;
  }
}

__device__ __forceinline__ u32x4_t pack_bf16x8(const f32x4& a, const f32x4& b) {
  const bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  return __builtin_bit_cast(u32x4_t, o);
}

// activation + dropout on 4 of the lane's 8 elements (columns col .. col + 3 of row `row`)
template <int ACT, bool DROP>
__device__ __forceinline__ void wide_half(const GemmParams& p, f32x4& v, const bf16x4 z, uint64_t idx, u32x2_t& dpack) {
  if (ACT == XVIT_ACT_GELU) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = gelu_f(v[c]);
  } else if (ACT == ACT_GELU_D) {   // gelu and, packed to bf16 at once (registers), its derivative for the aux tensor
    f32x4 d;
#pragma unroll
    for (int c = 0; c < 4; ++c) { float a_, d_; gelu_and_grad(v[c], a_, d_); v[c] = a_; d[c] = d_; }
    const bf16x4 db = {f2bf(d[0]), f2bf(d[1]), f2bf(d[2]), f2bf(d[3])};
    dpack = __builtin_bit_cast(u32x2_t, db);
  } else if (ACT == XVIT_ACT_DGELU) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] *= dgelu_f(bf2f(z[c]));
  } else if (ACT == ACT_MULAUX) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] *= bf2f(z[c]);
  }
  if (DROP) {
    const uint32_t thr = (uint32_t)(p.drop_p * 16777216.0f);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (hash32(drop_seed_at(p.drop_seed, p.drop_epoch), idx + c) & 0xFFFFFFu) >= thr ? v[c] * p.drop_inv : 0.f;
  }
}

template <int ACT, bool DROP>
__device__ __forceinline__ void wide_body(const GemmParams& p, WideEpi& e, f32x4 v0, f32x4 v1, u32x4_t auxv) {
  const bool ok = e.col_ok && e.row < (uint32_t)p.M;
  v0 += e.bias0; v1 += e.bias1;
  if (ACT == XVIT_ACT_GELU && e.has_aux) __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v0, v1), e.raux, ok ? e.aux : OOB, 0, XVIT_EPI_STORE_AUX);
  const bf16x8 z = __builtin_bit_cast(bf16x8, auxv);
  const uint64_t idx = e.drop_base + (uint64_t)e.row * p.N + e.col;
  // the two halves one after the other (a scheduling fence between them): eight interleaved GELU / hash evaluations
  // need more temporaries than the allocator has next to the accumulators still waiting for their LDS pass
  u32x2_t d0 = {0u, 0u}, d1 = {0u, 0u};
  wide_half<ACT, DROP>(p, v0, bf16x4{z[0], z[1], z[2], z[3]}, idx, d0);
  if ((ACT != XVIT_ACT_NONE && ACT != ACT_MULAUX) || DROP) __builtin_amdgcn_sched_barrier(0);
  wide_half<ACT, DROP>(p, v1, bf16x4{z[4], z[5], z[6], z[7]}, idx + 4, d1);
  if (ACT == ACT_GELU_D && e.has_aux) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{d0[0], d0[1], d1[0], d1[1]}, e.raux, ok ? e.aux : OOB, 0, XVIT_EPI_STORE_AUX);
  __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v0, v1), e.rc, ok ? e.c : OOB, 0, XVIT_EPI_STORE_AUX);
#pragma unroll
  for (int c = 0; c < 4; ++c) { e.csum0[c] += ok ? v0[c] : 0.f; e.csum1[c] += ok ? v1[c] : 0.f; }
  e.row += 8; e.c += e.c_step; e.aux += e.aux_step;
}

// Region R (0..7) = 2 bodies = rows 16 R .. 16 R + 15 of the wave tile (one scheduling window: with 4 bodies of 8 elements
// in one window the allocator spills); regions 0-3 and 4-7 share one LDS transpose pass.
template <int ACT, bool DROP, int R>
__device__ __forceinline__ void wide_regions(const GemmParams& p, WideEpi& e, WideCursor& lc, WideLoads& cur, const f32x4 (&acc)[8][4], XVIT_LDS char* slice,
                                             const uint32_t (&woff)[4], const uint32_t (&roffw)[2][2]) {
  if constexpr ((R & 3) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(XVIT_LDS f32x4*)(slice + i * 4096 + woff[j]) = acc[(R >> 2) * 4 + i][j];
  }
  WideLoads nxt;
  if constexpr ((ACT == XVIT_ACT_DGELU || ACT == ACT_MULAUX) && R < 7) wide_issue_aux(p, e, lc, nxt);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int it = (R & 3) * 2 + b;   // 8-row group inside the 64-row pass
    const f32x4 v0 = *(const XVIT_LDS f32x4*)(slice + it * 2048 + roffw[it & 1][0]);
    const f32x4 v1 = *(const XVIT_LDS f32x4*)(slice + it * 2048 + roffw[it & 1][1]);
    wide_body<ACT, DROP>(p, e, v0, v1, cur.aux[b]);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (R < 7) wide_regions<ACT, DROP, R + 1>(p, e, lc, nxt, acc, slice, woff, roffw);
}

template <int ACT, bool DROP>
__device__ __forceinline__ void wide_epilogue(const GemmParams& p, WideEpi& e, const f32x4 (&acc)[8][4], XVIT_LDS char* slice, const uint32_t (&woff)[4],
                                              const uint32_t (&roffw)[2][2], float* colsum_dst) {
  WideCursor lc = {e.row, e.aux};
  WideLoads first;
  if constexpr (ACT == XVIT_ACT_DGELU || ACT == ACT_MULAUX) wide_issue_aux(p, e, lc, first);
  wide_regions<ACT, DROP, 0>(p, e, lc, first, acc, slice, woff, roffw);
  if (colsum_dst) {   // lanes l, l+8, ..., l+56 share their 8 columns
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t0 = e.csum0[c], t1 = e.csum1[c];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); }
      e.csum0[c] = t0; e.csum1[c] = t1;
    }
    if ((threadIdx.x & 63) < 8 && e.col_ok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { unsafeAtomicAdd(colsum_dst + c, e.csum0[c]); unsafeAtomicAdd(colsum_dst + 4 + c, e.csum1[c]); }
    }
  }
}

template <int ACT>
__device__ __forceinline__ void wave_tile_epilogue_wide(const GemmParams& p, const f32x4 (&acc)[8][4], XVIT_LDS char* smem, int wave, int lane, int row0, int col0,
                                                        int batch) {
  WideEpi e;
  e.rc = make_rsrc((const char*)p.C + batch * p.sC * 2, clamp_bytes(0x7FFFFFF0ll));
  e.has_aux = p.aux != nullptr;
  const bool has_bias = p.bias != nullptr;
  e.raux = make_rsrc(e.has_aux ? (const void*)(p.aux + batch * p.sAux) : (const void*)p.C, e.has_aux ? 0x7FFFFFF0u : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(has_bias ? (const void*)(p.bias + batch * p.sBias) : (const void*)p.C, has_bias ? (uint32_t)(p.N * 4) : 0u);
  int pin = 0;                       // keep the address arithmetic below the K loop (see wave_tile_epilogue)
  asm volatile("" : "+v"(pin));
  const int wl = lane + pin;
  uint32_t woff[4], roffw[2][2];
  {
    const int r = wl & 15, g = wl >> 4, c = wl & 7, rr = wl >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = (uint32_t)(r * 256 + (((j * 4 + g) ^ r) << 4));
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int k = 0; k < 2; ++k) roffw[par][k] = (uint32_t)(rr * 256 + (((2 * c + k) ^ ((par * 8 + rr) & 15)) << 4));
  }
  XVIT_LDS char* slice = smem + wave * EPI_WAVE_BYTES;
  e.col = (uint32_t)(col0 + (wl & 7) * 8);
  e.col_ok = e.col < (uint32_t)p.N;
  e.bias0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, e.col * 4, 0, 0));
  e.bias1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, e.col * 4 + 16, 0, 0));
  e.drop_base = (uint64_t)batch * ((uint64_t)p.M * p.N);
  e.csum0 = f32x4{0.f, 0.f, 0.f, 0.f}; e.csum1 = f32x4{0.f, 0.f, 0.f, 0.f};
  float* colsum_dst = p.colsum ? p.colsum + batch * p.sBias + e.col : nullptr;
  e.row = (uint32_t)(row0 + (wl >> 3));
  e.c = e.row * (uint32_t)p.ldc * 2u + e.col * 2u;
  e.aux = e.row * (uint32_t)p.ldaux * 2u + e.col * 2u;
  e.c_step = 8u * (uint32_t)p.ldc * 2u;
  e.aux_step = 8u * (uint32_t)p.ldaux * 2u;
  wide_epilogue<ACT, false>(p, e, acc, slice, woff, roffw, colsum_dst);
}

// Epilogue of one wave's 128 x 64 accumulator tile whose first element is (row0, col0); shared by every tile kernel
// (the block's stage buffers are idle by now: wave w bounces through smem + w * EPI_WAVE_BYTES).
template <bool PERM = false>
__device__ __forceinline__ void wave_tile_epilogue(const GemmParams& p, const f32x4 (&acc)[8][4], XVIT_LDS char* smem, int wave, int lane, int row0, int col0,
                                                   int batch, int split) {
  const int nbatch = gridDim.z / p.split_k;
  BigEpi e;
  const int64_t celt = p.c_f32 ? 4 : 2;
  e.rc = make_rsrc((const char*)p.C + batch * p.sC * celt, clamp_bytes(0x7FFFFFF0ll));
  e.has_aux = p.aux != nullptr; e.has_res = p.res != nullptr; e.to_slab = p.slab != nullptr;
  const bool has_bias = p.bias != nullptr;
  e.raux = make_rsrc(e.has_aux ? (const void*)(p.aux + batch * p.sAux) : (const void*)p.C, e.has_aux ? 0x7FFFFFF0u : 0u);
  e.rres = make_rsrc(e.has_res ? (const void*)(p.res + batch * p.sR) : (const void*)p.C, e.has_res ? 0x7FFFFFF0u : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(has_bias ? (const void*)(p.bias + batch * p.sBias) : (const void*)p.C, has_bias ? (uint32_t)(p.N * 4) : 0u);
  e.rslab = make_rsrc(e.to_slab ? (const void*)(p.slab + ((int64_t)split * nbatch + batch) * (int64_t)p.M * p.N) : (const void*)p.C,
                      e.to_slab ? clamp_bytes((int64_t)p.M * p.N * 4) : 0u);
  // the epilogue's address arithmetic is loop-invariant: without this opaque dependency (placed AFTER the K loop)
  // LLVM hoists it above the MFMA loop and spills the accumulators
  int pin = 0;
  asm volatile("" : "+v"(pin));
  const int wl = lane + pin;
  // LDS slice offsets: accumulator layout (row r = lane & 15 of each 16-row tile, chunk 4 J + g) and read-back
  // layout (row 4 it + rr, chunk k), both with chunk ^= row & 15
  uint32_t woff[8], roff[4];
  {
    const int r = wl & 15, g = wl >> 4, k = wl & 15, rr = wl >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = woff[4 + j] = (uint32_t)(r * 256 + (((j * 4 + g) ^ r) << 4));
    if constexpr (PERM) {   // LDS row 8 jj + ii of a 32-row group holds tile row (ii % 4) 8 + 2 jj + ii / 4 (BigLoader::init_gather_rows)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int l32 = 16 * half + r, jj = l32 >> 3, ii = l32 & 7, rho = (ii & 3) * 8 + 2 * jj + (ii >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) woff[half * 4 + j] = (uint32_t)(rho * 256 + (((j * 4 + g) ^ (rho & 15)) << 4));
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) roff[q] = (uint32_t)(rr * 256 + ((k ^ (q * 4 + rr)) << 4));
  }
  XVIT_LDS char* slice = smem + wave * EPI_WAVE_BYTES;
  e.col = (uint32_t)(col0 + (wl & 15) * 4);
  e.col_ok = e.col < (uint32_t)p.N;
  e.bias = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, e.col * 4, 0, 0));   // zero-size descriptor when there is no bias
  e.drop_base = (uint64_t)batch * ((uint64_t)p.M * p.N);
  e.csum = f32x4{0.f, 0.f, 0.f, 0.f};
  e.colsum_dst = p.colsum ? p.colsum + batch * p.sBias + e.col : nullptr;
  e.row = (uint32_t)(row0 + (wl >> 4));
  {
    const bool rmap = p.res_row_mod > 0, smap = p.seg_rows > 0;
    e.rmod = rmap ? (uint32_t)p.res_row_mod : 0x7FFFFFFFu;
    e.smod = smap ? (uint32_t)p.seg_rows : 0x7FFFFFFFu;
    e.rrem = rmap ? e.row % e.rmod : e.row;
    e.srem = smap ? e.row % e.smod : e.row;
    const uint32_t rrow = rmap ? (uint32_t)p.res_row_off + e.rrem : e.row;
    const uint32_t orow = smap ? e.row + (e.row / e.smod) * (uint32_t)p.seg_skip + (uint32_t)p.row_off : e.row;
    const uint32_t ce = p.c_f32 ? 4u : 2u;
    e.c = orow * (uint32_t)p.ldc * ce + e.col * ce;
    e.aux = e.row * (uint32_t)p.ldaux * 2u + e.col * 2u;
    e.res = rrow * (uint32_t)p.ldr * 4u + e.col * 4u;
    e.slab = e.row * (uint32_t)p.N * 4u + e.col * 4u;
    e.c_step = 4u * (uint32_t)p.ldc * ce;
    e.aux_step = 4u * (uint32_t)p.ldaux * 2u;
    e.res_step = 4u * (uint32_t)p.ldr * 4u;
    e.slab_step = 4u * (uint32_t)p.N * 4u;
    e.c_gap = smap ? (uint32_t)p.seg_skip * (uint32_t)p.ldc * ce : 0u;
    e.res_wrap = rmap ? e.rmod * (uint32_t)p.ldr * 4u : 0u;
  }
  // one specialised, fully unrolled copy per (activation, dropout): every acc[][] index is a compile-time constant
  if constexpr (PERM) {   // the patch-embedding forward: bias + positional residual only
    big_epilogue<XVIT_ACT_NONE, false, true>(p, e, acc, slice, woff, roff);
    return;
  }
  if (p.drop_p > 0.f) {
    if (p.act == XVIT_ACT_GELU) big_epilogue<XVIT_ACT_GELU, true>(p, e, acc, slice, woff, roff);
    else if (p.act == XVIT_ACT_DGELU) big_epilogue<XVIT_ACT_DGELU, true>(p, e, acc, slice, woff, roff);
    else big_epilogue<XVIT_ACT_NONE, true>(p, e, acc, slice, woff, roff);
  } else if (p.act == XVIT_ACT_GELU) big_epilogue<XVIT_ACT_GELU, false>(p, e, acc, slice, woff, roff);
  else if (p.act == XVIT_ACT_DGELU) big_epilogue<XVIT_ACT_DGELU, false>(p, e, acc, slice, woff, roff);
  else big_epilogue<XVIT_ACT_NONE, false>(p, e, acc, slice, woff, roff);
}

// WIDE_ACT >= 0: the 16-byte-per-lane bf16 epilogue with that activation and no dropout, as its own instantiation (with
// several fully unrolled epilogue variants behind one K loop the register allocator spills: one variant per kernel here;
// the narrow kernels keep the run-time switch over activation x dropout and do not spill).  WIDE_ACT = -1: narrow.
#ifdef XVIT_GEMM_CLOCK_PROBE
// Diagnostic build (tools/gemm_kstep_probe.py): workgroup b < 4096 leaves its lifetime in shader cycles (clock64) and in
// ticks of the constant 100 MHz counter (wall_clock64); their ratio is the clock the CU ran at under this kernel's load.
__device__ long long xvit_dbg_clk[2 * 4096];
struct ClockProbe {
  long long c0, w0;
  __device__ ClockProbe() : c0(clock64()), w0(wall_clock64()) {}
  __device__ ~ClockProbe() {
    if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.x < 4096) { xvit_dbg_clk[2 * blockIdx.x] = clock64() - c0; xvit_dbg_clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
  }
};
#endif
template <bool A_KS, bool B_KS, int WIDE_ACT, int GATHER = 0>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  XVIT_LDS char* smem = (XVIT_LDS char*)smem_raw;
#ifdef XVIT_GEMM_CLOCK_PROBE
  ClockProbe clock_probe;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;   // 2 x 4 waves

  // XCD-aware bijective remap over the whole (tile, batch x split) grid: workgroups are dealt to the 8 XCDs
  // round-robin in dispatch order (x fastest, then z), so logical ids [xcd * total/8, (xcd+1) * total/8) run on
  // one XCD.  Consecutive logical ids walk the tiles of ONE K-split (sharing its A and B panels through that
  // XCD's L2) before moving to the next split; with the tile index alone (36 tiles, not a multiple of 8) the
  // split-K wgrads fetched 2.3x their algorithmic bytes.
  const int nblk = gridDim.x, total = nblk * gridDim.z, lin = blockIdx.x + nblk * blockIdx.z;
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  const int zz = logical / nblk, tile = logical - zz * nblk;
  // Within one (batch, split) the tiles are walked super-column by super-column: p.ncg column tiles wide, row tiles outermost
  // inside it, columns fastest.  The ~32 tiles an XCD runs at one time then share a few A row-panels AND a W slice narrow
  // enough to stay in its 4 MiB L2 across row-panels (with all N / 256 column tiles fastest, a 3072-wide Linear re-streams
  // its whole 4.7 MB weight matrix for every 2-3 row-panels: 689 MB fetched per FFN1 launch against 104 MB of operands).
  const int grp = min(tile / (p.ntm * p.ncg), (p.ntn - 1) / p.ncg);
  const int rem = tile - grp * p.ntm * p.ncg;
  const int gw = min(p.ncg, p.ntn - grp * p.ncg);
  const int tm = rem / gw, tn = grp * p.ncg + (rem - tm * gw);
  const int m0 = tm * TBM, n0 = tn * TBN;
  const int batch = zz / p.split_k, split = zz - batch * p.split_k;
  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = (k_end - k_begin + BK - 1) / BK;   // may be 0 for a trailing split: still writes its (zero) slab

  BigLoader<A_KS> la;
  BigLoader<B_KS> lb;
  {
    const bf16* Ab = p.A + batch * p.sA;
    if constexpr (GATHER == 1) la.init_gather_rows(p.g, Ab, m0, p.M, wave, lane);
    else if constexpr (GATHER == 2 || GATHER == 3) {   // dY rows are re-indexed per K-step (CLS rows skipped): base = row 0, K-step offsets from gather_soff_wgrad
      const int64_t rows = (int64_t)(p.K / p.g.pcount) * p.g.ntok;
      la.init(Ab + m0, ((rows - 1) * p.lda + (p.M - m0)) * 2, p.lda, wave, lane, GATHER == 2 ? &p.g : nullptr);
    } else if (A_KS) la.init(Ab + (int64_t)k_begin * p.lda + m0, ((int64_t)(k_end - 1 - k_begin) * p.lda + (p.M - m0)) * 2, p.lda, wave, lane);
    else la.init(Ab + (int64_t)m0 * p.lda + k_begin, ((int64_t)(p.M - 1 - m0) * p.lda + (k_end - k_begin)) * 2, p.lda, wave, lane);
    const bf16* Bb = p.B + batch * p.sB;
    if constexpr (GATHER == 2 || GATHER == 3) lb.init_gather_ks(p.g, Bb, n0, p.N, wave, lane);
    else if (B_KS) lb.init(Bb + (int64_t)k_begin * p.ldb + n0, ((int64_t)(k_end - 1 - k_begin) * p.ldb + (p.N - n0)) * 2, p.ldb, wave, lane);
    else lb.init(Bb + (int64_t)n0 * p.ldb + k_begin, ((int64_t)(p.N - 1 - n0) * p.ldb + (k_end - k_begin)) * 2, p.ldb, wave, lane);
  }
  // byte offsets of K-step kt (relative to this split's first) for the two loaders
  const int ktg0 = k_begin / BK;
  uint32_t fix_a[4], fix_b[4];       // mode 3: the lanes' fixed parts (column chunk of dx, feature chunk of the patch); the token part is placed per K-step
  if constexpr (GATHER == 3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int krow = (wave * 4 + j) * 2 + (lane >> 5);
      const int chunk = (lane & 31) ^ swz_ks(krow);
      fix_a[j] = m0 + chunk * 8 < p.M ? (uint32_t)(chunk * 16) : GATHER_OOB;
      const int e = n0 + (chunk << 3);
      fix_b[j] = e < p.N ? 2u * gather_elem_off(p.g, e) : GATHER_OOB;
    }
  }
  auto step_off = [&](int kt, uint32_t& oa, uint32_t& ob) {
    if constexpr (GATHER == 1) { oa = gather_soff_rows(p.g, ktg0 + kt); ob = (uint32_t)kt * lb.kstep; }
    else if constexpr (GATHER == 2) { GatherStep o; o = gather_soff_wgrad(p.g, ktg0 + kt, p.lda); oa = o.a; ob = o.b; }
    else if constexpr (GATHER == 3) {
      la.place_tokens(p.g, ktg0 + kt, (uint32_t)p.K, p.lda, false, fix_a, wave, lane);
      lb.place_tokens(p.g, ktg0 + kt, (uint32_t)p.K, 0, true, fix_b, wave, lane);
      oa = ob = 0u;
    }
    else { oa = (uint32_t)kt * la.kstep; ob = (uint32_t)kt * lb.kstep; }
  };
  BigFrag<A_KS, 8> fa;
  BigFrag<B_KS, 4> fb;
  fa.init(wr * 8, lane);
  fb.init(wc * 4, lane);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t null_rsrc = make_rsrc(p.A, 0u);   // every access out of range: the DMA fills zeros, fetches nothing
  if (nk > 0) {
    uint32_t oa, ob;
    step_off(0, oa, ob);
    la.issue_at(smem, wave, oa);
    lb.issue_at(smem + T_OPER, wave, ob);
  }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = kt + 1 < nk;
    uint32_t soff_a, soff_b;
    step_off(kt + 1, soff_a, soff_b);
    XVIT_LDS char* nxt = smem + ((kt + 1) & 1) * T_STAGE;
    const XVIT_LDS char* sa = smem + (kt & 1) * T_STAGE;
    const XVIT_LDS char* sb = sa + T_OPER;
#ifndef XVIT_DEBUG_NO_MMA
    big_tile_mma<A_KS, B_KS>(sa, sb, fa, fb, acc, la, lb, more ? la.rsrc : null_rsrc, more ? lb.rsrc : null_rsrc, nxt, wave, soff_a, soff_b);
#endif
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last iteration's (zero-fill) DMAs have landed before the stage buffers are recycled
  __syncthreads();   // every wave is done reading the last stage: the stage buffers become the transpose slices
  if constexpr (WIDE_ACT >= 0) wave_tile_epilogue_wide<WIDE_ACT>(p, acc, smem, wave, lane, m0 + wr * 128, n0 + wc * 64, batch);
  else wave_tile_epilogue<GATHER == 1>(p, acc, smem, wave, lane, m0 + wr * 128, n0 + wc * 64, batch, split);
}

// split-K second pass: sum the partial tiles in a fixed order (bit-reproducible), then the full epilogue
__global__ void splitk_epilogue_kernel(const GemmParams p, int nbatch) {
  const int nv = p.N >> 2;
  const int64_t per = (int64_t)p.M * nv, total = per * nbatch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const int64_t r = i - (int64_t)b * per;
    const int row = (int)(r / nv), col = (int)(r - (int64_t)row * nv) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.split_k; ++s) v += *(const f32x4*)(p.slab + (((int64_t)s * nbatch + b) * p.M + row) * p.N + col);
    const f32x4 o = epilogue_apply_rt(p, v, row, col, b * p.sC, p.bias ? p.bias + b * p.sBias : nullptr, p.res ? p.res + b * p.sR : nullptr,
                                      p.aux ? p.aux + b * p.sAux : nullptr);
    if (p.colsum) {
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(p.colsum + b * p.sBias + col + e, o[e]);
    }
  }
}

}  // namespace xvit

using namespace xvit;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 256x256 tiles (one block per CU, 128 FLOP per staged byte) unless their grid would occupy at most half of the
// 256 CUs: then the 128x128 kernel (4x the blocks, two per CU) finishes sooner despite its lower intensity
// (e.g. the reference's batch 8: M = 4104 rows -> 17 x 3 big tiles for a d-wide GEMM).
static std::atomic<int> g_gemm_tile{0};      // xvit_set_option("gemm_tile"): 0 = auto, 1 = always the 128x128 kernel
static std::atomic<int> g_gemm_epi{0};       // xvit_set_option("gemm_epilogue"): 0 = auto, 1 = always the 8-byte-per-lane epilogue
static std::atomic<int> g_gemm_group{0};     // xvit_set_option("gemm_group"): 0 = auto, n > 0 = column tiles per super-column of the 256x256 tile walk

static bool use_big_tile(const xvit_gemm_args* a) {
  if (a->M < 256 || a->N < 256 || g_gemm_tile.load(std::memory_order_relaxed) == 1) return false;
  if (g_gemm_tile.load(std::memory_order_relaxed) == 2) return true;   // diagnostics: the 256x256 kernel on grids of any size
  const int64_t big_blocks = (int64_t)((a->M + 255) / 256) * ((a->N + 255) / 256) * a->batch * (a->split_k > 0 ? a->split_k : 1);
  return big_blocks > 128;
}

extern "C" int xvit_set_option(const char* name, int value) {
  XVIT_REQUIRE(name != nullptr, "xvit_set_option: null name");
  const std::string n(name);
  if (n == "gemm_tile") { XVIT_REQUIRE(value >= 0 && value <= 2, "xvit_set_option: gemm_tile must be 0 (auto), 1 (128x128 only) or 2 (256x256 whenever M, N >= 256)"); g_gemm_tile = value; return 0; }
  if (n == "gemm_group") { XVIT_REQUIRE(value >= 0 && value <= 4096, "xvit_set_option: gemm_group must be in [0, 4096]"); g_gemm_group = value; return 0; }
  if (n == "gemm_epilogue") { XVIT_REQUIRE(value == 0 || value == 1, "xvit_set_option: gemm_epilogue must be 0 (auto) or 1 (narrow)"); g_gemm_epi = value; return 0; }
  if (n == "attn_peel") { XVIT_REQUIRE(value >= 0 && value <= 2, "xvit_set_option: attn_peel must be 0 (token 0 stays on the tile grid), 1 (auto: large grids) or 2 (whenever N = 64 m + 1)"); set_attn_peel(value); return 0; }
  set_error("xvit_set_option: unknown option '%s'", name);
  return XVIT_ERR_ARG;
}

extern "C" int64_t xvit_gemm_workspace_bytes(const xvit_gemm_args* a) {
  if (!a || a->split_k <= 1) return 0;
  return (int64_t)a->split_k * a->batch * a->M * a->N * (int64_t)sizeof(float);
}

extern "C" int xvit_gemm(const xvit_gemm_args* a, xvit_stream_t stream) {
  XVIT_REQUIRE(a != nullptr, "xvit_gemm: null args");
  XVIT_REQUIRE(a->layout >= 0 && a->layout <= 2, "xvit_gemm: bad layout %d", a->layout);
  XVIT_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0, "xvit_gemm: M,N,K,batch must be > 0 (got %d,%d,%d,%d)", a->M, a->N, a->K, a->batch);
  XVIT_REQUIRE(a->A && a->B && a->C, "xvit_gemm: null A/B/C");
  const bool a_ks = a->layout == XVIT_GEMM_TN, b_ks = a->layout != XVIT_GEMM_NT;
  XVIT_REQUIRE(a_ks || a->K % 64 == 0, "xvit_gemm: K=%d must be a multiple of 64 when A is k-contiguous", a->K);
  XVIT_REQUIRE(b_ks || a->K % 64 == 0, "xvit_gemm: K=%d must be a multiple of 64 when B is k-contiguous", a->K);
  XVIT_REQUIRE(a->N % 4 == 0, "xvit_gemm: N=%d must be a multiple of 4 (use xvit_small_linear_*)", a->N);
  XVIT_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->stride_a % 8 == 0 && a->stride_b % 8 == 0, "xvit_gemm: lda/ldb/strides must be multiples of 8 elements");
  XVIT_REQUIRE(aligned16(a->A) && aligned16(a->B) && aligned16(a->C), "xvit_gemm: A/B/C must be 16-byte aligned");
  XVIT_REQUIRE(a->ldc % 4 == 0 && a->stride_c % 4 == 0, "xvit_gemm: ldc/stride_c must be multiples of 4");
  XVIT_REQUIRE(a->lda >= (a_ks ? a->M : a->K) && a->ldb >= (b_ks ? a->N : a->K) && a->ldc >= a->N, "xvit_gemm: leading dimension smaller than the row length");
  const int64_t a_rows = a_ks ? a->K : a->M, b_rows = b_ks ? a->K : a->N;
  XVIT_REQUIRE(a_rows * a->lda * 2 < (1ll << 31) && b_rows * a->ldb * 2 < (1ll << 31), "xvit_gemm: an operand matrix exceeds 2 GiB (unsupported addressing range)");
  XVIT_REQUIRE(a->split_k >= 1, "xvit_gemm: split_k must be >= 1");
  {
    const int64_t out_rows = a->out_seg_rows > 0 ? (int64_t)a->M + (a->M / a->out_seg_rows + 1) * a->out_seg_skip + a->out_row_off : a->M;
    XVIT_REQUIRE(out_rows * a->ldc * 4 < (1ll << 31) && (!a->aux || (int64_t)a->M * a->ldaux * 2 < (1ll << 31)) &&
                     (!a->residual || (int64_t)a->M * a->ldr * 4 < (1ll << 31)),
                 "xvit_gemm: C / aux / residual of one batch exceed 2 GiB (unsupported addressing range)");
  }
  XVIT_REQUIRE(!(a->accumulate && a->c_dtype != XVIT_F32), "xvit_gemm: accumulate needs fp32 C");
  XVIT_REQUIRE(a->act != XVIT_ACT_DGELU || a->aux, "xvit_gemm: ACT_DGELU needs aux (pre-activation)");
  XVIT_REQUIRE(a->aux_mode == 0 || a->aux_mode == 1, "xvit_gemm: aux_mode must be 0 (aux = pre-activation) or 1 (aux = GELU')");
  if (a->aux) XVIT_REQUIRE(a->ldaux % 4 == 0 && a->ldaux >= a->N && (reinterpret_cast<uintptr_t>(a->aux) & 7) == 0, "xvit_gemm: bad aux layout");
  if (a->residual) XVIT_REQUIRE(a->ldr % 4 == 0 && a->ldr >= a->N && aligned16(a->residual), "xvit_gemm: bad residual layout");
  if (a->bias) XVIT_REQUIRE(aligned16(a->bias) && a->stride_bias % 4 == 0, "xvit_gemm: bias must be 16-byte aligned");
  XVIT_REQUIRE((int64_t)a->batch * a->split_k <= 65535, "xvit_gemm: batch*split_k too large");
  const bool big = use_big_tile(a);
  const int64_t ws_need = xvit_gemm_workspace_bytes(a);
  if (ws_need > 0)
    XVIT_REQUIRE(a->workspace && a->workspace_bytes >= ws_need && aligned16(a->workspace), "xvit_gemm: split_k=%d needs %lld bytes of 16-byte aligned workspace (got %lld)",
                 a->split_k, (long long)ws_need, (long long)a->workspace_bytes);

  GemmParams p;
  p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C; p.bias = a->bias; p.res = a->residual; p.aux = (bf16*)a->aux;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
  p.sA = a->stride_a; p.sB = a->stride_b; p.sC = a->stride_c; p.sBias = a->stride_bias; p.sR = a->stride_r; p.sAux = a->stride_aux;
  p.M = a->M; p.N = a->N; p.K = a->K; p.split_k = a->split_k;
  const int ktiles = (a->K + BK - 1) / BK;
  p.k_per_split = ((ktiles + a->split_k - 1) / a->split_k) * BK;
  p.c_f32 = a->c_dtype == XVIT_F32; p.act = a->act; p.accumulate = a->accumulate;
  p.res_row_mod = a->res_row_mod; p.res_row_off = a->res_row_off;
  p.seg_rows = a->out_seg_rows; p.seg_skip = a->out_seg_skip; p.row_off = a->out_row_off;
  p.slab = ws_need > 0 ? (float*)a->workspace : nullptr;
  p.colsum = a->colsum;
  XVIT_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "xvit_gemm: dropout_p must be in [0, 1)");
  p.drop_p = a->dropout_p; p.drop_inv = 1.0f / (1.0f - a->dropout_p); p.drop_seed = a->dropout_seed;
  p.drop_epoch = a->dropout_p > 0.f ? drop_epoch_ptr() : nullptr;
  p.narrow_epi = g_gemm_epi.load(std::memory_order_relaxed);
  p.aux_deriv = a->aux_mode;
  p.ncg = 1;
  p.g.mode = 0;
  hipStream_t s = (hipStream_t)stream;

  static std::once_flag attr_once;   // the library is re-entrant: concurrent first calls from several host threads
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, true, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<true, true, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, XVIT_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, XVIT_ACT_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, true, XVIT_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, true, XVIT_ACT_DGELU>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, ACT_GELU_D>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, true, ACT_MULAUX>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
  });
  if (big) {
    p.ntm = (a->M + TBM - 1) / TBM; p.ntn = (a->N + TBN - 1) / TBN;
    {
      // measured at M = 64638, K = 768 (tools/gemm_model_bench.py): forward Linears with >= 6 column tiles gain 4-14 % from
      // super-columns of 3 (qkv 238 -> 204 us, kv 152 -> 141, FFN1 406 -> 391); dgrads / wgrads and narrow outputs do not
      const int forced = g_gemm_group.load(std::memory_order_relaxed);
      p.ncg = forced > 0 ? std::min(forced, p.ntn) : (a->layout == XVIT_GEMM_NT && p.ntn >= 6 ? 3 : p.ntn);
    }
    const dim3 grid(p.ntm * p.ntn, 1, a->batch * a->split_k), block(512);
    // bf16 C with nothing but bias / activation / column sums behind it: the 16-byte-per-lane epilogue (forward Linears use
    // NONE or GELU, dgrads NONE or GELU'; wgrads store fp32; with dropout the narrow kernels run)
    const bool wide = !p.c_f32 && !p.slab && !p.res && p.seg_rows == 0 && !p.narrow_epi && p.drop_p == 0.f && a->layout != XVIT_GEMM_TN &&
                      ((p.N | (int)p.ldc | (int)p.ldaux) & 7) == 0 && (a->layout == XVIT_GEMM_NT ? p.act != XVIT_ACT_DGELU : p.act != XVIT_ACT_GELU);
    if (wide && a->layout == XVIT_GEMM_NT && p.act == XVIT_ACT_NONE) hipLaunchKernelGGL((gemm_big_kernel<false, false, XVIT_ACT_NONE>), grid, block, T_LDS, s, p);
    else if (wide && a->layout == XVIT_GEMM_NT && p.aux_deriv) hipLaunchKernelGGL((gemm_big_kernel<false, false, ACT_GELU_D>), grid, block, T_LDS, s, p);
    else if (wide && a->layout == XVIT_GEMM_NT) hipLaunchKernelGGL((gemm_big_kernel<false, false, XVIT_ACT_GELU>), grid, block, T_LDS, s, p);
    else if (wide && p.act == XVIT_ACT_NONE) hipLaunchKernelGGL((gemm_big_kernel<false, true, XVIT_ACT_NONE>), grid, block, T_LDS, s, p);
    else if (wide && p.aux_deriv) hipLaunchKernelGGL((gemm_big_kernel<false, true, ACT_MULAUX>), grid, block, T_LDS, s, p);
    else if (wide) hipLaunchKernelGGL((gemm_big_kernel<false, true, XVIT_ACT_DGELU>), grid, block, T_LDS, s, p);
    else if (a->layout == XVIT_GEMM_NT) hipLaunchKernelGGL((gemm_big_kernel<false, false, -1>), grid, block, T_LDS, s, p);
    else if (a->layout == XVIT_GEMM_NN) hipLaunchKernelGGL((gemm_big_kernel<false, true, -1>), grid, block, T_LDS, s, p);
    else hipLaunchKernelGGL((gemm_big_kernel<true, true, -1>), grid, block, T_LDS, s, p);
  } else {
    p.ntm = (a->M + BM - 1) / BM; p.ntn = (a->N + BN - 1) / BN;
    const dim3 grid(p.ntm * p.ntn, 1, a->batch * a->split_k), block(256);
    switch (a->layout) {
      case XVIT_GEMM_NT: hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, GEMM_LDS, s, p); break;
      case XVIT_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, GEMM_LDS, s, p); break;
      default: hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, GEMM_LDS, s, p); break;
    }
  }
  if (p.slab) {
    const int64_t work = (int64_t)a->batch * a->M * (a->N / 4);
    const int g = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(g), dim3(256), 0, s, p, a->batch);
  }
  return check_launch("xvit_gemm");
}

// ------------------------------------------------------------------------------------------------------------------
// Patch embedding straight from the volume (reference model_cross.py:193-197): the [rows, dp hp wp] patch matrix of the
// einops rearrange is never written — gemm_big_kernel's LDS-DMA loaders gather the 16-byte chunks of every patch row from
// the [B, M, 1, D, H, W] bf16 tensor (a 64-deep K-step of one patch row = 64 / wp runs of wp contiguous voxels).
// ------------------------------------------------------------------------------------------------------------------
static int64_t pe_rows(const xvit_patch_geom* g) {
  return (int64_t)g->M * g->B * (g->cls_rows + (int64_t)(g->D / g->dp) * (g->H / g->hp) * (g->W / g->wp));
}

extern "C" int xvit_patch_embed_supported(const xvit_patch_geom* g, int d) {
  if (!g || g->B <= 0 || g->M <= 0 || g->dp <= 0 || g->hp <= 0 || g->wp <= 0 || (g->cls_rows != 0 && g->cls_rows != 1)) return 0;
  if (g->D % g->dp || g->H % g->hp || g->W % g->wp) return 0;
  const int Dn = g->D / g->dp, Hn = g->H / g->hp, Wn = g->W / g->wp;
  const int64_t pd = (int64_t)g->dp * g->hp * g->wp, pcount = (int64_t)Dn * Hn * Wn;
  if (g->wp % 8 || 64 % g->wp || (g->hp * g->wp) % 64) return 0;          // a K-step = whole runs of one (p1) slab
  // (any patch grid: where 64 consecutive tokens are not whole d-columns of one h — 64 % Dn, (Dn Wn) % 64 — the weight gradient
  // places its k-rows per K-step, pe_aligned() below)
  if (pd % 256 || d % 256 || d < 256) return 0;                            // full 256-wide tiles on both outputs
  const int64_t rows = pe_rows(g);
  if (rows < 2048) return 0;                                               // small problems: the 128x128 kernels on a stored patch matrix
  if ((int64_t)g->B * g->M * g->D * g->H * g->W * 2 >= (1ll << 31)) return 0;
  if (rows * d * 4 >= (1ll << 31) || pcount > (1 << 24)) return 0;
  return 1;
}

// 64 consecutive patch tokens of a sample are whole d-columns of one h-row: the weight gradient's K-step offsets are wave-uniform (mode 2)
static bool pe_aligned(const xvit_patch_geom* g) {
  const int Dn = g->D / g->dp, Wn = g->W / g->wp;
  return 64 % Dn == 0 && (Dn * Wn) % 64 == 0;
}

static void pe_fill(GemmParams& p, const xvit_patch_geom* g, int mode) {
  p.g.mode = mode;
  p.g.dp = g->dp; p.g.hp = g->hp; p.g.wp = g->wp;
  p.g.Dn = g->D / g->dp; p.g.Wn = g->W / g->wp;
  p.g.Sy = g->W; p.g.Sz = g->H * g->W;
  p.g.pcount = (g->D / g->dp) * (g->H / g->hp) * (g->W / g->wp);
  p.g.cls = g->cls_rows; p.g.ntok = p.g.cls + p.g.pcount;
  p.g.nb = g->B;
  p.g.sMd = (int64_t)g->D * g->H * g->W; p.g.sBt = p.g.sMd * g->M;
  p.g.vol_bytes = (uint32_t)((int64_t)g->B * g->M * g->D * g->H * g->W * 2);
  auto magic = [](int d) { return d <= 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / (uint64_t)d); };
  p.g.m_pcount = magic(p.g.pcount); p.g.m_dn = magic(p.g.Dn); p.g.m_wn = magic(p.g.Wn); p.g.m_nb = magic(p.g.nb);
}

static void pe_defaults(GemmParams& p) {
  p.bias = nullptr; p.res = nullptr; p.aux = nullptr; p.slab = nullptr; p.colsum = nullptr;
  p.lda = p.ldb = p.ldc = p.ldr = p.ldaux = 0;
  p.sA = p.sB = p.sC = p.sBias = p.sR = p.sAux = 0;
  p.c_f32 = 1; p.act = XVIT_ACT_NONE; p.accumulate = 0;
  p.res_row_mod = 0; p.res_row_off = 0; p.seg_rows = 0; p.seg_skip = 0; p.row_off = 0;
  p.drop_p = 0.f; p.drop_inv = 1.f; p.drop_seed = 0; p.drop_epoch = nullptr; p.narrow_epi = 1; p.split_k = 1; p.aux_deriv = 0;
}

static void pe_attrs() {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, -1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<true, true, -1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_big_kernel<true, true, -1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
  });
}

extern "C" int xvit_patch_embed_fwd(const void* img, const xvit_patch_geom* g, const void* W, int64_t ldw, const float* bias, const float* pos, int64_t ldpos,
                                    float* x, int64_t ldx, int d, xvit_stream_t stream) {
  XVIT_REQUIRE(img && g && W && x, "xvit_patch_embed_fwd: null pointer");
  XVIT_REQUIRE(xvit_patch_embed_supported(g, d) == 1, "xvit_patch_embed_fwd: geometry not supported by the fused kernel (xvit_patch_embed_supported)");
  const int64_t pd = (int64_t)g->dp * g->hp * g->wp;
  XVIT_REQUIRE(ldw >= pd && ldw % 8 == 0 && ldx >= d && ldx % 4 == 0 && (!pos || (ldpos >= d && ldpos % 4 == 0)), "xvit_patch_embed_fwd: bad leading dimension");
  XVIT_REQUIRE(aligned16(img) && aligned16(W) && aligned16(x) && (!bias || aligned16(bias)) && (!pos || aligned16(pos)), "xvit_patch_embed_fwd: pointers must be 16-byte aligned");
  GemmParams p;
  pe_defaults(p);
  pe_fill(p, g, 1);
  p.A = (const bf16*)img; p.B = (const bf16*)W; p.C = x; p.bias = bias; p.res = pos;
  p.ldb = ldw; p.ldc = ldx; p.ldr = ldpos;
  p.M = (int)pe_rows(g); p.N = d; p.K = (int)pd;
  p.k_per_split = p.K;
  if (pos) { p.res_row_mod = p.g.ntok; p.res_row_off = 0; }   // x row (s, n) += pos[n]  (model_cross.py:197)
  p.ntm = (p.M + TBM - 1) / TBM; p.ntn = (p.N + TBN - 1) / TBN; p.ncg = p.ntn;
  pe_attrs();
  hipLaunchKernelGGL((gemm_big_kernel<false, false, -1, 1>), dim3(p.ntm * p.ntn, 1, 1), dim3(512), T_LDS, (hipStream_t)stream, p);
  return check_launch("xvit_patch_embed_fwd");
}

static int pe_wgrad_split(const xvit_patch_geom* g, int d) {
  const int64_t pd = (int64_t)g->dp * g->hp * g->wp;
  const int tiles = (int)((d / 256) * (pd / 256));
  const int64_t ksteps = pe_rows(g) / 64;   // ~ contraction length / 64
  int split = 256 / (tiles > 0 ? tiles : 1);
  if (split > ksteps / 8) split = (int)(ksteps / 8);
  if (split > 16) split = 16;
  return split < 1 ? 1 : split;
}

extern "C" int64_t xvit_patch_embed_wgrad_workspace_bytes(const xvit_patch_geom* g, int d) {
  if (!g || xvit_patch_embed_supported(g, d) != 1) return 0;
  const int split = pe_wgrad_split(g, d);
  return split > 1 ? (int64_t)split * d * g->dp * g->hp * g->wp * (int64_t)sizeof(float) : 0;
}

extern "C" int xvit_patch_embed_wgrad(const void* img, const xvit_patch_geom* g, const void* dx, int64_t lddx, float* dW, int64_t lddw, int d, void* workspace,
                                      int64_t workspace_bytes, xvit_stream_t stream) {
  XVIT_REQUIRE(img && g && dx && dW, "xvit_patch_embed_wgrad: null pointer");
  XVIT_REQUIRE(xvit_patch_embed_supported(g, d) == 1, "xvit_patch_embed_wgrad: geometry not supported by the fused kernel (xvit_patch_embed_supported)");
  const int64_t pd = (int64_t)g->dp * g->hp * g->wp;
  XVIT_REQUIRE(lddx >= d && lddx % 8 == 0 && lddw >= pd && lddw % 4 == 0, "xvit_patch_embed_wgrad: bad leading dimension");
  XVIT_REQUIRE(aligned16(img) && aligned16(dx) && aligned16(dW), "xvit_patch_embed_wgrad: pointers must be 16-byte aligned");
  XVIT_REQUIRE(pe_rows(g) * lddx * 2 < (1ll << 31), "xvit_patch_embed_wgrad: dx exceeds 2 GiB (unsupported addressing range)");
  const int64_t need = xvit_patch_embed_wgrad_workspace_bytes(g, d);
  XVIT_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && aligned16(workspace)), "xvit_patch_embed_wgrad: needs %lld bytes of 16-byte aligned workspace (got %lld)",
               (long long)need, (long long)workspace_bytes);
  GemmParams p;
  pe_defaults(p);
  const bool aligned = pe_aligned(g);
  pe_fill(p, g, aligned ? 2 : 3);
  p.A = (const bf16*)dx; p.B = (const bf16*)img; p.C = dW;
  p.lda = lddx; p.ldc = lddw;
  p.M = d; p.N = (int)pd;
  p.K = (int)((int64_t)g->M * g->B * p.g.pcount);          // contraction over the patch rows; CLS rows carry no patch
  p.split_k = pe_wgrad_split(g, d);
  const int ktiles = (p.K + BK - 1) / BK;                     // mode 3: K need not be a multiple of 64 (tokens past it are zero rows)
  p.k_per_split = ((ktiles + p.split_k - 1) / p.split_k) * BK;
  p.slab = need > 0 ? (float*)workspace : nullptr;
  p.ntm = (p.M + TBM - 1) / TBM; p.ntn = (p.N + TBN - 1) / TBN; p.ncg = p.ntn;
  pe_attrs();
  hipStream_t s = (hipStream_t)stream;
  if (aligned) hipLaunchKernelGGL((gemm_big_kernel<true, true, -1, 2>), dim3(p.ntm * p.ntn, 1, p.split_k), dim3(512), T_LDS, s, p);
  else hipLaunchKernelGGL((gemm_big_kernel<true, true, -1, 3>), dim3(p.ntm * p.ntn, 1, p.split_k), dim3(512), T_LDS, s, p);
  if (p.slab) {
    const int64_t work = (int64_t)p.M * (p.N / 4);
    const int gsz = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(gsz), dim3(256), 0, s, p, 1);
  }
  return check_launch("xvit_patch_embed_wgrad");
}

#ifdef XVIT_GEMM_CLOCK_PROBE
extern "C" int xvit_dbg_read_clk(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xvit::xvit_dbg_clk), (size_t)n * 16); }
#endif
