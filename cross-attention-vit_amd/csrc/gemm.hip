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

namespace xvit {

struct GemmParams {
  const bf16* A; const bf16* B; void* C; const float* bias; const float* res; bf16* aux;
  int64_t lda, ldb, ldc, ldr, ldaux;
  int64_t sA, sB, sC, sBias, sR, sAux;
  int M, N, K, k_per_split, split_k, ntm, ntn;
  int c_f32, act, accumulate;
  int res_row_mod, res_row_off, seg_rows, seg_skip, row_off;
};

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
  if (nk <= 0) return;

  OperandLoader<A_KS> la;
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

  la.issue(smem, wave, 0);
  lb.issue(smem + OPER_BYTES, wave, 0);

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
  const bool atomic = p.split_k > 1;

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
    const int col_base = n0 + wc * 64;
    if (atomic) {
      // 256 contiguous bytes per wave-instruction: the shape float atomics run fastest at
      const int col = col_base + lane;
      float* C = (float*)p.C + cb;
#pragma unroll 4
      for (int r = 0; r < 32; ++r) {
        const int row = row_base + r;
        const float v = slab[r * EPI_LD + lane];
        if (row < p.M && col < p.N) unsafeAtomicAdd(C + (int64_t)row * p.ldc + col, v);
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const int row = row_base + rl;
        const int col = col_base + (lane & 15) * 4;
        f32x4 v = *(const XVIT_LDS f32x4*)(slab + rl * EPI_LD + (lane & 15) * 4);
        if (row < p.M && col < p.N) {
          if (bias) v += *(const f32x4*)(bias + col);
          if (p.act == XVIT_ACT_GELU) {
            if (aux) {
              bf16x4 z = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
              *(bf16x4*)(aux + (int64_t)row * p.ldaux + col) = z;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
          } else if (p.act == XVIT_ACT_DGELU) {
            const bf16x4 z = *(const bf16x4*)(aux + (int64_t)row * p.ldaux + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(bf2f(z[e]));
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
        }
      }
    }
  }
}

}  // namespace xvit

using namespace xvit;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
  if (a->split_k > 1)
    XVIT_REQUIRE(a->c_dtype == XVIT_F32 && a->act == XVIT_ACT_NONE && !a->bias && !a->residual && a->out_seg_rows == 0, "xvit_gemm: split_k > 1 needs a plain fp32 accumulate epilogue");
  XVIT_REQUIRE(!(a->accumulate && a->c_dtype != XVIT_F32), "xvit_gemm: accumulate needs fp32 C");
  XVIT_REQUIRE(a->act != XVIT_ACT_DGELU || a->aux, "xvit_gemm: ACT_DGELU needs aux (pre-activation)");
  if (a->aux) XVIT_REQUIRE(a->ldaux % 4 == 0 && a->ldaux >= a->N && (reinterpret_cast<uintptr_t>(a->aux) & 7) == 0, "xvit_gemm: bad aux layout");
  if (a->residual) XVIT_REQUIRE(a->ldr % 4 == 0 && a->ldr >= a->N && aligned16(a->residual), "xvit_gemm: bad residual layout");
  if (a->bias) XVIT_REQUIRE(aligned16(a->bias) && a->stride_bias % 4 == 0, "xvit_gemm: bias must be 16-byte aligned");

  GemmParams p;
  p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C; p.bias = a->bias; p.res = a->residual; p.aux = (bf16*)a->aux;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
  p.sA = a->stride_a; p.sB = a->stride_b; p.sC = a->stride_c; p.sBias = a->stride_bias; p.sR = a->stride_r; p.sAux = a->stride_aux;
  p.M = a->M; p.N = a->N; p.K = a->K; p.split_k = a->split_k;
  const int ktiles = (a->K + BK - 1) / BK;
  p.k_per_split = ((ktiles + a->split_k - 1) / a->split_k) * BK;
  p.ntm = (a->M + BM - 1) / BM; p.ntn = (a->N + BN - 1) / BN;
  p.c_f32 = a->c_dtype == XVIT_F32; p.act = a->act; p.accumulate = a->accumulate;
  p.res_row_mod = a->res_row_mod; p.res_row_off = a->res_row_off;
  p.seg_rows = a->out_seg_rows; p.seg_skip = a->out_seg_skip; p.row_off = a->out_row_off;

  const dim3 grid(p.ntm * p.ntn, 1, a->batch * a->split_k), block(256);
  XVIT_REQUIRE((int64_t)a->batch * a->split_k <= 65535, "xvit_gemm: batch*split_k too large");
  hipStream_t s = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    attr_done = true;
  }
  switch (a->layout) {
    case XVIT_GEMM_NT: hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, GEMM_LDS, s, p); break;
    case XVIT_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, GEMM_LDS, s, p); break;
    default: hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, GEMM_LDS, s, p); break;
  }
  return check_launch("xvit_gemm");
}
