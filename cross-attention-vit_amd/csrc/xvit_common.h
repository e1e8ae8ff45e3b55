// Shared device/host helpers for the xvit gfx950 kernels.  CDNA4 only: wave64, MFMA,
// buffer_load ... lds (LDS-DMA), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xvit.h"

namespace xvit {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define XVIT_LDS __attribute__((address_space(3)))

constexpr int kWave = 64;

// ---- error plumbing (host) -----------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
const uint64_t* drop_epoch_ptr();   // xvit_set_dropout_epoch: device address of the dropout epoch counter, or nullptr (core.hip)
void set_attn_peel(int v);   // xvit_set_option("attn_peel") -> attention.hip

#define XVIT_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::xvit::set_error(__VA_ARGS__);      \
      return XVIT_ERR_ARG;                 \
    }                                      \
  } while (0)

// ---- device helpers ------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Wave-uniform value made provably uniform for the compiler (SGPR).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Buffer resource over [base, base+bytes): out-of-range loads return 0, stores are dropped.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ uint32_t clamp_bytes(int64_t b) {
  return b <= 0 ? 0u : (b > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)b);
}

// 16-byte LDS-DMA: LDS[lds_base + lane*16 .. +16) <- buffer[voff + soff .. +16)
//
// Issued from inline asm on purpose.  With the __builtin_amdgcn_raw_ptr_buffer_load_lds form hipcc's
// waitcnt pass cannot prove that a later ds_read_b64_tr_b16 does not alias the DMA destination and puts
// `s_waitcnt vmcnt(0)` in front of the first transposed read of every iteration, which drains the
// next-stage prefetch before any MFMA has issued (seen in the NN/TN GEMM and all attention loops).  The
// kernels order DMA -> LDS read themselves (counted `s_waitcnt vmcnt` + s_barrier), so the compiler must
// simply not model these loads.  Untracked VMEM ops can only make the compiler's own vmcnt waits
// stricter, never weaker (the counter retires in order).  lds_base, soff and r must be wave-uniform.
// M0 is written in the same statement that consumes it; nothing else in these kernels uses M0.
#ifndef XVIT_GLDS_POLICY
#define XVIT_GLDS_POLICY ""      // cache-policy modifiers of the LDS-DMA loads (" nt", " sc1", ...): A/B builds only
#endif
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, XVIT_LDS void* lds_base, uint32_t voff, uint32_t soff) {
  const uint32_t m0v = (uint32_t)(uintptr_t)lds_base;
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen" XVIT_GLDS_POLICY " lds"
               :
               : "v"(voff), "s"(r), "s"(soff), "s"(m0v)
               : "memory");
}
// 4-byte variant: LDS[lds_base + lane*4 .. +4) <- buffer[voff + soff .. +4)
__device__ __forceinline__ void glds4(__amdgpu_buffer_rsrc_t r, XVIT_LDS void* lds_base, uint32_t voff, uint32_t soff) {
  const uint32_t m0v = (uint32_t)(uintptr_t)lds_base;
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(r), "s"(soff), "s"(m0v)
               : "memory");
}

__device__ __forceinline__ s16x4 lds_read_tr16(const XVIT_LDS void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((XVIT_LDS s16x4*)p);
}

// Wave-wide reductions, result in every lane.  Four DPP steps inside each 16-lane row (quad_perm, row_half_mirror, row_mirror: the
// operand permutation rides on the VALU instruction itself) and two row / half exchanges (v_permlane16_swap, v_permlane32_swap):
// eight VALU instructions, no LDS pipe.  (__shfl_xor compiles to ds_bpermute_b32: six dependent LDS round trips of ~100 cycles each,
// which is what a one-row-per-wave LayerNorm spends its time waiting on.)
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void swap_rows16(float x, float& a, float& b) {   // (x of the even row, x of the odd row) of each row pair, in both rows
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t_;
  const unsigned bits = __builtin_bit_cast(unsigned, x);
  const u32x2_t_ r = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
  const unsigned r0 = r.x, r1 = r.y;
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void swap_halves32(float x, float& a, float& b) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t_;
  const unsigned bits = __builtin_bit_cast(unsigned, x);
  const u32x2_t_ r = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
  const unsigned r0 = r.x, r1 = r.y;
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_perm<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_perm<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_perm<0x141>(v);   // row_half_mirror
  v += dpp_perm<0x140>(v);   // row_mirror
  float a, b;
  swap_rows16(v, a, b);
  v = a + b;
  swap_halves32(v, a, b);
  return a + b;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_perm<0xB1>(v));
  v = fmaxf(v, dpp_perm<0x4E>(v));
  v = fmaxf(v, dpp_perm<0x141>(v));
  v = fmaxf(v, dpp_perm<0x140>(v));
  float a, b;
  swap_rows16(v, a, b);
  v = fmaxf(a, b);
  swap_halves32(v, a, b);
  return fmaxf(a, b);
}

// Combine a value across the two 32-lane halves of a wave (lane l with lane l ^ 32) in ONE VALU instruction:
// v_permlane32_swap exchanges the upper half of one register with the lower half of another, so afterwards the pair
// holds (x.lo, x.lo) and (x.hi, x.hi).  __shfl_xor(x, 32) compiles to a ds_bpermute round trip through the LDS
// pipe (address arithmetic + lgkmcnt wait) instead.
// (Note for anyone touching this: __builtin_bit_cast(float, vec.y) on an ext-vector ELEMENT reads element 0 with this
// hipcc — go through a scalar temporary, as below.)
__device__ __forceinline__ void swap_halves(float x, float& lo, float& hi) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t_;
  const unsigned bits = __builtin_bit_cast(unsigned, x);
  const u32x2_t_ r = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
  const unsigned r0 = r.x, r1 = r.y;
  lo = __builtin_bit_cast(float, r0);
  hi = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float half_max(float x) {
  float lo, hi;
  swap_halves(x, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float half_sum(float x) {
  float lo, hi;
  swap_halves(x, lo, hi);
  return lo + hi;
}

// Exact-form (erf) GELU and its derivative in fp32.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// i.e. fp32 round-off level; libm erff costs ~4x the VALU work and the epilogue is VALU-bound):
//   erf(u) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) e^{-u^2},  t = 1 / (1 + p u),  u >= 0.
// GELU and GELU' share the one exponential: with u = |x|/sqrt(2), e^{-u^2} = e^{-x^2/2} is also the
// normal pdf up to 1/sqrt(2 pi).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  const float e = __expf(-u * u);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erf_abs = fmaf(-poly * t, e, 1.0f);
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ void gelu_and_grad(float x, float& a, float& d) {   // a = gelu(x), d = gelu'(x): one exponential for both
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  a = x * cdf;
  d = fmaf(x, pdf, cdf);
}
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}

// counter-based RNG for dropout: one 32-bit hash per element index (same mask in fwd and bwd)
__device__ __forceinline__ uint32_t hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

// A captured step (HIP graph) freezes every kernel argument, the dropout seeds included.  A launch may therefore carry the device address
// of an epoch counter (xvit_set_dropout_epoch): the seed a kernel then uses is seed + epoch * odd constant, read at run time, so that a
// replay whose graph increments the counter first draws new masks — the same ones in its forward and its backward.
__device__ __forceinline__ uint64_t drop_seed_at(uint64_t seed, const uint64_t* __restrict__ epoch) {
  return epoch ? seed + *epoch * 0xD1B54A32D192ED03ull : seed;
}

}  // namespace xvit
