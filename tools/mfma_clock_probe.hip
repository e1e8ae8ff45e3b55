// Diagnostic (not part of the product): what MFMA rate does the chip SUSTAIN when n CUs issue v_mfma_f32_32x32x16_bf16
// back to back from registers (no memory traffic at all), and at what shader clock?  Each block = 8 waves (2 per SIMD, the
// occupancy of gemm_big_kernel), each wave keeps 4 independent accumulator tiles in flight.  clock64() counts shader
// cycles, wall_clock64() the constant 100 MHz reference: their ratio is the clock the CU actually ran at.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o tools/_bin/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(4))) float f32x4;

// the same FLOPs per iteration from v_mfma_f32_16x16x32_bf16 (32 per iteration on 8 accumulator tiles = the same 64 accumulator registers)
__global__ __launch_bounds__(512) void mfma_loop16(int iters, long long* cyc, long long* wall, float* sink) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  unsigned h = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
  for (int j = 0; j < 8; ++j) {
    h = h * 1664525u + 1013904223u; a[j] = (__bf16)(((int)(h >> 8 & 0xFFFF) - 32768) / 16384.0f);
    h = h * 1664525u + 1013904223u; b[j] = (__bf16)(((int)(h >> 8 & 0xFFFF) - 32768) / 16384.0f);
  }
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}

__global__ __launch_bounds__(512) void mfma_loop(int iters, long long* cyc, long long* wall, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  unsigned h = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
  for (int j = 0; j < 8; ++j) {   // random operands of magnitude ~1 (data-dependent power: constants would flatter the clock)
    h = h * 1664525u + 1013904223u; a[j] = (__bf16)(((int)(h >> 8 & 0xFFFF) - 32768) / 16384.0f);
    h = h * 1664525u + 1013904223u; b[j] = (__bf16)(((int)(h >> 8 & 0xFFFF) - 32768) / 16384.0f);
  }
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;   // 16 MFMAs per iteration per wave
  long long *cyc, *wall; float* sink;
  hipMalloc(&cyc, 4096 * 8); hipMalloc(&wall, 4096 * 8); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {16, -16, 128, -128, 256, -256, 256, -256, 256, -256};   // negative: the 16x16x32 loop
  for (int gs : grids) {
    const int g = gs < 0 ? -gs : gs;
    if (gs < 0) mfma_loop16<<<g, 512>>>(iters / 10, cyc, wall, sink); else mfma_loop<<<g, 512>>>(iters / 10, cyc, wall, sink);
    hipEventRecord(e0);
    if (gs < 0) mfma_loop16<<<g, 512>>>(iters, cyc, wall, sink); else mfma_loop<<<g, 512>>>(iters, cyc, wall, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(g), w(g);
    hipMemcpy(c.data(), cyc, g * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), wall, g * 8, hipMemcpyDeviceToHost);
    double mhz_min = 1e9, mhz_max = 0, mhz_sum = 0;
    for (int i = 0; i < g; ++i) { const double f = (double)c[i] / ((double)w[i] / 100.0); mhz_min = std::min(mhz_min, f); mhz_max = std::max(mhz_max, f); mhz_sum += f; }
    const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8.0 * g;
    const double cyc_per_mfma = ms * 1e-3 * (mhz_sum / g) * 1e6 / ((gs < 0 ? 32.0 : 16.0) * iters * 2.0);   // launch time x measured clock / MFMAs per SIMD   // as counted by clock64()   // 2 waves share a SIMD
    printf("%s %3d blocks: %8.3f ms  %7.1f TFLOP/s  (%5.2f per CU)  clock64/wall: %6.0f MHz mean (%6.0f .. %6.0f)   %.1f cycles per MFMA on one SIMD\n",
           gs < 0 ? "16x16x32" : "32x32x16", g, ms, flops / ms / 1e9, flops / ms / 1e9 / g, mhz_sum / g, mhz_min, mhz_max, cyc_per_mfma);
    fflush(stdout);
  }
  return 0;
}
