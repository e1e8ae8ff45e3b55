// Diagnostic (not part of the product): operand lane maps, scale semantics and fp8 encoding of
// v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3) on gfx950, found with one-hot operands — the attention fp8 kernel
// (csrc/attention_fp8.hip) relies on.  Found: lane (r = l & 31, h = l >> 5) holds row r of A / column r of B; byte p of an A lane
// pairs with byte p of the B lanes of the SAME half; the lane's scale byte (e8m0, 2^(b - 127), byte opsel of the scale VGPR)
// does NOT apply to the lane's own 32 bytes but to bytes 0..15 of both lanes of the row (h = 0) or bytes 16..31 of both (h = 1):
// the instruction is two K = 32 steps, byte p of lane (r, h) is k = 32 (p / 16) + 16 h + p % 16, scale block h is k = 32 h .. + 31.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_scale_probe.hip -o tools/_bin/mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void run(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, float* d, int opa, int opb) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = ((const int*)a)[l * 8 + i]; bv[i] = ((const int*)b)[l * 8 + i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  if (opa == 0 && opb == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
  else if (opa == 1 && opb == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 1, sa[l], 0, sb[l]);
  else if (opa == 2 && opb == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 2, sa[l], 3, sb[l]);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;   // standard 32x32 C/D map
    d[row * 32 + col] = c[i];
  }
}
__global__ void cvt(const float* x, unsigned* out) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
  out[0] = (unsigned)w;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], w, true);
  out[1] = (unsigned)w;
}

int main() {
  unsigned char *a, *b; int *sa, *sb; float* d;
  hipMalloc(&a, 2048); hipMalloc(&b, 2048); hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&d, 4096);
  std::vector<unsigned char> ha(2048), hb(2048);
  std::vector<int> hs(64, 0x7F7F7F7F);
  std::vector<float> hd(1024);
  const unsigned char ONE = 0x38;   // OCP e4m3: 1.0
  auto launch = [&](int opa, int opb, const std::vector<int>& s_a, const std::vector<int>& s_b) {
    hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(sa, s_a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(sb, s_b.data(), 256, hipMemcpyHostToDevice);
    run<<<1, 64>>>(a, b, sa, sb, d, opa, opb);
    hipMemcpy(hd.data(), d, 4096, hipMemcpyDeviceToHost);
  };
  // 1. row / col of a lane, k pairing
  int bad = 0;
  for (int la : {0, 7, 32, 45}) for (int pa = 0; pa < 32; ++pa) {
    std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
    ha[la * 32 + pa] = ONE;
    const int lb = (la & 32) + 5, pb = pa;   // expected partner: same half, same byte, any column
    hb[lb * 32 + pb] = ONE;
    launch(0, 0, hs, hs);
    int nz = 0, r = -1, c = -1;
    for (int i = 0; i < 1024; ++i) if (hd[i] != 0.f) { ++nz; r = i / 32; c = i % 32; }
    const bool ok = nz == 1 && r == (la & 31) && c == 5 && hd[r * 32 + c] == 1.0f;
    if (!ok) { ++bad; printf("A lane %d byte %d x B lane %d byte %d: %d nonzero, at (%d, %d) value %g\n", la, pa, lb, pb, nz, r, c, nz ? hd[r * 32 + c] : 0.f); }
  }
  // other half / other byte must NOT pair
  for (int pa : {0, 13}) {
    std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0);
    ha[0 * 32 + pa] = ONE; hb[37 * 32 + pa] = ONE;   // same byte, other half
    launch(0, 0, hs, hs);
    for (int i = 0; i < 1024; ++i) if (hd[i] != 0.f) { ++bad; printf("cross-half pairing at byte %d!\n", pa); break; }
    std::fill(hb.begin(), hb.end(), 0); hb[5 * 32 + ((pa + 1) & 31)] = ONE;   // same half, other byte
    launch(0, 0, hs, hs);
    for (int i = 0; i < 1024; ++i) if (hd[i] != 0.f) { ++bad; printf("cross-byte pairing at byte %d!\n", pa); break; }
  }
  printf("byte pairing (A lane (r, h) byte p x B lane (c, h) byte p -> D[r][c], nothing across halves or bytes): %s\n", bad ? "MISMATCH" : "confirmed");
  // 2. scales: all ones operands; A lanes' scale byte 0 = 2^1 for h = 0 and 2^3 for h = 1, B scale 2^-2: D = 32*2*0.25 + 32*8*0.25 = 80
  std::fill(ha.begin(), ha.end(), ONE); std::fill(hb.begin(), hb.end(), ONE);
  std::vector<int> s_a(64), s_b(64, 0x7F7F7F7D);
  for (int l = 0; l < 64; ++l) s_a[l] = l < 32 ? 0x11111180 : 0x11111182;   // byte 0 selected by opsel 0
  launch(0, 0, s_a, s_b);
  printf("scales, opsel 0/0: D[0][0] = %g (expect 80), D[31][31] = %g\n", hd[0], hd[1023]);
  for (int l = 0; l < 64; ++l) s_a[l] = l < 32 ? 0x11118011 : 0x11118211;   // byte 1
  launch(1, 0, s_a, s_b);
  printf("scales, opsel 1/0: D[0][0] = %g (expect 80)\n", hd[0]);
  for (int l = 0; l < 64; ++l) { s_a[l] = l < 32 ? 0x11801111 : 0x11821111; s_b[l] = 0x7D111111; }   // A byte 2, B byte 3
  launch(2, 3, s_a, s_b);
  printf("scales, opsel 2/3: D[0][0] = %g (expect 80)\n", hd[0]);
  // per-lane scale: only lane 3's A scale doubled -> row 3 differs
  std::fill(s_a.begin(), s_a.end(), 0x7F7F7F7F); s_a[3] = 0x7F7F7F80; std::fill(s_b.begin(), s_b.end(), 0x7F7F7F7F);
  launch(0, 0, s_a, s_b);
  printf("per-lane A scale: D[3][0] = %g (expect 96), D[4][0] = %g (expect 64)\n", hd[3 * 32], hd[4 * 32]);
  // per-lane B scale: lane 5 (col 5, k 0..31) x2, lane 37 (col 5, k 32..63) x8
  std::fill(s_a.begin(), s_a.end(), 0x7F7F7F7F); std::fill(s_b.begin(), s_b.end(), 0x7F7F7F7F); s_b[5] = 0x7F7F7F80; s_b[37] = 0x7F7F7F82;
  launch(0, 0, s_a, s_b);
  printf("per-lane B scale: D[0][5] = %g (expect 32*2 + 32*8 = 320), D[0][6] = %g (expect 64), D[5][0] = %g (expect 64)\n", hd[5], hd[6], hd[5 * 32]);
  // which bytes does a lane's scale cover?  A = ones on the h = 0 lanes only, B scales as above
  std::fill(ha.begin(), ha.end(), 0); for (int l = 0; l < 32; ++l) for (int p = 0; p < 32; ++p) ha[l * 32 + p] = ONE;
  launch(0, 0, s_a, s_b);
  printf("A = ones on the h = 0 lanes only: D[0][5] = %g (64 if a lane's scale covered its own 32 bytes; 160 = 16*2 + 16*8: bytes 0..15 are in scale block 0, bytes 16..31 in block 1), D[0][6] = %g\n", hd[5], hd[6]);
  // 3. cvt_pk_fp8_f32 encoding
  float hx[8] = {1.0f, -2.0f, 0.5f, 448.0f, 1000.0f, 0.0019531f, 1.0625f, 1.1875f}; float* dx; unsigned* dw; unsigned hw[2];
  hipMalloc(&dx, 32); hipMalloc(&dw, 8); hipMemcpy(dx, hx, 32, hipMemcpyHostToDevice);
  cvt<<<1, 1>>>(dx, dw); hipMemcpy(hw, dw, 8, hipMemcpyDeviceToHost);
  printf("cvt_pk_fp8_f32(1, -2, .5, 448 | 1000, 2^-9, 1.0625, 1.1875) = %08x %08x  (OCP e4m3fn: 38 c0 30 7e | 7e(sat) or 7f(nan), 01, 38 or 39 (RNE), 3a)\n", hw[0], hw[1]);
  return 0;
}
