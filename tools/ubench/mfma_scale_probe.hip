// v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950: (1) which byte of which lane of A meets which byte of which
// lane of B (the k pairing), and what the E8M0 scale bytes do; (2) the rate and the clock of the mix "32 f16 MFMAs + 4 scaled
// fp8 MFMAs" against "48 f16 MFMAs" on random data (the attention tile with the two cross products of P.V on the fp8 pipe).
// hipcc --offload-arch=gfx950 -O3 -o mfma_scale_probe mfma_scale_probe.hip && ./mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_probe(const int* a_in, const int* b_in, const int* sa, const int* sb, float* out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = a_in[lane * 8 + j]; b[j] = b_in[lane * 8 + j]; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa[lane], 0, sb[lane]);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

// MIX 0: 48 x f16 32x32x16 per "tile"; MIX 1: 32 x f16 + 4 x scaled fp8 32x32x64; MIX 2: 16 f16 + 8 fp8 (both products' cross terms)
template <int MIX>
__global__ void __launch_bounds__(256, 2) k_rate(const f16x8* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(blockIdx.x * 8 + j) * 64 + lane]; b[j] = src[((blockIdx.x + 7) * 8 + j) * 64 + lane]; }
  i32x8 a8[2], b8[2];
  for (int q = 0; q < 2; ++q)
    for (int j = 0; j < 8; ++j) {
      // random bytes without the two NaN encodings of e4m3 (0x7f, 0xff)
      a8[q][j] = ((const int*)&a[2 * q + (j >> 2)])[j & 3] & 0x7e7e7e7e;
      b8[q][j] = ((const int*)&b[2 * q + (j >> 2)])[j & 3] & 0x7e7e7e7e;
    }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int sc = 0x70707070;   // 2^-15 per block: keeps the accumulators finite
  for (int it = 0; it < iters; ++it) {
    constexpr int kF16 = MIX == 0 ? 48 : MIX == 1 ? 32 : 16;
#pragma unroll
    for (int u = 0; u < kF16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u >> 2) & 7], acc[u & 3], 0, 0, 0);
    constexpr int kF8 = MIX == 0 ? 0 : MIX == 1 ? 4 : 8;
#pragma unroll
    for (int u = 0; u < kF8; ++u)
      acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[u & 1], b8[(u >> 1) & 1], acc[u & 3], 0, 0, 0, sc, 0, sc);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] *= 0.001f;      // (keeps the values bounded; 64 VALU per "tile")
  }
  float sres = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) sres += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = sres;
}

template <int MIX>
float run_rate(const f16x8* src, float* d, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<MIX>), dim3(512), dim3(256), 0, 0, src, d, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_rate<MIX>), dim3(512), dim3(256), 0, 0, src, d, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

int main() {
  int *a_d, *b_d, *sa_d, *sb_d; float* out_d;
  CK(hipMalloc(&a_d, 64 * 8 * 4)); CK(hipMalloc(&b_d, 64 * 8 * 4)); CK(hipMalloc(&sa_d, 256)); CK(hipMalloc(&sb_d, 256));
  CK(hipMalloc(&out_d, 64 * 16 * 4));
  std::vector<unsigned char> A(64 * 32), Bm(64 * 32);
  std::vector<int> sa(64, 127), sb(64, 127);
  std::vector<float> out(64 * 16);
  // e4m3 codes of the integers 1..8: 1 = 0x38, 2 = 0x40, 3 = 0x44, 4 = 0x48, 5 = 0x4a, 6 = 0x4c, 7 = 0x4e, 8 = 0x50
  const unsigned char code[9] = {0, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  auto launch = [&]() {
    CK(hipMemcpy(a_d, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(b_d, Bm.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(sa_d, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sb_d, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a_d, b_d, sa_d, sb_d, out_d);
    CK(hipMemcpy(out.data(), out_d, 4096, hipMemcpyDeviceToHost));
  };
  // D[row][col]: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  auto D = [&](int row, int col) { const int hh = (row >> 2) & 1, reg = (row & 3) + 4 * (row >> 3); return out[(col + 32 * hh) * 16 + reg]; };
  printf("pairing: B position (half h', byte b') of column 5 meets A position (half, byte) of row 9:\n");
  bool natural = true;
  for (int hb = 0; hb < 2; ++hb)
    for (int bb = 0; bb < 32; ++bb) {
      int got[3];
      for (int runi = 0; runi < 3; ++runi) {
        for (int l = 0; l < 64; ++l)
          for (int b = 0; b < 32; ++b) A[l * 32 + b] = code[runi == 0 ? 1 + (b & 7) : runi == 1 ? 1 + (b >> 3) : 1 + (l >> 5)];
        std::fill(Bm.begin(), Bm.end(), 0);
        Bm[(5 + 32 * hb) * 32 + bb] = 0x38;
        launch();
        got[runi] = (int)D(9, 5);
      }
      const int ah = got[2] - 1, ab = (got[0] - 1) + 8 * (got[1] - 1);
      if (ah != hb || ab != bb) natural = false;
      if (bb % 8 == 0 || ah != hb || ab != bb) printf("  B(h'=%d, b'=%2d) x A(h=%d, b=%2d)\n", hb, bb, ah, ab);
    }
  printf("  => %s\n", natural ? "same (half, byte) on both sides: k = 32 * half + byte" : "PERMUTED, see above");
  // scales: A all ones, B one-hot at (half h', byte b'); which LANE's scale byte multiplies that k?
  for (int which = 0; which < 4; ++which) {      // scale 2^3 on: lane 9 (A) | lane 41 (A) | lane 5 (B) | lane 37 (B)
    std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
    if (which == 0) sa[9] = 130; else if (which == 1) sa[41] = 130; else if (which == 2) sb[5] = 130; else sb[37] = 130;
    printf("scale 2^3 in byte 0 of %s lane %d:", which < 2 ? "scale_a" : "scale_b", which == 0 ? 9 : which == 1 ? 41 : which == 2 ? 5 : 37);
    for (int hb = 0; hb < 2; ++hb)
      for (int bb = 0; bb < 32; bb += 8) {
        std::fill(A.begin(), A.end(), 0x38); std::fill(Bm.begin(), Bm.end(), 0); Bm[(5 + 32 * hb) * 32 + bb] = 0x38;
        launch();
        printf("  (h=%d,b=%2d): %g", hb, bb, D(9, 5));
      }
    printf("\n");
  }
  std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
  sa[9] = 127 | (130 << 8); sa[41] = 127 | (130 << 8);
  std::fill(A.begin(), A.end(), 0x38); std::fill(Bm.begin(), Bm.end(), 0); Bm[5 * 32 + 0] = 0x38;
  launch();
  printf("scale byte 1 set to 2^3, byte 0 = 127, opsel 0: D = %g (expect 1)\n", D(9, 5));
  // rate
  const size_t n = 1 << 22;
  std::vector<_Float16> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
  f16x8* src; float* d;
  CK(hipMalloc(&src, n * 2)); CK(hipMalloc(&d, 512 * 256 * 4));
  CK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    const float t0 = run_rate<0>(src, d, iters), t1 = run_rate<1>(src, d, iters), t2 = run_rate<2>(src, d, iters);
    // matrix-pipe cycles per iteration and wave: 48 x 32 = 1536 | 32 x 32 + 4 x 64 = 1280 | 16 x 32 + 8 x 64 = 1024; 2 waves per SIMD
    auto ghz = [&](float ms, int cyc) { return 2.0 * cyc * iters / (ms * 1e-3) / 1e9; };
    printf("rate: 48 f16: %.3f ms (%.2f GHz if pipe-bound) | 32 f16 + 4 fp8: %.3f ms (%.2f GHz) = %.3f x | 16 f16 + 8 fp8: %.3f ms (%.2f GHz) = %.3f x\n",
           t0, ghz(t0, 1536), t1, ghz(t1, 1280), t1 / t0, t2, ghz(t2, 1024), t2 / t0);
  }
  return 0;
}
