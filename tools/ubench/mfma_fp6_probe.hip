// v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands on gfx950, operands built by the conversion instructions the kernels
// would use: A by v_cvt_scalef32_2xpk16_fp6_f32 (16 + 16 floats), B by v_cvt_scalef32_pk32_fp6_f16 (32 halves).  Which source
// position of A meets which source position of B, which positions form the two scale blocks, what the decode returns, and the
// rate of "32 f16 + 4 fp6 MFMAs" against "32 f16 + 4 fp8" and "48 f16".
// hipcc --offload-arch=gfx950 -O3 -o mfma_fp6_probe mfma_fp6_probe.hip && ./mfma_fp6_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ i32x8 widen(i32x6 v) { return i32x8{v[0], v[1], v[2], v[3], v[4], v[5], 0, 0}; }

// a_in: [64 lanes][32] floats (0..15 -> src0, 16..31 -> src1); b_in: [64][32] floats (-> 32 halves)
__global__ void k_probe(const float* a_in, const float* b_in, const int* sa, const int* sb, float cvt_scale, float* out, float* dec) {
  const int lane = threadIdx.x;
  f32x16 a0, a1; f16x32 bh;
  for (int i = 0; i < 16; ++i) { a0[i] = a_in[lane * 32 + i]; a1[i] = a_in[lane * 32 + 16 + i]; }
  for (int i = 0; i < 32; ++i) bh[i] = (_Float16)b_in[lane * 32 + i];
  const i32x6 a6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, cvt_scale);
  const i32x6 b6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(bh, cvt_scale);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a6), widen(b6), acc, 2, 2, 0, sa[lane], 0, sb[lane]);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
  const f16x32 d = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(a6, cvt_scale);
  for (int i = 0; i < 32; ++i) dec[lane * 32 + i] = (float)d[i];
}

template <int MIX>   // 0: 48 f16; 1: 32 f16 + 4 fp8; 2: 32 f16 + 4 fp6
__global__ void __launch_bounds__(256, 2) k_rate(const f16x8* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(blockIdx.x * 8 + j) * 64 + lane]; b[j] = src[((blockIdx.x + 7) * 8 + j) * 64 + lane]; }
  i32x8 a8[2], b8[2];
  for (int q = 0; q < 2; ++q)
    for (int j = 0; j < 8; ++j) {
      a8[q][j] = ((const int*)&a[2 * q + (j >> 2)])[j & 3] & 0x7e7e7e7e;
      b8[q][j] = ((const int*)&b[2 * q + (j >> 2)])[j & 3] & 0x7e7e7e7e;
    }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int sc = 0x70707070;
  for (int it = 0; it < iters; ++it) {
    constexpr int kF16 = MIX == 0 ? 48 : 32;
#pragma unroll
    for (int u = 0; u < kF16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u >> 2) & 7], acc[u & 3], 0, 0, 0);
    if (MIX == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[u & 1], b8[(u >> 1) & 1], acc[u & 3], 0, 0, 0, sc, 0, sc);
    } else if (MIX == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[u & 1], b8[(u >> 1) & 1], acc[u & 3], 2, 2, 0, sc, 0, sc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] *= 0.001f;
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
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

int main() {
  float *a_d, *b_d, *out_d, *dec_d; int *sa_d, *sb_d;
  CK(hipMalloc(&a_d, 64 * 32 * 4)); CK(hipMalloc(&b_d, 64 * 32 * 4)); CK(hipMalloc(&out_d, 64 * 16 * 4)); CK(hipMalloc(&dec_d, 64 * 32 * 4));
  CK(hipMalloc(&sa_d, 256)); CK(hipMalloc(&sb_d, 256));
  std::vector<float> A(64 * 32), Bm(64 * 32), out(64 * 16), dec(64 * 32);
  std::vector<int> sa(64, 127), sb(64, 127);
  float cvt_scale = 1.0f;
  auto launch = [&]() {
    CK(hipMemcpy(a_d, A.data(), 64 * 32 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b_d, Bm.data(), 64 * 32 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sa_d, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sb_d, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a_d, b_d, sa_d, sb_d, cvt_scale, out_d, dec_d);
    CK(hipMemcpy(out.data(), out_d, 64 * 16 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(dec.data(), dec_d, 64 * 32 * 4, hipMemcpyDeviceToHost));
  };
  auto D = [&](int row, int col) { const int hh = (row >> 2) & 1, reg = (row & 3) + 4 * (row >> 3); return out[(col + 32 * hh) * 16 + reg]; };
  // ---- pairing: B one-hot at source position j of lane half h' (column 5); A carries codes ----
  bool natural = true;
  for (int hb = 0; hb < 2; ++hb)
    for (int j = 0; j < 32; ++j) {
      float got[3];
      for (int runi = 0; runi < 3; ++runi) {
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 32; ++i)
            A[l * 32 + i] = runi == 0 ? 0.125f * ((i & 15) + 1) : runi == 1 ? (i < 16 ? 1.0f : 2.0f) : (l < 32 ? 1.0f : 2.0f);
        std::fill(Bm.begin(), Bm.end(), 0.f);
        Bm[(5 + 32 * hb) * 32 + j] = 1.0f;
        launch();
        got[runi] = D(9, 5);
      }
      const int ai = (int)(got[0] / 0.125f + 0.5f) - 1 + (got[1] > 1.5f ? 16 : 0), ah = got[2] > 1.5f ? 1 : 0;
      if (ai != j || ah != hb) natural = false;
      if (j % 8 == 0 || ai != j || ah != hb) printf("  B(h'=%d, src %2d) x A(h=%d, src %2d)   [codes %g %g %g]\n", hb, j, ah, ai, got[0], got[1], got[2]);
    }
  printf("pairing => %s\n", natural ? "source position i (0..15 = first source, 16..31 = second) of lane half h meets the same (h, i) of the other operand" : "PERMUTED, see above");
  // decode of A as converted (lane 9)
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) A[l * 32 + i] = 0.125f * ((i & 15) + 1) * (i < 16 ? 1.f : 2.f);
  launch();
  printf("decode(lane 9):"); for (int i = 0; i < 32; ++i) printf(" %g", dec[9 * 32 + i]); printf("\n");
  // conversion scale and saturation
  cvt_scale = 4.0f;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) A[l * 32 + i] = (float)(i + 1);
  launch();
  printf("scale 4, inputs 1..32 decoded with the same scale:"); for (int i = 0; i < 32; ++i) printf(" %g", dec[9 * 32 + i]); printf("\n");
  cvt_scale = 1.0f;
  // ---- scale blocks ----
  for (int which = 0; which < 4; ++which) {
    std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
    if (which == 0) sa[9] = 130; else if (which == 1) sa[41] = 130; else if (which == 2) sb[5] = 130; else sb[37] = 130;
    printf("scale 2^3 in byte 0 of %s lane %2d:", which < 2 ? "scale_a" : "scale_b", which == 0 ? 9 : which == 1 ? 41 : which == 2 ? 5 : 37);
    for (int hb = 0; hb < 2; ++hb)
      for (int j = 0; j < 32; j += 8) {
        std::fill(A.begin(), A.end(), 1.0f); std::fill(Bm.begin(), Bm.end(), 0.f); Bm[(5 + 32 * hb) * 32 + j] = 1.0f;
        launch();
        printf("  (h=%d,src %2d): %g", hb, j, D(9, 5));
      }
    printf("\n");
  }
  // ---- rate ----
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
    printf("rate: 48 f16: %.3f ms | 32 f16 + 4 fp8: %.3f ms = %.3f x | 32 f16 + 4 fp6: %.3f ms = %.3f x\n", t0, t1, t1 / t0, t2, t2 / t0);
  }
  return 0;
}
