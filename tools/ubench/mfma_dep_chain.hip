// One wave per SIMD (a key-split attention workgroup at B = 1): cycles per 32x32x16 f16 MFMA when every MFMA accumulates into the SAME
// registers (S = K Q'^T over its 24 partial products) against 2 and 4 alternating accumulators.  No other instructions in the loop.
// hipcc --offload-arch=gfx950 -O3 -o mfma_dep_chain mfma_dep_chain.hip && ./mfma_dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int NACC>
__global__ void __launch_bounds__(256, 1) k(const f16x8* __restrict__ src, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  f16x8 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(blockIdx.x * 8 + j) * 64 + lane]; b[j] = src[((blockIdx.x + 7) * 8 + j) * 64 + lane]; }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u >> 2) & 7], acc[u % NACC], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  const size_t n = 1 << 22;
  std::vector<_Float16> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
  f16x8* src; float* d; long long* cyc;
  CK(hipMalloc(&src, n * 2)); CK(hipMalloc(&d, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 8));
  CK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int pass = 0; pass < 6; ++pass) {
    const int nacc = (pass < 3) ? (4 >> pass) : (1 << (pass - 3));      // 4, 2, 1, 1, 2, 4: order effects (temperature) show up as asymmetry
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, src, d, iters, cyc);
      else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, src, d, iters, cyc);
      else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, src, d, iters, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%d workgroups of 4 waves, %d accumulator(s): %.1f ns per MFMA; s_memtime ticks per MFMA %.1f (=> %.2f GHz if a tick is a core cycle)\n", grid, nacc,
           ms * 1e6 / (iters * 24.0), (double)c / (iters * 24.0), (double)c / (ms * 1e6));
  }
  return 0;
}
