// How much vector work hides under a 32x32x16 f16 MFMA?  Loop body: one MFMA (dependent chain) + K independent v_fma_f32 on other registers,
// for K = 0 .. 12, with one and with two waves per SIMD (32 workgroups only: the clock stays near its maximum, cycles are what is compared).
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int K>
__global__ void __launch_bounds__(256, 2) k(const f16x8* __restrict__ src, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  const f16x8 a = src[blockIdx.x * 64 + lane], b = src[(blockIdx.x + 7) * 64 + lane];
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = (float)a[i & 7] + i;
  const float m = (float)b[0] * 1e-3f + 1.0f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K; ++i) v[i] = __builtin_fmaf(v[i], m, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int K>
void run(const f16x8* src, float* d, long long* cyc, int grid, int waves_per_simd) {
  const int iters = 4000;
  hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, src, d, iters, cyc);
  hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, src, d, iters, cyc);
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("  K = %2d vector instructions per MFMA: %.1f cycles per MFMA\n", K, (double)c / (iters * 8.0));
}
int main() {
  const size_t n = 1 << 20;
  std::vector<_Float16> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
  f16x8* src; float* d; long long* cyc;
  CK(hipMalloc(&src, n * 2)); CK(hipMalloc(&d, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 8));
  CK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  for (int wps = 1; wps <= 2; ++wps) {
    // wps = 2: two workgroups per CU need every CU filled twice - use the whole chip (the clock drops, cycles still compare)
    const int grid = wps == 1 ? 32 : 2 * prop.multiProcessorCount;
    printf("%d wave(s) per SIMD (%d workgroups):\n", wps, grid);
    run<0>(src, d, cyc, grid, wps); run<2>(src, d, cyc, grid, wps); run<4>(src, d, cyc, grid, wps); run<6>(src, d, cyc, grid, wps);
    run<8>(src, d, cyc, grid, wps); run<10>(src, d, cyc, grid, wps); run<12>(src, d, cyc, grid, wps);
  }
  return 0;
}
