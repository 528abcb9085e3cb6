// Microbenchmark: f16 MFMA 32x32x16 vs 16x16x32 at equal FLOPs on random data (DVFS: MI355X_MICROARCH.md item 7).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 32x32x16 (one 16-register accumulator chain x4), SHAPE 1: 16x16x32 (16 accumulators of 4 registers)
template <int SHAPE, bool LDS>
__global__ void __launch_bounds__(256, 2) k(const f16x8* __restrict__ src, float* out, int iters) {
  __shared__ f16x8 sh[1024];
  const int lane = threadIdx.x & 63;
  f16x8 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(blockIdx.x * 8 + j) * 64 + lane]; b[j] = src[((blockIdx.x + 7) * 8 + j) * 64 + lane]; }
  for (int j = threadIdx.x; j < 1024; j += 256) sh[j] = src[j + 4096];
  __syncthreads();
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc32[q][r] = 0.f;
  for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) acc16[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f16x8 av = a[s];
      if (LDS) av = sh[((it + s) & 15) * 64 + lane];
      if (SHAPE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc32[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b[(s + q) & 7], acc32[q], 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc16[(q + 8 * (s & 1))] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b[(s + q) & 7], acc16[(q + 8 * (s & 1))], 0, 0, 0);
      }
    }
  }
  float sres = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) sres += acc32[q][r];
  for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) sres += acc16[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = sres;
}

template <int SHAPE, bool LDS>
float run(const f16x8* src, float* d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(512), dim3(256), 0, 0, src, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(512), dim3(256), 0, 0, src, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const size_t n = 1 << 22;
  std::vector<_Float16> hsrc(n * 8);
  srand(1);
  for (auto& v : hsrc) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
  f16x8* src; float* d;
  (void)hipMalloc(&src, n * 16); (void)hipMalloc(&d, 512 * 256 * 4);
  (void)hipMemcpy(src, hsrc.data(), n * 16, hipMemcpyHostToDevice);
  const int iters = 20000;   // x 32 MFMAs of 32x32x16 (or 64 of 16x16x32) per wave
  const double flops = 512.0 * 4 * iters * 32 * 32768.0;
  for (int rep = 0; rep < 2; ++rep) {
    float t;
    t = run<0, false>(src, d, iters); printf("32x32x16 regs : %.3f ms  %.0f TFLOP/s\n", t, flops / t / 1e9);
    t = run<1, false>(src, d, iters); printf("16x16x32 regs : %.3f ms  %.0f TFLOP/s\n", t, flops / t / 1e9);
    t = run<0, true>(src, d, iters);  printf("32x32x16 +LDS : %.3f ms  %.0f TFLOP/s\n", t, flops / t / 1e9);
    t = run<1, true>(src, d, iters);  printf("16x16x32 +LDS : %.3f ms  %.0f TFLOP/s\n", t, flops / t / 1e9);
  }
  return 0;
}
