// Microbenchmark: fp32 MFMA forms at equal FLOPs on random data - 32x32x2, 16x16x4 and the 4-block 4x4x1 are the candidates
// for the training GEMM (k_gemm_lds).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x2 (4 accumulators of 16 registers), SHAPE 1: 16x16x4 (16 accumulators of 4 registers)
template <int SHAPE, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const float* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(blockIdx.x * 8 + j) * 64 + lane]; b[j] = src[((blockIdx.x + 7) * 8 + j) * 64 + lane]; }
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc32[q][r] = 0.f;
  for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) acc16[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (SHAPE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc32[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + q) & 7], acc32[q], 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc16[(q + 8 * (s & 1))] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[(s + q) & 7], acc16[(q + 8 * (s & 1))], 0, 0, 0);
      }
    }
  }
  float sres = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) sres += acc32[q][r];
  for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) sres += acc16[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = sres;
}

template <int SHAPE, int WPS>
float run(const float* src, float* d, int iters, int grid) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(grid), dim3(256), 0, 0, src, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(grid), dim3(256), 0, 0, src, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const size_t n = 1 << 22;
  std::vector<float> hsrc(n);
  srand(1);
  for (auto& v : hsrc) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
  float* src; float* d;
  (void)hipMalloc(&src, n * 4); (void)hipMalloc(&d, 512 * 256 * 4);
  (void)hipMemcpy(src, hsrc.data(), n * 4, hipMemcpyHostToDevice);
  const int iters = 4000;   // x 32 MFMAs of 32x32x2 (or 64 of 16x16x4) per wave
  for (int rep = 0; rep < 2; ++rep) {
    for (int grid : {256, 512}) {
      const double flops = (double)grid * 4 * iters * 32 * 4096.0;
      float t;
      t = run<0, 1>(src, d, iters, grid); printf("grid %d 32x32x2 : %.3f ms  %.1f TFLOP/s\n", grid, t, flops / t / 1e9);
      t = run<1, 1>(src, d, iters, grid); printf("grid %d 16x16x4 : %.3f ms  %.1f TFLOP/s\n", grid, t, flops / t / 1e9);
      t = run<0, 2>(src, d, iters, grid); printf("grid %d 32x32x2 (2 WG/CU): %.3f ms  %.1f TFLOP/s\n", grid, t, flops / t / 1e9);
    }
  }
  return 0;
}
