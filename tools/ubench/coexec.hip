// Microbenchmark: how do bf16 / f32 MFMAs and VALU work share a SIMD on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: MFMA only (dependent chain), 1: VALU only, 2: fine interleave (V valu per mfma), 3: phases (all mfma then all valu)
template <int MODE, int V, bool F32>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters) {
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.01f); }
  float fa = threadIdx.x * 0.001f, fb = 0.5f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      if (MODE != 1) {
        if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int q = 0; q < V; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
      }
      if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < 48 * V; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// even workgroups: MFMA only; odd workgroups: VALU only (V per slot).  With 2 workgroups per CU every SIMD
// hosts one wave of each kind.
template <int V, bool F32>
__global__ void __launch_bounds__(256, 2) k_mixed(float* out, int iters) {
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.01f); }
  float fa = threadIdx.x * 0.001f, fb = 0.5f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
  if (blockIdx.x < 256) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 48; ++m) {
        if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 48 * V; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, bool F32>
float run_mixed(float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mixed<V, F32>), dim3(512), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_mixed<V, F32>), dim3(512), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int MODE, int V, bool F32>
float run(float* d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, V, F32>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, V, F32>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d;
  hipMalloc(&d, 4096 * 256 * 4);
  const int iters = 2000;
  for (int occ = 1; occ <= 2; ++occ) {
    const int blocks = 256 * occ;   // occ workgroups of 4 waves per CU -> occ waves per SIMD
    printf("== %d wave(s) per SIMD, %d x 48 MFMAs per wave ==\n", occ, iters);
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (iters * 48.0 * occ); };   // cycles per MFMA slot per SIMD at 2.4 GHz nominal
    float t;
    t = run<0, 0, false>(d, blocks, iters); printf("bf16 MFMA only            %.3f ms  (%.1f cyc/MFMA/SIMD)\n", t, cyc(t));
    t = run<1, 5, false>(d, blocks, iters); printf("VALU only, 5 per slot     %.3f ms  (%.1f cyc/slot)\n", t, cyc(t));
    t = run<2, 5, false>(d, blocks, iters); printf("bf16 + 5 VALU interleaved %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<3, 5, false>(d, blocks, iters); printf("bf16 then 5 VALU phases   %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<2, 2, false>(d, blocks, iters); printf("bf16 + 2 VALU interleaved %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<2, 8, false>(d, blocks, iters); printf("bf16 + 8 VALU interleaved %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<3, 8, false>(d, blocks, iters); printf("bf16 then 8 VALU phases   %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<1, 8, false>(d, blocks, iters); printf("VALU only, 8 per slot     %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<0, 0, true>(d, blocks, iters);  printf("f32 MFMA only             %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<2, 5, true>(d, blocks, iters);  printf("f32 + 5 VALU interleaved  %.3f ms  (%.1f)\n", t, cyc(t));
    t = run<3, 5, true>(d, blocks, iters);  printf("f32 then 5 VALU phases    %.3f ms  (%.1f)\n", t, cyc(t));
  }
  printf("== one MFMA-only wave + one VALU-only wave per SIMD (2000 x 48 slots each) ==\n");
  { float t;
    t = run_mixed<5, false>(d, iters);  printf("bf16 MFMA wave || VALU wave (5/slot)   %.3f ms\n", t);
    t = run_mixed<8, false>(d, iters);  printf("bf16 MFMA wave || VALU wave (8/slot)   %.3f ms\n", t);
    t = run_mixed<12, false>(d, iters); printf("bf16 MFMA wave || VALU wave (12/slot)  %.3f ms\n", t);
    t = run_mixed<5, true>(d, iters);   printf("f32  MFMA wave || VALU wave (5/slot)   %.3f ms\n", t);
    t = run_mixed<12, true>(d, iters);  printf("f32  MFMA wave || VALU wave (12/slot)  %.3f ms\n", t);
  }
  return 0;
}
