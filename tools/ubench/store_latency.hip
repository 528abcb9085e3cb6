// Round trip of a global_store_dwordx4 (1 KiB per wave) as the vmcnt counter sees it, and whether a LOAD issued after a store can retire before
// it: s_memtime around "4 stores; s_waitcnt vmcnt(0)" and around "4 stores; 1 load (L2-hot); s_waitcnt vmcnt(4)" (in-order retirement would make
// the second as slow as the first), with 256 workgroups (light) and 2048 (every CU busy storing).
// hipcc --offload-arch=gfx950 -O3 -o store_latency store_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int MODE>
__global__ void __launch_bounds__(256) k(f32x4* __restrict__ out, const f32x4* __restrict__ hot, int iters, long long* cyc, float* sink) {
  const int lane = threadIdx.x;
  f32x4* p = out + ((size_t)blockIdx.x * iters) * 4 * 256 + lane;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  long long total = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < 4; ++q) p[(it * 4 + q) * 256] = v;
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 r;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(hot + lane) : "memory");   // MODE 1: the reference (everything)
      if (MODE == 2) { }
      acc += r[0];
    }
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = total / iters;
  if (acc == 123.f) *sink = acc;
}
// MODE 3: load AFTER the stores, waited for with vmcnt(0) on the load only if retirement were out of order: use vmcnt(4) = "all but the 4 youngest":
// the load is the YOUNGEST, so vmcnt(0) is the only wait that covers it - instead issue the load FIRST in program order after the stores and wait
// vmcnt(0)...  The meaningful test: stores first, then load, then wait until the LOAD's data is usable (a dependent use).
__global__ void __launch_bounds__(256) k_dep(f32x4* __restrict__ out, const f32x4* __restrict__ hot, int iters, long long* cyc, float* sink) {
  const int lane = threadIdx.x;
  f32x4* p = out + ((size_t)blockIdx.x * iters) * 4 * 256 + lane;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  long long total = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < 4; ++q) p[(it * 4 + q) * 256] = v;
    const f32x4 r = hot[lane + 256 * (it & 3)];          // compiler: s_waitcnt vmcnt(0) before the use (the load is the youngest)
    acc += r[0];
    asm volatile("" :: "v"(acc));
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = total / iters;
  if (acc == 123.f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_load_only(const f32x4* __restrict__ hot, int iters, long long* cyc, float* sink) {
  const int lane = threadIdx.x;
  long long total = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const long long t0 = __builtin_readcyclecounter();
    const f32x4 r = hot[lane + 256 * (it & 3)];
    acc += r[0];
    asm volatile("" :: "v"(acc));
    const long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = total / iters;
  if (acc == 123.f) *sink = acc;
}
int main() {
  const int iters = 200;
  f32x4 *out, *hot; long long* cyc; float* sink;
  CK(hipMalloc(&out, (size_t)2048 * iters * 4 * 256 * 16)); CK(hipMalloc(&hot, 4096 * 16)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(hot, 0, 4096 * 16));
  for (int grid : {256, 2048}) {
    long long c0, c1, c2;
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, hot, iters, cyc, sink);
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, hot, iters, cyc, sink);
    CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_dep, dim3(grid), dim3(256), 0, 0, out, hot, iters, cyc, sink);
    hipLaunchKernelGGL(k_dep, dim3(grid), dim3(256), 0, 0, out, hot, iters, cyc, sink);
    CK(hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_load_only, dim3(grid), dim3(256), 0, 0, hot, iters, cyc, sink);
    hipLaunchKernelGGL(k_load_only, dim3(grid), dim3(256), 0, 0, hot, iters, cyc, sink);
    CK(hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost));
    printf("%4d workgroups: 4 stores of 1 KiB + vmcnt(0): %lld cycles | 4 stores, then an L2-hot load and its use: %lld cycles | the load and its use alone: %lld cycles\n",
           grid, c0, c1, c2);
  }
  return 0;
}
