// What does a grid-wide barrier cost inside one kernel on this 8-XCD part, against a kernel boundary?  (one
// persistent kernel per small-grid layer instead of three launches.)  G workgroups (one per CU), each iteration: write 16 KiB,
// barrier, read the 16 KiB another workgroup (on another XCD) wrote and check it.  Forms of the barrier:
//   0  kernel boundary: one launch per iteration (the product's form)
//   1  agent-scope release / acquire fences around one atomic counter (buffer_wbl2 + buffer_inv per workgroup)
//   2  no cache maintenance: the data is written and read past the L2s (sc0 sc1 stores / loads), the barrier is the counter alone
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWords = 4096;   // 16 KiB per workgroup and iteration

template <int FORM>
__device__ __forceinline__ void put(float* p, float v) {
  if (FORM == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <int FORM>
__device__ __forceinline__ float get(const float* p) {
  if (FORM == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

template <int FORM>
__device__ __forceinline__ void step_body(float* buf, int G, int it, int* bad) {
  float* mine = buf + ((size_t)(it & 1) * G + blockIdx.x) * kWords;
  for (int i = threadIdx.x; i < kWords; i += 256) put<FORM>(mine + i, (float)(it * 131 + blockIdx.x + i));
}
template <int FORM>
__device__ __forceinline__ void check_body(const float* buf, int G, int it, int* bad) {
  const int other = (blockIdx.x + 37) % G;
  const float* theirs = buf + ((size_t)(it & 1) * G + other) * kWords;
  int wrong = 0;
  for (int i = threadIdx.x; i < kWords; i += 256) wrong += get<FORM>(theirs + i) != (float)(it * 131 + other + i);
  if (wrong) atomicAdd(bad, wrong);
}

__global__ void k_boundary_a(float* buf, int G, int it, int* bad) { step_body<0>(buf, G, it, bad); }
__global__ void k_boundary_b(float* buf, int G, int it, int* bad) { check_body<0>(buf, G, it, bad); }

template <int FORM>
__global__ __launch_bounds__(256) void k_persistent(float* buf, int G, int iters, unsigned* counter, int* bad) {
  for (int it = 0; it < iters; ++it) {
    step_body<FORM>(buf, G, it, bad);
    // ---- barrier: every workgroup reaches it (the exit condition of the spin is a count all G workgroups contribute to;
    //      the grid is G <= #CUs workgroups of one wave group each, so all are resident)
    if (FORM == 2) __builtin_amdgcn_s_waitcnt(0);   // (stores past the caches: wait until they have left)
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FORM == 1) __atomic_thread_fence(__ATOMIC_RELEASE);            // agent scope by default in HIP: buffer_wbl2 sc1
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * (unsigned)G;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      if (FORM == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);            // buffer_inv sc1
    }
    __syncthreads();
    check_body<FORM>(buf, G, it, bad);
  }
}

int main(int argc, char** argv) {
  const int iters = 200;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int G = argc > 1 ? atoi(argv[1]) : prop.multiProcessorCount;
  float* buf; unsigned* counter; int* bad;
  CK(hipMalloc(&buf, (size_t)2 * G * kWords * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int form = 0; form < 3; ++form) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(counter, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      if (form == 0) {
        for (int it = 0; it < iters; ++it) {
          hipLaunchKernelGGL(k_boundary_a, dim3(G), dim3(256), 0, 0, buf, G, it, bad);
          hipLaunchKernelGGL(k_boundary_b, dim3(G), dim3(256), 0, 0, buf, G, it, bad);
        }
      } else if (form == 1) {
        hipLaunchKernelGGL(k_persistent<1>, dim3(G), dim3(256), 0, 0, buf, G, iters, counter, bad);
      } else {
        hipLaunchKernelGGL(k_persistent<2>, dim3(G), dim3(256), 0, 0, buf, G, iters, counter, bad);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int hbad; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      if (rep == 1)
        printf("form %d (%s): %.2f us per write+barrier+check iteration (form 0: two kernel boundaries per iteration), %d workgroups, wrong words %d\n", form,
               form == 0 ? "two launches" : form == 1 ? "release/acquire fences" : "cache-bypassing data", ms * 1000.f / iters, G, hbad);
    }
  }
  return 0;
}
