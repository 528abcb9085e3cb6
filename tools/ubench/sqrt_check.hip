// How often does a one-step-corrected v_sqrt_f32 differ from the correctly rounded sqrtf?  (GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__device__ __forceinline__ float sqrt_nr(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(fmaxf(x, 1e-36f));
  const float r = fmaf(-y, y, x);
  return fmaf(r, h, y);
}
__global__ void k(const float* x, unsigned long long* bad_nr, unsigned long long* bad_raw, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float ref = sqrtf(v);
  if (sqrt_nr(v) != ref) atomicAdd(bad_nr, 1ULL);
  if (__builtin_amdgcn_sqrtf(v) != ref) atomicAdd(bad_raw, 1ULL);
}
int main() {
  const int n = 1 << 26;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {   // squared distances of points in a 3 m box (and a few zeros / tiny values)
    float a = 3.f * rand() / RAND_MAX, b = 3.f * rand() / RAND_MAX, c = 3.f * rand() / RAND_MAX;
    h[i] = (i % 1000 == 0) ? 0.f : a * a + b * b + c * c * ((i % 7 == 0) ? 1e-4f : 1.f);
  }
  float* d; unsigned long long *b1, *b2;
  hipMalloc(&d, n * 4); hipMalloc(&b1, 8); hipMalloc(&b2, 8);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(b1, 0, 8); hipMemset(b2, 0, 8);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, b1, b2, n);
  unsigned long long r1, r2;
  hipMemcpy(&r1, b1, 8, hipMemcpyDeviceToHost); hipMemcpy(&r2, b2, 8, hipMemcpyDeviceToHost);
  printf("n = %d: corrected v_sqrt differs from sqrtf in %llu cases, raw v_sqrt in %llu cases\n", n, r1, r2);
  return 0;
}
