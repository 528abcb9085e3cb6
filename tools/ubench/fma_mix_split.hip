#include <hip/hip_runtime.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, unsigned* outh) {
  float x0 = in[2 * threadIdx.x], x1 = in[2 * threadIdx.x + 1];
  const f32x2 x = {x0, x1};
  const f16x2 hh = __builtin_convertvector(x, f16x2);
  const unsigned hu = __builtin_bit_cast(unsigned, hh);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hu), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hu), "v"(x1));
  const f32x2 r = {r0, r1};
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  out[2 * threadIdx.x] = r0; out[2 * threadIdx.x + 1] = r1;
  outh[threadIdx.x] = __builtin_bit_cast(unsigned, ll);
}
#include <cstdio>
#include <cmath>
int main() {
  float h[128], o[128]; unsigned oh[64];
  for (int i = 0; i < 128; ++i) h[i] = (float)(sin(i * 1.37) * 300.0 + 0.001 * i);
  float *d, *e; unsigned* f;
  hipMalloc(&d, 512); hipMalloc(&e, 512); hipMalloc(&f, 256);
  hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, f);
  hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) { _Float16 hi = (_Float16)h[i]; float ref = h[i] - (float)hi; bad += (o[i] != ref); }
  printf("fma_mix split check: %d mismatches of 128\n", bad);
  return bad;
}
