// Semantics check of v_fma_mix_f32 as "fp16 half of a register times an f32 plus an f32" (the compat product of the
// throughput-mode attention, c_fp16 * s - m): hipcc -O2 --offload-arch=gfx950 fma_mix_mul.hip -o fma_mix_mul ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void k(const unsigned* packed, const float* s, const float* a, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned hu = packed[i];
  const float sv = s[i], av = a[i];
  float r0, r1, r2, r3;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hu), "v"(sv), "v"(av));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hu), "v"(sv), "v"(av));
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(hu), "v"(sv));
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(hu), "v"(sv));
  out[4 * i] = r0; out[4 * i + 1] = r1; out[4 * i + 2] = r2; out[4 * i + 3] = r3;
}

int main() {
  const int n = 4096;
  std::vector<unsigned> hp(n); std::vector<float> hs(n), ha(n), ho(4 * n);
  std::vector<_Float16> lo(n), hi(n);
  srand(3);
  for (int i = 0; i < n; ++i) {
    lo[i] = (_Float16)(rand() / (float)RAND_MAX); hi[i] = (_Float16)(rand() / (float)RAND_MAX);
    unsigned short l, h; __builtin_memcpy(&l, &lo[i], 2); __builtin_memcpy(&h, &hi[i], 2);
    hp[i] = (unsigned)l | ((unsigned)h << 16);
    hs[i] = (rand() / (float)RAND_MAX - 0.5f) * 20.f; ha[i] = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
  }
  unsigned* dp; float *ds, *da, *dout;
  (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&dout, 4 * n * 4);
  (void)hipMemcpy(dp, hp.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ds, hs.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, ds, da, dout, n);
  (void)hipMemcpy(ho.data(), dout, 4 * n * 4, hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float e0 = __builtin_fmaf((float)lo[i], hs[i], ha[i]), e1 = __builtin_fmaf((float)hi[i], hs[i], ha[i]);
    const float e2 = (float)lo[i] * hs[i], e3 = (float)hi[i] * hs[i];
    bad[0] += ho[4 * i] != e0; bad[1] += ho[4 * i + 1] != e1; bad[2] += ho[4 * i + 2] != e2; bad[3] += ho[4 * i + 3] != e3;
  }
  printf("fma_mix c_lo*s+a: %d bad | c_hi*s+a: %d bad | c_lo*s+0: %d bad | c_hi*s+0: %d bad  (of %d)\n", bad[0], bad[1], bad[2], bad[3], n);
  printf("sample: lo %f hi %f s %f a %f -> %f %f %f %f\n", (float)lo[0], (float)hi[0], hs[0], ha[0], ho[0], ho[1], ho[2], ho[3]);
  return 0;
}
