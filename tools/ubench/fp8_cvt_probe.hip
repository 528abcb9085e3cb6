// Semantics of the gfx950 fp8 (OCP e4m3) conversions used by the attention kernel's P planes and the V image:
// scaled conversions (is it x * scale or x / scale?), rounding, decode.   hipcc --offload-arch=gfx950 -O3 -o fp8_cvt_probe fp8_cvt_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float scale, unsigned* enc32, unsigned* enc16, unsigned* encn, float* dec, float* dech) {
  const int i = threadIdx.x;
  const float a = in[2 * i], b = in[2 * i + 1];
  i16x2 old = {0, 0};
  const i16x2 e = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, scale, false);
  enc32[i] = (unsigned short)e[0];
  const f16x2 hh = {(_Float16)a, (_Float16)b};
  const i16x2 e2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, hh, scale, false);
  enc16[i] = (unsigned short)e2[0];
  encn[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
  const f32x2 d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8((int)(unsigned short)e[0], scale, false);
  dec[2 * i] = d[0]; dec[2 * i + 1] = d[1];
  const f16x2 dh = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)(unsigned short)e[0], scale, false);
  dech[2 * i] = (float)dh[0]; dech[2 * i + 1] = (float)dh[1];
}
int main() {
  const float h[16] = {1.0f, 1.0625f, 1.07f, 1.19f, 300.f, 500.f, 0.001f, 0.0021f, 17.3f, -5.5f, 0.f, -0.f, 3.5e-3f, 240.f, 256.f, 272.f};
  float *in, *dec, *dech; unsigned *e32, *e16, *en;
  hipMalloc(&in, 64); hipMalloc(&dec, 64); hipMalloc(&dech, 64); hipMalloc(&e32, 32); hipMalloc(&e16, 32); hipMalloc(&en, 32);
  hipMemcpy(in, h, 64, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, in, scale, e32, e16, en, dec, dech);
    unsigned a[8], b[8], c[8]; float d[16], dh[16];
    hipMemcpy(a, e32, 32, hipMemcpyDeviceToHost); hipMemcpy(b, e16, 32, hipMemcpyDeviceToHost); hipMemcpy(c, en, 32, hipMemcpyDeviceToHost);
    hipMemcpy(d, dec, 64, hipMemcpyDeviceToHost); hipMemcpy(dh, dech, 64, hipMemcpyDeviceToHost);
    printf("scale %g\n", scale);
    for (int i = 0; i < 8; ++i)
      printf("  (%g, %g): scalef32_f32 %04x  scalef32_f16 %04x  unscaled %04x   decoded (same scale) f32 (%g, %g) f16 (%g, %g)\n", h[2 * i], h[2 * i + 1], a[i], b[i], c[i],
             d[2 * i], d[2 * i + 1], dh[2 * i], dh[2 * i + 1]);
  }
  return 0;
}
