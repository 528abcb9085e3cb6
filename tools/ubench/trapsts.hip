// Does the wave's sticky exception field (TRAPSTS.EXCP, accumulated regardless of EXCP_EN) see an fp32 -> fp16 conversion that
// overflows, an inf - inf, and stay clear on benign arithmetic?   hipcc --offload-arch=gfx950 -O3 -o trapsts trapsts.hip && ./trapsts
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned read_excp(float dep = 0.f) {
  unsigned v;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(v) : "v"(dep) : "memory");   // (dep: keeps the read behind the arithmetic)
  return v;
}
__global__ void k(const float* in, float* out, unsigned* flags, int mode) {
  const unsigned before = read_excp();
  float x = in[threadIdx.x];
  float r = 0.f;
  if (mode == 0) {                       // benign: fma chain, exp2 of a negative number, max with -inf
    r = __builtin_amdgcn_exp2f(x * -3.f) + fmaxf(-INFINITY, x) * 0.5f;
  } else if (mode == 1) {                // cvt overflow: 1e6 -> fp16
    const f32x2 v = {x * 1e6f, x};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    r = (float)h[0] + (float)h[1];
  } else if (mode == 2) {                // inf - inf
    const float a = x * INFINITY;
    r = a - a;
  } else if (mode == 3) {                // exp2(-inf), -inf - finite, 0 * finite
    r = __builtin_amdgcn_exp2f(-INFINITY - x) + 0.f * x;
  } else if (mode == 4) {                // underflow / denormal products (benign for us)
    r = x * 1e-30f * 1e-30f;
  }
  out[threadIdx.x] = r;
  const unsigned after = read_excp(r);
  if (threadIdx.x == 0) { flags[2 * mode] = before; flags[2 * mode + 1] = after; }
}
int main() {
  float *in, *out; unsigned* fl;
  hipMalloc(&in, 256); hipMalloc(&out, 256); hipMalloc(&fl, 64);
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 0.01f;
  hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  hipMemset(fl, 0, 64);
  for (int m = 0; m < 5; ++m) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, fl, m);
  unsigned f[16]; hipMemcpy(f, fl, 64, hipMemcpyDeviceToHost);
  const char* names[5] = {"benign", "cvt f32->f16 overflow", "inf - inf", "exp2(-inf), 0 * x", "underflow"};
  for (int m = 0; m < 5; ++m) printf("%-24s EXCP before 0x%03x after 0x%03x\n", names[m], f[2 * m], f[2 * m + 1]);
  return 0;
}
