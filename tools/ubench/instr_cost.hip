// Issue cost of the instructions the attention tile is made of, for ONE wave per SIMD: loop body = one 32x32x16 f16 MFMA (34.5 cycles alone)
// + 8 copies of the instruction under test; a wave's own vector work does not overlap its MFMA (mfma_valu_overlap.hip), so
// (period - 34.5) / 8 is what one such instruction costs the wave.    hipcc --offload-arch=gfx950 -O3 -o instr_cost instr_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int OP>
__global__ void __launch_bounds__(256, 1) k(const f16x8* __restrict__ src, float* out, int iters, long long* cyc) {
  __shared__ f32x4 sh[1024];
  const int lane = threadIdx.x & 63;
  const f16x8 a = src[blockIdx.x * 64 + lane], b = src[(blockIdx.x + 7) * 64 + lane];
  for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i] * 0.01f + 0.001f * i;
  const float m = (float)b[0] * 1e-3f + 1.0f;
  const f16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float x = v[i];
        if (OP == 1) x = __builtin_fmaf(x, m, 0.5f);
        if (OP == 2) x = __builtin_amdgcn_exp2f(x * 1e-3f);                                  // (mul + exp: subtract the fma row)
        if (OP == 3) { const f32x2 p = {x, m}; const f16x2 hh = __builtin_convertvector(p, f16x2); x = (float)hh[0] + (float)hh[1]; }   // cvt_pk + 2 cvt back + add
        if (OP == 4) { i16x2 o = {0, 0}; o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, f16x2{(_Float16)x, (_Float16)m}, 4.0f, false); x = (float)o[0]; }
        if (OP == 5) { const f16x2 d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(__float_as_int(x), 0.5f, false); x = (float)d[0]; }
        if (OP == 6) x = __builtin_amdgcn_fdot2(f16x2{(_Float16)x, (_Float16)m}, ones, x, false);
        if (OP == 7) x = __builtin_fmaxf(x, m) * 0.999f;
        if (OP == 8) { const f32x4 t = sh[(lane + 64 * i + u) & 1023]; x += t[0]; }             // ds_read_b128 + add (latency exposed: dependent)
        if (OP == 9) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = __uint_as_float(r[0]) * 0.5f; }
        v[i] = x;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
void run(const char* name, const f16x8* src, float* d, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<OP>, dim3(32), dim3(256), 0, 0, src, d, iters, cyc);
  hipLaunchKernelGGL(k<OP>, dim3(32), dim3(256), 0, 0, src, d, iters, cyc);
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per = (double)c / (iters * 8.0);
  printf("  %-72s period %.1f  => %.1f cycles per copy\n", name, per, (per - 34.5) / 8.0);
}
int main() {
  const size_t n = 1 << 20;
  std::vector<_Float16> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
  f16x8* src; float* d; long long* cyc;
  CK(hipMalloc(&src, n * 2)); CK(hipMalloc(&d, 64 * 256 * 4)); CK(hipMalloc(&cyc, 8));
  CK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
  run<0>("nothing (MFMA alone)", src, d, cyc);
  run<1>("v_fma_f32", src, d, cyc);
  run<2>("v_mul_f32 + v_exp_f32", src, d, cyc);
  run<3>("v_cvt_pk_f16_f32 + 2 x v_cvt_f32_f16 + v_add", src, d, cyc);
  run<4>("(cvt f16, pack) + v_cvt_scalef32_pk_fp8_f16 + cvt back", src, d, cyc);
  run<5>("v_cvt_scalef32_pk_f16_fp8 + cvt back", src, d, cyc);
  run<6>("(cvt f16, pack) + v_dot2_f32_f16", src, d, cyc);
  run<7>("v_max_f32 + v_mul_f32", src, d, cyc);
  run<8>("ds_read_b128 + dependent v_add (latency exposed)", src, d, cyc);
  run<9>("v_permlane32_swap + v_mul", src, d, cyc);
  return 0;
}
