// Bit-exactness of the 3-instruction fp16 split (v_cvt_pk + v_fma_mixlo_f16 + v_fma_mixhi_f16) against the reference form
// lo = fp16(x - float(fp16(x))) over random magnitudes 2^-30 ... 2^15, signs, exact fp16 values and subnormal results.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/fma_mixlo_split.hip -o /tmp/mixlo && /tmp/mixlo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out_new, unsigned* out_ref, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float x0 = in[2 * t], x1 = in[2 * t + 1];
  const f32x2 x = {x0, x1};
  const f16x2 hh = __builtin_convertvector(x, f16x2);
  const unsigned hu = __builtin_bit_cast(unsigned, hh);
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hu), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hu), "v"(x1));
  const f32x2 r = {x0 - (float)hh[0], x1 - (float)hh[1]};
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  out_new[t] = d;
  out_ref[t] = __builtin_bit_cast(unsigned, ll);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> h(2 * n);
  srand(7);
  for (int i = 0; i < 2 * n; ++i) {
    const double m = 1.0 + (double)rand() / RAND_MAX;
    const int e = rand() % 46 - 30;
    float v = (float)(ldexp(m, e) * ((rand() & 1) ? 1 : -1));
    if (i % 97 == 0) v = (float)(_Float16)v;          // exactly representable: lo = 0
    if (i % 101 == 0) v = 0.0f;
    h[i] = v;
  }
  float* d; unsigned *a, *b;
  hipMalloc(&d, 8 * n); hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n);
  hipMemcpy(d, h.data(), 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, n);
  std::vector<unsigned> ha(n), hb(n);
  hipMemcpy(ha.data(), a, 4 * n, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, 4 * n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    // +0 and -0 low halves are the same number: compare as values
    for (int half = 0; half < 2; ++half) {
      const unsigned short u = (ha[i] >> (16 * half)) & 0xffff, v = (hb[i] >> (16 * half)) & 0xffff;
      if (u != v && !(((u | v) & 0x7fff) == 0)) { if (bad < 5) printf("mismatch at %d.%d: %04x vs %04x (x = %g)\n", i, half, u, v, h[2 * i + half]); ++bad; }
    }
  }
  printf("fma_mixlo split check: %d mismatches of %d\n", bad, 2 * n);
  return bad != 0;
}
