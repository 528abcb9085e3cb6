// Checks the operand / result lane layout of v_mfma_f32_16x16x32_f16 on gfx950 against the convention the q16 attention
// kernel assumes:  A: lane (g = l>>4, c = l&15) holds A[c][8g..8g+7];  B: B[8g..8g+7][c];  D: D[4g + r][c], r = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, g = l >> 4, c = l & 15;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[c * 32 + 8 * g + e]; b[e] = (_Float16)B[(8 * g + e) * 16 + c]; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[r];
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  srand(3);
  for (auto& v : hA) v = (float)(rand() % 7 - 3);
  for (auto& v : hB) v = (float)(rand() % 5 - 2);
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
  float *dA, *dB, *dD;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("mfma_f32_16x16x32_f16 layout check: %d mismatches of 256\n", bad);
  return bad != 0;
}
