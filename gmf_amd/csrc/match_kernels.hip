// Descriptor-space nearest-neighbour matching: the step that PRODUCES the putative correspondences
// (SURVEY.md section 8 row f-2).  Replaces
//   GMF_PointDSC/datasets/ThreeDMatch.py:164-166, demo_registration.py:101-103
//       distance = sqrt(2 - 2 Fs Ft^T + 1e-6); source_idx = argmin(distance, axis=1)
//   GMF_DeepGlobalRegistration/*/core/knn.py:23-74 (find_knn_gpu, knn = 1) with core/metrics.py:62-69 (pdist)
//       argmin_j ||f0_i - f1_j||^2, chunked by nn_max_n rows to bound the [rows, N1, d] temporary.
// One fused pass: S^T = F1tile F0^T on the f32 MFMA (rows on lanes), the per-key score ||b_j||^2 - 2 <a_i, b_j> and a
// running (min, argmin) per lane; nothing of size N0 x N1 is ever written.  The reported distance of the winner is then
// recomputed with the reference's own formula.
#include "mfma_core.hpp"
#include "launchers.hpp"

namespace gmf {

// KG = K/8 groups of the (padded) descriptor width K; tiles of 32 keys x K floats, TPS tiles per 16 KiB-or-less stage
template <int KG>
__global__ void __launch_bounds__(256, 2)
k_nn_match(const float* __restrict__ f0_img, const float* __restrict__ f1_img, const float* __restrict__ f1_norm2,
           const float* __restrict__ F0, const float* __restrict__ F1, int* __restrict__ idx_out,
           float* __restrict__ dist_out, int N0, int N1, int d, int mode) {
  constexpr int K = 8 * KG, KF = 4 * KG;
  constexpr int TPS = (4096 / (32 * K)) > 0 ? (4096 / (32 * K)) : 1;
  constexpr int kStage = TPS * 32 * K;
  __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles0 = (N0 + 31) / 32, tiles1 = (N1 + 31) / 32;
  const int tile_raw = blockIdx.x * 4 + wave;
  const bool active = tile_raw < tiles0;
  const int tile = active ? tile_raw : tiles0 - 1;

  float a[KF];
  load_frag_p32<KF>(a, f0_img + (size_t)tile * (32 * K), lane);

  const int stages = (tiles1 + TPS - 1) / TPS;
  StageStream ss;
  ss.stage_floats = kStage;
  ss.init(lds, lds + kStage, wave, 4, lane, f1_img, stages);
  ss.prime();
  float best = INFINITY;
  int bidx = 0x7fffffff;
  for (int st = 0; st < stages; ++st) {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int t = st * TPS + tt;
      if (t < tiles1) {
        f32x16 acc = zero16();
        mma_wx<KF>(acc, lw + tt * (32 * K / 4), a);
        const int jbase = t * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = jbase + 8 * (r >> 2) + (r & 3);
          const float sc = (j < N1) ? fmaf(-2.0f, acc[r], f1_norm2[j]) : INFINITY;
          if (sc < best) { best = sc; bidx = j; }
        }
      }
    }
  }
  {
    const float ov = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64);
    if (ov < best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  const int row = tile * 32 + i;
  if (active && row < N0 && h == 0) {
    const float* pa = F0 + (size_t)row * d;
    const float* pb = F1 + (size_t)bidx * d;
    float out;
    if (mode == 0) {            // PointDSC: sqrt(2 - 2 <a,b> + 1e-6)
      float dot = 0.f;
      for (int k = 0; k < d; ++k) dot = fmaf(pa[k], pb[k], dot);
      out = sqrtf(2.0f - 2.0f * dot + 1e-6f);
    } else {                    // DGR: mode 1 = sqrt(sum (a-b)^2 + 1e-7) (pdist 'L2'), mode 2 = sum (a-b)^2 ('SquareL2')
      float d2 = 0.f;
      for (int k = 0; k < d; ++k) { const float df = pa[k] - pb[k]; d2 = fmaf(df, df, d2); }
      out = (mode == 1) ? sqrtf(d2 + 1e-7f) : d2;
    }
    idx_out[row] = bidx;
    dist_out[row] = out;
  }
}

// squared norms of the rows of F1 (mode 0 uses a constant: for unit descriptors argmin distance = argmax dot)
__global__ void k_row_norm2(const float* __restrict__ F, float* __restrict__ n2, int N, int d, int unit) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  float s = 0.f;
  if (!unit) for (int k = 0; k < d; ++k) s = fmaf(F[(size_t)r * d + k], F[(size_t)r * d + k], s);
  n2[r] = s;
}

// row-major [N, d] -> P32 image with K = padded width (extra columns zero)
__global__ void k_pack_desc(const float* __restrict__ src, float* __restrict__ dst, int N, int d, int K, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int lane = idx & 63;
  const long gi = idx >> 6;
  const int kg = K / 8;
  const int g = gi % kg;
  const long tile = gi / kg;
  const long row = tile * 32 + (lane & 31);
  const int k0 = 8 * g + 4 * (lane >> 5);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < N)
    for (int e = 0; e < 4; ++e) if (k0 + e < d) v[e] = src[row * d + k0 + e];
  reinterpret_cast<float4*>(dst)[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

int padded_desc_width(int d) { return d <= 32 ? 32 : d <= 40 ? 40 : d <= 64 ? 64 : d <= 128 ? 128 : -1; }

hipError_t launch_nn_match(const float* F0, const float* F1, float* f0_img, float* f1_img, float* norm2, int* idx,
                           float* dist, int N0, int N1, int d, int mode, hipStream_t s) {
  const int K = padded_desc_width(d);
  if (K < 0) return hipErrorInvalidValue;
  const int t0 = (N0 + 31) / 32, t1 = (N1 + 31) / 32;
  const long n40 = (long)t0 * (K / 8) * 64, n41 = (long)t1 * (K / 8) * 64;
  hipLaunchKernelGGL(k_pack_desc, dim3((unsigned)((n40 + 255) / 256)), dim3(256), 0, s, F0, f0_img, N0, d, K, n40);
  hipLaunchKernelGGL(k_pack_desc, dim3((unsigned)((n41 + 255) / 256)), dim3(256), 0, s, F1, f1_img, N1, d, K, n41);
  hipLaunchKernelGGL(k_row_norm2, dim3((N1 + 255) / 256), dim3(256), 0, s, F1, norm2, N1, d, mode == 0 ? 1 : 0);
  const dim3 grid((t0 + 3) / 4);
  switch (K) {
    case 32: hipLaunchKernelGGL(k_nn_match<4>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, F0, F1, idx, dist, N0, N1, d, mode); break;
    case 40: hipLaunchKernelGGL(k_nn_match<5>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, F0, F1, idx, dist, N0, N1, d, mode); break;
    case 64: hipLaunchKernelGGL(k_nn_match<8>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, F0, F1, idx, dist, N0, N1, d, mode); break;
    default: hipLaunchKernelGGL(k_nn_match<16>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, F0, F1, idx, dist, N0, N1, d, mode); break;
  }
  return hipGetLastError();
}

}  // namespace gmf
