// Descriptor-space nearest-neighbour matching: the step that PRODUCES the putative correspondences
// (SURVEY.md section 8 row f-2).  Replaces
//   GMF_PointDSC/datasets/ThreeDMatch.py:164-166, demo_registration.py:101-103
//       distance = sqrt(2 - 2 Fs Ft^T + 1e-6); source_idx = argmin(distance, axis=1)
//   GMF_DeepGlobalRegistration/*/core/knn.py:23-74 (find_knn_gpu, knn = 1) with core/metrics.py:62-69 (pdist)
//       argmin_j ||f0_i - f1_j||^2, chunked by nn_max_n rows to bound the [rows, N1, d] temporary.
// One fused pass: S^T = F1tile F0^T on the f32 MFMA (rows on lanes), the per-key score ||b_j||^2 - 2 <a_i, b_j> and a
// running (min, argmin) per lane; nothing of size N0 x N1 is ever written.  The reported distance of the winner is then
// recomputed with the reference's own formula.
#include <algorithm>
#include "mfma_core.hpp"
#include "launchers.hpp"

namespace gmf {

// KG = K/8 groups of the (padded) descriptor width K; tiles of 32 keys x K floats, TPS tiles per 16 KiB-or-less stage.
// [r4] grid (ceil(tiles0 / 4), KS): workgroup (x, ks) sweeps the key STAGES [stages * ks / KS, stages * (ks + 1) / KS) for its 128 query rows
// and folds its (score, index) into best[row] with ONE 64-bit atomic minimum per row - the score mapped to an order-preserving
// unsigned word in the high half, the key index in the low half, so that the minimum is the smallest score and, among equal scores,
// the smallest index: the reference's first argmin, whatever the order the workgroups arrive in.  Until round 4 a query block swept
// all keys alone: 63 workgroups for 8000 x 8000 descriptors, 556 us for 41 us of matrix work.
template <int KG>
__global__ void __launch_bounds__(256, 2)
k_nn_match(const float* __restrict__ f0_img, const float* __restrict__ f1_img, const float* __restrict__ f1_norm2,
           unsigned long long* __restrict__ best_out, int N0, int N1) {
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

  const int stages_all = (tiles1 + TPS - 1) / TPS;
  const int s0 = (int)(((long)stages_all * blockIdx.y) / gridDim.y), s1 = (int)(((long)stages_all * (blockIdx.y + 1)) / gridDim.y);
  const int stages = s1 - s0;
  StageStream ss;
  ss.stage_floats = kStage;
  ss.init(lds, lds + kStage, wave, 4, lane, f1_img + (size_t)s0 * kStage, stages);
  ss.prime();
  float best = INFINITY;
  int bidx = 0x7fffffff;
  for (int st = 0; st < stages; ++st) {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int t = (s0 + st) * TPS + tt;
      if (t < tiles1) {
        // the norms of this lane's 16 keys (4 h + 8 g .. + 3 of the tile) as four 16-byte loads, requested before the MFMAs; keys
        // beyond N1 carry +inf (k_match_prep pads the array to whole tiles), so they never win and need no index test
        const int jbase = t * 32 + 4 * h;
        const float4* np = reinterpret_cast<const float4*>(f1_norm2 + jbase);
        const float4 n0 = np[0], n1 = np[2], n2 = np[4], n3 = np[6];
        const float nn[16] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
        f32x16 acc = zero16();
        mma_wx<KF>(acc, lw + tt * (32 * K / 4), a);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = jbase + 8 * (r >> 2) + (r & 3);
          const float sc = fmaf(-2.0f, acc[r], nn[r]);
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
  if (active && row < N0 && h == 0 && bidx != 0x7fffffff) {
    unsigned u = __float_as_uint(best + 0.0f);                   // (-0 -> +0: the two compare equal as floats)
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);             // order-preserving map of the finite floats and +-inf
    atomicMin(best_out + row, ((unsigned long long)u << 32) | (unsigned)bidx);
  }
}

// the winner of every row and its distance by the reference's own formula.  A row whose scores were all NaN has no winner: index 0
// (an in-range row; the reported distance is then NaN too).
__global__ void __launch_bounds__(256)
k_nn_finish(const unsigned long long* __restrict__ best, const float* __restrict__ F0, const float* __restrict__ F1,
            int* __restrict__ idx_out, float* __restrict__ dist_out, int N0, int N1, int d, int mode) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N0) return;
  unsigned bi = (unsigned)(best[row] & 0xffffffffull);
  const int bidx = (bi < (unsigned)N1) ? (int)bi : 0;
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

// [r5] Everything k_nn_match needs, in ONE launch of four workgroup ranges (it was four launches - two packing kernels, the norms and
// a 9 us fill kernel behind hipMemsetAsync - in front of a 22 us matching kernel):
//   [0, b0)   F0 row-major [N0, d] -> P32 image with K = padded width (extra columns zero)
//   [b0, b1)  F1 likewise
//   [b1, b2)  squared norms of the rows of F1 (mode 0 uses a constant: for unit descriptors argmin distance = argmax dot); n2 holds
//             whole tiles: entries N1 .. 32 * ceil(N1 / 32) - 1 are +inf (a key that does not exist never wins, k_nn_match)
//   [b2, ..)  best[row] = all ones (the identity of the 64-bit atomic minimum)
GMF_DEVINL void pack_desc(const long idx, const float* __restrict__ src, float* __restrict__ dst, int N, int d, int K, long total4) {
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

__global__ void k_match_prep(const float* __restrict__ F0, const float* __restrict__ F1, float* __restrict__ f0_img,
                             float* __restrict__ f1_img, float* __restrict__ n2, unsigned long long* __restrict__ best, int N0, int N1,
                             int d, int K, long n40, long n41, int unit, int b0, int b1, int b2) {
  const int blk = blockIdx.x;
  if (blk < b0) { pack_desc((long)blk * 256 + threadIdx.x, F0, f0_img, N0, d, K, n40); return; }
  if (blk < b1) { pack_desc((long)(blk - b0) * 256 + threadIdx.x, F1, f1_img, N1, d, K, n41); return; }
  if (blk < b2) {
    const int r = (blk - b1) * 256 + threadIdx.x;
    if (r >= ((N1 + 31) & ~31)) return;
    if (r >= N1) { n2[r] = INFINITY; return; }
    float s = 0.f;
    if (!unit) for (int k = 0; k < d; ++k) s = fmaf(F1[(size_t)r * d + k], F1[(size_t)r * d + k], s);
    n2[r] = s;
    return;
  }
  const int r = (blk - b2) * 256 + threadIdx.x;
  if (r < N0) best[r] = ~0ull;
}

int padded_desc_width(int d) { return d <= 32 ? 32 : d <= 40 ? 40 : d <= 64 ? 64 : d <= 128 ? 128 : -1; }

hipError_t launch_nn_match(const float* F0, const float* F1, float* f0_img, float* f1_img, float* norm2, unsigned long long* best,
                           int* idx, float* dist, int N0, int N1, int d, int mode, hipStream_t s) {
  const int K = padded_desc_width(d);
  if (K < 0) return hipErrorInvalidValue;
  const int t0 = (N0 + 31) / 32, t1 = (N1 + 31) / 32;
  const long n40 = (long)t0 * (K / 8) * 64, n41 = (long)t1 * (K / 8) * 64;
  {
    const int b0 = (int)((n40 + 255) / 256), b1 = b0 + (int)((n41 + 255) / 256), b2 = b1 + (t1 * 32 + 255) / 256, b3 = b2 + (N0 + 255) / 256;
    hipLaunchKernelGGL(k_match_prep, dim3(b3), dim3(256), 0, s, F0, F1, f0_img, f1_img, norm2, best, N0, N1, d, K, n40, n41,
                       mode == 0 ? 1 : 0, b0, b1, b2);
  }
  // key splits: about four workgroups per CU in all (two are resident), every split at least two stages long
  const int tps = (4096 / (32 * K)) > 0 ? (4096 / (32 * K)) : 1;
  const int stages = (t1 + tps - 1) / tps, wg0 = (t0 + 3) / 4;
  int ks = (1024 + wg0 - 1) / wg0;
  ks = std::max(1, std::min(ks, stages / 2));
  const dim3 grid(wg0, ks);
  switch (K) {
    case 32: hipLaunchKernelGGL(k_nn_match<4>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, best, N0, N1); break;
    case 40: hipLaunchKernelGGL(k_nn_match<5>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, best, N0, N1); break;
    case 64: hipLaunchKernelGGL(k_nn_match<8>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, best, N0, N1); break;
    default: hipLaunchKernelGGL(k_nn_match<16>, grid, dim3(256), 0, s, f0_img, f1_img, norm2, best, N0, N1); break;
  }
  hipLaunchKernelGGL(k_nn_finish, dim3((N0 + 255) / 256), dim3(256), 0, s, best, F0, F1, idx, dist, N0, N1, d, mode);
  return hipGetLastError();
}

}  // namespace gmf
