// Backward of the pose head: the gradient of the transformation loss through the best
// hypothesis of each pair, and the DGR weighted-Procrustes gradient with respect to the correspondence weights.
//
// Replaces the autograd graph of (file:line relative to /root/reference/):
//   k_tl_backward          GMF_PointDSC/libs/loss.py:35-64          d TransformationLoss / d trans
//   k_pose_best_backward   GMF_PointDSC/models/PointDSC.py:330-365,405,421,437-448; models/common.py:10-50
//                          final_trans = seedwise_trans[argmax fitness]  ->  d / d normed features, d / d sigma
//   k_wp_backward          GMF_DeepGlobalRegistration/*/core/registration.py:91-113 (w from the inlier network,
//                          core/trainer.py:594-614)              d (R, t) / d w
//
// The 3x3 SVD is differentiated in closed form (kabsch.hpp): no singularity at equal singular values.
#include <hip/hip_runtime.h>
#include <math.h>

#include "launchers_pose.hpp"
#include "kabsch.hpp"

namespace gmf {

namespace {

GMF_DEVINL double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
GMF_DEVINL float wsumf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NV>
GMF_DEVINL void bsum(double (&v)[NV], double* sh /* >= NV*16 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wsum(v[k]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) sh[k * 16 + wave] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double s = 0;
    for (int w = 0; w < nw; ++w) s += sh[k * 16 + w];
    v[k] = s;
  }
}

constexpr int kKMax = 64;     // neighbours per seed (as k_seed_power)
constexpr int kCF = 128;      // feature width

}  // namespace

// ---------------------------------------------------------------------------------------
// loss = (1/bs) sum_i [any(probs_i > 0)] mean_{b',n} |R_i p_in + t_i - q_b'n|^2   (loss.py:47-48,57-62: pair i's warped
// source points against the target points of EVERY pair b' - the reference's broadcast).  g_trans [bs,4,4] = d loss / d trans
// (row 3 zero).  grid (bs), block 256.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tl_backward(const float* __restrict__ trans, const float* __restrict__ src, const float* __restrict__ tgt,
              const float* __restrict__ probs, float* __restrict__ g_trans, int B, int N) {
  __shared__ double sh[12 * 16];
  __shared__ int any_pos;
  const int pair = blockIdx.x;
  if (threadIdx.x == 0) any_pos = 0;
  __syncthreads();
  bool pos = false;
  for (int j = threadIdx.x; j < N; j += blockDim.x) pos = pos || (probs[(size_t)pair * N + j] > 0.f);
  if (pos) any_pos = 1;
  __syncthreads();
  const float* T = trans + (size_t)pair * 16;
  double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (any_pos) {
    const float* ps = src + (size_t)pair * N * 3;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const float x = ps[3 * j], y = ps[3 * j + 1], z = ps[3 * j + 2];
      const float wx = (T[0] * x + T[1] * y + T[2] * z) + T[3];
      const float wy = (T[4] * x + T[5] * y + T[6] * z) + T[7];
      const float wz = (T[8] * x + T[9] * y + T[10] * z) + T[11];
      double rx = 0, ry = 0, rz = 0;
      for (int b = 0; b < B; ++b) {
        const float* q = tgt + ((size_t)b * N + j) * 3;
        rx += (double)(wx - q[0]); ry += (double)(wy - q[1]); rz += (double)(wz - q[2]);
      }
      acc[0] += rx * x; acc[1] += rx * y; acc[2] += rx * z; acc[3] += rx;
      acc[4] += ry * x; acc[5] += ry * y; acc[6] += ry * z; acc[7] += ry;
      acc[8] += rz * x; acc[9] += rz * y; acc[10] += rz * z; acc[11] += rz;
    }
  }
  bsum<12>(acc, sh);
  if (threadIdx.x < 16) {
    const double scale = 2.0 / ((double)B * (double)N * (double)B);
    g_trans[(size_t)pair * 16 + threadIdx.x] = (threadIdx.x < 12) ? (float)(acc[threadIdx.x] * scale) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// Backward through the best hypothesis of one pair.  One wave per pair, lane a = neighbour a of the best seed.
//   forward (replayed):  G = F F^T over the k gathered unit features;  M = clamp(1-(1-G)/sigma^2, 0) * ms, diag 0
//   (ms = the spatial term, no gradient);  v_0 = 1, v_{i+1} = M v_i / (|M v_i| + 1e-6) for stop_it + 1 iterations (the
//   iterates come from k_seed_power's snapshots; stop_it from the pair's convergence flags as in k_seed_kabsch);
//   w = v / (sum v + 1e-6);  weighted Kabsch (fp64) -> R, t.
//   backward:  g_T -> (g_R, g_t) -> g_H (kabsch_backward) -> g_w -> g_v -> the power iterations in reverse -> g_M -> g_G,
//   g_sigma -> g_F rows, stored into g_feat (zero-filled by the caller; the best seed's neighbours are distinct rows).
// Dynamic LDS: F [64][129] | C [64][65] | V [(iters + 1)][64] | tmp [64] floats.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_pose_best_backward(const float* __restrict__ feat_n, const float* __restrict__ src, const float* __restrict__ tgt,
                     const int* __restrict__ knn_idx, const float* __restrict__ fitness, const float* __restrict__ snaps,
                     const unsigned char* __restrict__ conv, const float* __restrict__ g_T, float* __restrict__ g_feat,
                     float* __restrict__ g_sigma, int N, int S, int k, int iters, float sigma, float inv_sigmad2,
                     const int* __restrict__ stop_batch, const float* __restrict__ sigma_dev) {
  if (sigma_dev) sigma = *sigma_dev;                 // [r5] gmf_set_sigma_device
  extern __shared__ float dyn[];
  float* const F = dyn;                              // [64][129]
  float* const Cx = F + kKMax * (kCF + 1);           // [64][65]
  float* const Vs = Cx + kKMax * (kKMax + 1);        // [iters + 1][64]
  float* const tmp = Vs + (size_t)(iters + 1) * kKMax;
  __shared__ float P[kKMax * 8];
  const int pair = blockIdx.x, a = threadIdx.x;
  const bool valid = a < k;
  // best hypothesis: first maximal fitness (k_finalize_pose); stop iteration: first at which every seed of the pair passed
  // allclose (k_seed_kabsch)
  int best;
  {
    float bf = -1.f; int bi = 0x7fffffff;
    for (int s = a; s < S; s += 64) {
      const float f = fitness[(size_t)pair * S + s];
      if (f > bf) { bf = f; bi = s; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float of = __shfl_xor(bf, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (of > bf || (of == bf && oi < bi)) { bf = of; bi = oi; }
    }
    best = bi;
  }
  int stop_it = iters - 1;
  if (stop_batch) {                                  // batches: the stop iteration of the whole batch (k_stop_iteration)
    stop_it = *stop_batch;
  } else {
    const unsigned char* cv = conv + (size_t)pair * S * iters;
    for (int it = 0; it < iters - 1; ++it) {
      bool ok = true;
      for (int s2 = a; s2 < S; s2 += 64) ok = ok && (cv[(size_t)s2 * iters + it] != 0);
      if (__all(ok)) { stop_it = it; break; }
    }
  }
  const int* nb = knn_idx + ((size_t)pair * S + best) * k;
  const int my_row = valid ? nb[a] : 0;
  if (valid) {
    const float* ps = src + ((size_t)pair * N + my_row) * 3;
    const float* pt = tgt + ((size_t)pair * N + my_row) * 3;
    P[a * 8 + 0] = ps[0]; P[a * 8 + 1] = ps[1]; P[a * 8 + 2] = ps[2];
    P[a * 8 + 4] = pt[0]; P[a * 8 + 5] = pt[1]; P[a * 8 + 6] = pt[2];
  }
  for (int r = 0; r < k; ++r) {                       // coalesced row copies
    const float* row = feat_n + ((size_t)pair * N + nb[r]) * kCF;
    F[r * (kCF + 1) + a] = row[a];
    F[r * (kCF + 1) + 64 + a] = row[64 + a];
  }
  {
    const float* sn = snaps + ((size_t)pair * S + best) * iters * k;
    Vs[a] = 1.0f;
    for (int it = 0; it <= stop_it; ++it) Vs[(it + 1) * kKMax + a] = valid ? sn[it * k + a] : 0.f;
  }
  __syncthreads();

  const float inv_s2 = 1.0f / (sigma * sigma);
  float Mrow[kKMax], Grow[kKMax], msrow[kKMax];
  {
    float fa[kCF];
#pragma unroll
    for (int c = 0; c < kCF; ++c) fa[c] = valid ? F[a * (kCF + 1) + c] : 0.f;
#pragma unroll
    for (int b = 0; b < kKMax; ++b) {
      float g = 0.f, ms = 0.f, m = 0.f;
      if (b < k) {
#pragma unroll
        for (int c = 0; c < kCF; ++c) g = fmaf(fa[c], F[b * (kCF + 1) + c], g);
        if (valid && a != b) {
          const float ax = P[a * 8] - P[b * 8], ay = P[a * 8 + 1] - P[b * 8 + 1], az = P[a * 8 + 2] - P[b * 8 + 2];
          const float bx = P[a * 8 + 4] - P[b * 8 + 4], by = P[a * 8 + 5] - P[b * 8 + 5], bz = P[a * 8 + 6] - P[b * 8 + 6];
          const float d = sqrtf(ax * ax + ay * ay + az * az) - sqrtf(bx * bx + by * by + bz * bz);
          ms = fmaxf(1.0f - d * d * inv_sigmad2, 0.f);
          m = fmaxf(1.0f - (1.0f - g) * inv_s2, 0.f) * ms;
        }
      }
      Grow[b] = g; msrow[b] = ms; Mrow[b] = m;
    }
  }

  // ---- weights and the weighted Kabsch problem (fp64, every lane redundantly after the wave sums) ----
  const float* vfin = Vs + (stop_it + 1) * kKMax;
  const double va = valid ? (double)vfin[a] : 0.0;
  const double sv = (double)(float)wsum(va);
  const float inv_sv = 1.0f / ((float)sv + 1e-6f);
  float wf = valid ? vfin[a] * inv_sv : 0.f;
  const bool w_neg = wf < 0.f;
  if (w_neg) wf = 0.f;
  const double w = wf;
  double pa[3], pb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { pa[c] = valid ? P[a * 8 + c] : 0.0; pb[c] = valid ? P[a * 8 + 4 + c] : 0.0; }
  const double sw = wsum(w);
  double ca[3], cb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { ca[c] = wsum(w * pa[c]) / (sw + 1e-6); cb[c] = wsum(w * pb[c]) / (sw + 1e-6); }
  double am[3], bm[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { am[c] = pa[c] - ca[c]; bm[c] = pb[c] - cb[c]; }
  double H[9], Sa[3], Sb[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) H[3 * r + c] = wsum(w * am[r] * bm[c]);
    Sa[r] = wsum(w * am[r]);
    Sb[r] = wsum(w * bm[r]);
  }
  double R[9];
  kabsch_rotation_from_H(H, R);
  const float* gT = g_T + (size_t)pair * 16;
  double gR[9], gt[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    gt[r] = gT[4 * r + 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gR[3 * r + c] = (double)gT[4 * r + c] - gt[r] * ca[c];     // t = cb - R ca
  }
  double gH[9];
  kabsch_backward(H, gR, gH);
  double g_ca[3], g_cb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    g_ca[c] = -(R[0 + c] * gt[0] + R[3 + c] * gt[1] + R[6 + c] * gt[2]);
    g_cb[c] = gt[c];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { g_ca[r] -= gH[3 * r + c] * Sb[c]; g_cb[c] -= gH[3 * r + c] * Sa[r]; }
  double gw = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) gw += am[r] * gH[3 * r + c] * bm[c];
    gw += (g_ca[r] * am[r] + g_cb[r] * bm[r]) / (sw + 1e-6);
  }
  if (!valid || w_neg) gw = 0.0;
  // w = v / (sum v + 1e-6)
  const double gdot = wsum(gw * va);
  float gv = valid ? (float)(gw * (double)inv_sv - gdot * (double)inv_sv * (double)inv_sv) : 0.f;

  // ---- the power iterations in reverse ----
  float gMrow[kKMax];
#pragma unroll
  for (int b = 0; b < kKMax; ++b) gMrow[b] = 0.f;
  for (int it = stop_it; it >= 0; --it) {
    const float* vin = Vs + it * kKMax;
    float u = 0.f;
#pragma unroll
    for (int b = 0; b < kKMax; ++b) u = fmaf(Mrow[b], (b < k) ? vin[b] : 0.f, u);
    const float n = sqrtf(wsumf(u * u));
    const float dot = wsumf(gv * u);
    const float den = n + 1e-6f;
    const float gu = (n > 0.f) ? gv / den - u * dot / (den * den * n) : 0.f;
    __syncthreads();
    tmp[a] = valid ? gu : 0.f;
    __syncthreads();
    float gvin = 0.f;
#pragma unroll
    for (int b = 0; b < kKMax; ++b) {
      if (b < k) {
        gMrow[b] = fmaf(gu, vin[b], gMrow[b]);
        gvin = fmaf(Mrow[b], tmp[b], gvin);          // M is symmetric
      }
    }
    gv = valid ? gvin : 0.f;
  }

  // ---- M -> G, sigma ----
  double gs = 0.0;
  const float two_over_s3 = 2.0f / (sigma * sigma * sigma);
#pragma unroll
  for (int b = 0; b < kKMax; ++b) {
    float gG = 0.f;
    if (valid && b < k && a != b) {
      const float mf = 1.0f - (1.0f - Grow[b]) * inv_s2;
      if (mf > 0.f) {
        const float gmf = gMrow[b] * msrow[b];
        gG = gmf * inv_s2;
        gs += (double)(gmf * (1.0f - Grow[b]) * two_over_s3);
      }
    }
    if (b < k) Cx[a * (kKMax + 1) + b] = gG;
  }
  gs = wsum(gs);
  if (a == 0) g_sigma[pair] = (float)gs;
  __syncthreads();
  if (valid) {
    float gF[kCF];
#pragma unroll
    for (int c = 0; c < kCF; ++c) gF[c] = 0.f;
    for (int b = 0; b < k; ++b) {
      const float coef = Cx[a * (kKMax + 1) + b] + Cx[b * (kKMax + 1) + a];
#pragma unroll
      for (int c = 0; c < kCF; ++c) gF[c] = fmaf(coef, F[b * (kCF + 1) + c], gF[c]);
    }
    float4* out = reinterpret_cast<float4*>(g_feat + ((size_t)pair * N + my_row) * kCF);
#pragma unroll
    for (int c = 0; c < kCF; c += 4) out[c / 4] = make_float4(gF[c], gF[c + 1], gF[c + 2], gF[c + 3]);
  }
}

// ---------------------------------------------------------------------------------------
// DGR weighted_procrustes (k_weighted_procrustes) differentiated with respect to w.  grid (B), block 1024.
//   w~ = w/(sum|w| + eps);  mx = sum w~ x;  my = sum w~ y;  Sxy = sum (y-my)(w~ (x-mx))^T;  R = (Kabsch(Sxy))^T;  t = my - R mx
// g_R [B,9], g_t [B,3] -> g_w (ragged, same offsets as w).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_wp_backward(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ w,
              const int* __restrict__ offsets, float eps, const float* __restrict__ g_R, const float* __restrict__ g_t,
              float* __restrict__ g_w) {
  __shared__ double sh[9 * 16];
  const int pair = blockIdx.x;
  const int o0 = offsets[pair], n = offsets[pair + 1] - o0;
  const float* x = X + (size_t)o0 * 3;
  const float* y = Y + (size_t)o0 * 3;
  const float* ww = w + o0;
  float* gw = g_w + o0;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double wj = ww[j];
    acc[0] += fabs(wj);
    for (int c = 0; c < 3; ++c) { acc[1 + c] += wj * x[3 * j + c]; acc[4 + c] += wj * y[3 * j + c]; }
  }
  bsum<7>(acc, sh);
  const double inv = 1.0 / ((double)(float)acc[0] + (double)eps);
  const double mx[3] = {acc[1] * inv, acc[2] * inv, acc[3] * inv};
  const double my[3] = {acc[4] * inv, acc[5] * inv, acc[6] * inv};
  double Sm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double D[6] = {0, 0, 0, 0, 0, 0};                 // sum w~ (x - mx), sum w~ (y - my)
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double wn = ww[j] * inv;
    double xm[3], ym[3];
    for (int c = 0; c < 3; ++c) { xm[c] = x[3 * j + c] - mx[c]; ym[c] = y[3 * j + c] - my[c]; D[c] += wn * xm[c]; D[3 + c] += wn * ym[c]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Sm[3 * r + c] += ym[r] * (wn * xm[c]);
  }
  bsum<9>(Sm, sh);
  bsum<6>(D, sh);
  double Rt[9];
  kabsch_rotation_from_H(Sm, Rt);                     // R = Rt^T
  double gt[3], gRt[9];
  for (int r = 0; r < 3; ++r) gt[r] = g_t[(size_t)pair * 3 + r];
  // t = my - R mx:  g_R += -gt mx^T;  and d/dRt = (d/dR)^T
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) gRt[3 * c + r] = (double)g_R[(size_t)pair * 9 + 3 * r + c] - gt[r] * mx[c];
  double gS[9];
  kabsch_backward(Sm, gRt, gS);
  double g_mx[3], g_my[3];
  for (int c = 0; c < 3; ++c) {
    g_my[c] = gt[c];
    g_mx[c] = -(Rt[3 * c + 0] * gt[0] + Rt[3 * c + 1] * gt[1] + Rt[3 * c + 2] * gt[2]);    // -(R^T gt)_c, R^T = Rt
  }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { g_my[r] -= gS[3 * r + c] * D[c]; g_mx[c] -= gS[3 * r + c] * D[3 + r]; }
  double dotw[1] = {0};
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    double xm[3], ym[3], g = 0;
    for (int c = 0; c < 3; ++c) { xm[c] = x[3 * j + c] - mx[c]; ym[c] = y[3 * j + c] - my[c]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g += ym[r] * gS[3 * r + c] * xm[c];
    for (int c = 0; c < 3; ++c) g += g_mx[c] * x[3 * j + c] + g_my[c] * y[3 * j + c];
    gw[j] = (float)g;                                 // d / d w~_j, finished below
    dotw[0] += g * (double)ww[j];
  }
  bsum<1>(dotw, sh);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double wj = ww[j];
    const double sgn = (wj > 0.0) ? 1.0 : ((wj < 0.0) ? -1.0 : 0.0);
    gw[j] = (float)((double)gw[j] * inv - sgn * dotw[0] * inv * inv);
  }
}

// ---- launchers -------------------------------------------------------------------------
hipError_t launch_tl_backward(const float* trans, const float* src, const float* tgt, const float* probs, float* g_trans,
                              int B, int N, hipStream_t s) {
  hipLaunchKernelGGL(k_tl_backward, dim3(B), dim3(256), 0, s, trans, src, tgt, probs, g_trans, B, N);
  return hipGetLastError();
}

hipError_t launch_pose_best_backward(const float* feat_n, const float* src, const float* tgt, const int* knn_idx,
                                     const float* fitness, const float* snaps, const unsigned char* conv, const float* g_T,
                                     float* g_feat, float* g_sigma, int B, int N, int S, int k, int iters, float sigma,
                                     float sigma_d, const int* stop_batch, hipStream_t s, const float* sigma_dev) {
  const size_t lds = ((size_t)kKMax * (kCF + 1) + (size_t)kKMax * (kKMax + 1) + (size_t)(iters + 1) * kKMax + kKMax) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_best_backward),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_pose_best_backward, dim3(B), dim3(64), lds, s, feat_n, src, tgt, knn_idx, fitness, snaps, conv, g_T,
                     g_feat, g_sigma, N, S, k, iters, sigma, 1.0f / (sigma_d * sigma_d), stop_batch, sigma_dev);
  return hipGetLastError();
}

hipError_t launch_wp_backward(const float* X, const float* Y, const float* w, const int* offsets, int B, float eps,
                              const float* g_R, const float* g_t, float* g_w, hipStream_t s) {
  hipLaunchKernelGGL(k_wp_backward, dim3(B), dim3(1024), 0, s, X, Y, w, offsets, eps, g_R, g_t, g_w);
  return hipGetLastError();
}

}  // namespace gmf
