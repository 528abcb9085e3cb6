// Training primitives (second backward slice): what forward + backward of one FusionLayer
// (GMF_PointDSC/models/fusion_layer.py:32-128,172-201 - LCPE, PreNorm, Attention, GEGLU FeedForward) need, as plain
// row-major fp32 kernels that a torch.autograd.Function composes (gmf_amd/train.py):
//   k_gemm_f32          C = alpha op(A) op(B) + bias + R, batched, optional deterministic split-K      (every Linear, QK^T, PV
//                       and their gradient products: dX = dY W, dW = dY^T X with K = all rows of the batch)
//   k_lcpe_fwd / _bwd   depthwise k = 3 convolution + identity along the token axis                    fusion_layer.py:118-128
//   k_ln_fwd / _bwd     LayerNorm (eps 1e-5, biased variance) with saved mean / rstd                   fusion_layer.py:32-52
//   k_softmax_fwd/_bwd  row softmax of scale * S                                                       fusion_layer.py:90-91
//   k_geglu_fwd / _bwd  x * gelu_erf(gates)                                                            fusion_layer.py:54-57
//   k_colsum_*          per-column sums of x or x * y over all rows (bias, LayerNorm and LCPE-tap gradients), two passes in
//                       a fixed order: results do not depend on scheduling
// Unlike the inference kernels these are NOT fused: the saved activations go through HBM (1.5 GB per layer at 32 x 5000,
// which 288 GB make a non-issue) and the contractions run on the fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, no
// operand-range caveats for gradients of magnitude 1e-8) - a first, correct, device-resident training slice.
#include <algorithm>

#include "mfma_core.hpp"
#include "launchers.hpp"

namespace gmf {

// =========================================================================================
// k_gemm_f32: C[b] = alpha * op(A[b]) op(B[b]) (+ bias[col]) (+ R[b]),  op(X) = X or X^T (TA / TB), all row-major fp32.
//   op(A) is M x K: TA = false: A[m * lda + k]; TA = true: A[k * lda + m].   op(B) is K x N: TB = false: B[k * ldb + n];
//   TB = true: B[n * ldb + k].   A workgroup owns a 128 x 128 tile of C, a wave a 64 x 64 quarter (2 x 2 MFMA blocks).
//   Operands go global -> registers directly: lane (h, i) holds for its row (column) i the 8 contraction indices
//   k0 + 8 h .. + 7 of a 16-wide k-step; MFMA step e contracts (k0 + e, k0 + 8 + e) - any pairing is right as long as both
//   operands use it.  D = mfma(A rows, B columns): lane (h, j) register r = C[8 (r >> 2) + 4 h + (r & 3)][j].
//   ksplits > 1: workgroup z handles the k range [ks * kchunk, (ks + 1) * kchunk) and writes its un-scaled partial tile to
//   part[(b * ksplits + ks)][M][N]; k_gemm_reduce adds them in index order and applies alpha / bias / R.
//   grid (ceil(N / 128), ceil(M / 128), batch * ksplits), block 256.
// =========================================================================================
template <bool TA, bool TB>
__global__ void __launch_bounds__(256)
k_gemm_f32(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, const float* __restrict__ bias,
           const float* __restrict__ R, int M, int N, int K, long lda, long ldb, long ldc, long sA, long sB, long sC,
           int ksplits, int kchunk, float alpha, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int b = blockIdx.z / ksplits, ks = blockIdx.z - b * ksplits;
  const int kbeg = ks * kchunk, kend = min(K, kbeg + kchunk);
  const int m0 = blockIdx.y * 128 + 64 * wr, n0 = blockIdx.x * 128 + 64 * wc;
  A += (size_t)b * sA;
  Bm += (size_t)b * sB;
  f32x16 acc[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = zero16();
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    const int kk = k0 + 8 * h;
    float a[2][8], bf[2][8];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int r = m0 + 32 * rb + i;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kk + e;
        const bool ok = (r < M) && (k < kend);
        a[rb][e] = ok ? (TA ? A[(size_t)k * lda + r] : A[(size_t)r * lda + k]) : 0.f;
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int c = n0 + 32 * cb + i;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kk + e;
        const bool ok = (c < N) && (k < kend);
        bf[cb][e] = ok ? (TB ? Bm[(size_t)c * ldb + k] : Bm[(size_t)k * ldb + c]) : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma32(a[rb][e], bf[cb][e], acc[rb][cb]);
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = n0 + 32 * cb + i;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * rb + 8 * (r >> 2) + 4 * h + (r & 3);
        if (row >= M) continue;
        if (ksplits > 1) {
          part[((size_t)blockIdx.z * M + row) * N + col] = acc[rb][cb][r];
        } else {
          float v = alpha * acc[rb][cb][r];
          if (bias) v += bias[col];
          if (R) v += R[(size_t)b * sC + (size_t)row * ldc + col];
          C[(size_t)b * sC + (size_t)row * ldc + col] = v;
        }
      }
    }
}

__global__ void __launch_bounds__(256)
k_gemm_reduce(const float* __restrict__ part, float* __restrict__ C, const float* __restrict__ bias, const float* __restrict__ R,
              int M, int N, long ldc, long sC, int ksplits, float alpha, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long mn = (long)M * N;
  const int b = (int)(idx / mn);
  const long rem = idx - (long)b * mn;
  const int row = (int)(rem / N), col = (int)(rem - (long)row * N);
  float s = 0.f;
  for (int z = 0; z < ksplits; ++z) s += part[((size_t)(b * ksplits + z)) * mn + rem];
  float v = alpha * s;
  if (bias) v += bias[col];
  if (R) v += R[(size_t)b * sC + (size_t)row * ldc + col];
  C[(size_t)b * sC + (size_t)row * ldc + col] = v;
}

// =========================================================================================
// LCPE (fusion_layer.py:118-128): y[b,l,c] = x[l] + bias[c] + w[c][0] x[l-1] + w[c][1] x[l] + w[c][2] x[l+1], zero padded per
// sequence of L rows.  Backward for x: dx[l] = dy[l] (1 + w1) + w0 dy[l+1] + w2 dy[l-1]   (the taps transposed).
// =========================================================================================
__global__ void __launch_bounds__(256)
k_lcpe_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int L,
           int C, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long row = idx / C;
  const int l = (int)(row % L);
  const float xm = l > 0 ? x[idx - C] : 0.f, xc = x[idx], xp = l + 1 < L ? x[idx + C] : 0.f;
  y[idx] = xc + bias[c] + w[3 * c] * xm + w[3 * c + 1] * xc + w[3 * c + 2] * xp;
}

__global__ void __launch_bounds__(256)
k_lcpe_bwd(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int L, int C, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long row = idx / C;
  const int l = (int)(row % L);
  const float gm = l > 0 ? dy[idx - C] : 0.f, gc = dy[idx], gp = l + 1 < L ? dy[idx + C] : 0.f;
  dx[idx] = gc * (1.0f + w[3 * c + 1]) + w[3 * c] * gp + w[3 * c + 2] * gm;
}

// =========================================================================================
// LayerNorm over the last dim (C <= 1024, one wave per row).  Forward saves mean and rstd per row.
// Backward: g = dy * gamma ; dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dx_add when given: the residual path).
// =========================================================================================
GMF_DEVINL float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
GMF_DEVINL float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ void __launch_bounds__(256)
k_ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
         float* __restrict__ mean, float* __restrict__ rstd, long rows, int C) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mu = wave_sum_f(s) / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; v = fmaf(d, d, v); }
  const float rs = rsqrtf(wave_sum_f(v) / C + 1e-5f);
  for (int c = lane; c < C; c += 64) y[row * C + c] = fmaf((xr[c] - mu) * rs, gamma[c], beta[c]);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

__global__ void __launch_bounds__(256)
k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
         const float* __restrict__ rstd, const float* __restrict__ dx_add, float* __restrict__ dx, long rows, int C) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mu = mean[row], rs = rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float g = dy[row * C + c] * gamma[c], xh = (x[row * C + c] - mu) * rs;
    s1 += g;
    s2 = fmaf(g, xh, s2);
  }
  const float m1 = wave_sum_f(s1) / C, m2 = wave_sum_f(s2) / C;
  for (int c = lane; c < C; c += 64) {
    const float g = dy[row * C + c] * gamma[c], xh = (x[row * C + c] - mu) * rs;
    float v = rs * (g - m1 - xh * m2);
    if (dx_add) v += dx_add[row * C + c];
    dx[row * C + c] = v;
  }
}

// =========================================================================================
// Row softmax of scale * S (T <= 4096, one wave per row) and its backward dS = scale * P * (dP - sum_j dP_j P_j).
// =========================================================================================
__global__ void __launch_bounds__(256)
k_softmax_fwd(const float* __restrict__ S, float* __restrict__ P, long rows, int T, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* s = S + row * T;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) mx = fmaxf(mx, s[j] * scale);
  mx = wave_max_f(mx);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += expf(s[j] * scale - mx);
  const float inv = 1.0f / wave_sum_f(sum);
  for (int j = lane; j < T; j += 64) P[row * T + j] = expf(s[j] * scale - mx) * inv;
}

__global__ void __launch_bounds__(256)
k_softmax_bwd(const float* __restrict__ P, const float* __restrict__ dP, float* __restrict__ dS, long rows, int T, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float dot = 0.f;
  for (int j = lane; j < T; j += 64) dot = fmaf(P[row * T + j], dP[row * T + j], dot);
  dot = wave_sum_f(dot);
  for (int j = lane; j < T; j += 64) dS[row * T + j] = scale * P[row * T + j] * (dP[row * T + j] - dot);
}

// =========================================================================================
// GEGLU (fusion_layer.py:54-57): hdn [rows, 2 H] = (x | gates) -> g = x * gelu_erf(gates).
// Backward: dx = dg * gelu(gates) ; dgates = dg * x * (Phi(gates) + gates * phi(gates)).
// =========================================================================================
__global__ void __launch_bounds__(256)
k_geglu_fwd(const float* __restrict__ hdn, float* __restrict__ g, int H, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / H;
  const int c = (int)(idx - row * H);
  const float xv = hdn[row * 2 * H + c], gt = hdn[row * 2 * H + H + c];
  g[idx] = xv * (0.5f * gt * (1.0f + erff(gt * 0.70710678118654752440f)));
}

__global__ void __launch_bounds__(256)
k_geglu_bwd(const float* __restrict__ hdn, const float* __restrict__ dg, float* __restrict__ dh, int H, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / H;
  const int c = (int)(idx - row * H);
  const float xv = hdn[row * 2 * H + c], gt = hdn[row * 2 * H + H + c], d = dg[idx];
  const float Phi = 0.5f * (1.0f + erff(gt * 0.70710678118654752440f));
  const float phi = 0.3989422804014327f * expf(-0.5f * gt * gt);
  dh[row * 2 * H + c] = d * gt * Phi;
  dh[row * 2 * H + H + c] = d * xv * (Phi + gt * phi);
}

// =========================================================================================
// Column sums over all rows, deterministic (per-chunk partials, then one ordered pass):
//   out[c] = sum_r x[r][c] * y'[r + shift][c]      y' = y, or (y - mean[r']) * rstd[r'] when mean is given (LayerNorm xhat),
//                                                  or 1 when y is null; rows r + shift outside the sequence of L rows that
//                                                  contains r contribute 0 (the LCPE tap gradients, shift = -1, 0, +1).
//   grid (chunks), block 256: thread t handles columns t, t + 256, ...
// =========================================================================================
__global__ void __launch_bounds__(256)
k_colsum_partial(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ mean,
                 const float* __restrict__ rstd, int shift, int L, long rows, int C, int rows_per_chunk, float* __restrict__ part) {
  const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + (long)rows_per_chunk);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (long r = r0; r < r1; ++r) {
      float v = x[r * C + c];
      if (y) {
        const int l = (int)(r % L) + shift;
        if (l < 0 || l >= L) continue;
        const long ry = r + shift;
        float yv = y[ry * C + c];
        if (mean) yv = (yv - mean[ry]) * rstd[ry];
        v *= yv;
      }
      s += v;
    }
    part[(size_t)blockIdx.x * C + c] = s;
  }
}

__global__ void __launch_bounds__(256)
k_colsum_final(const float* __restrict__ part, int chunks, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[(size_t)k * C + c];
  out[c] = s;
}

// -----------------------------------------------------------------------------------------
static inline unsigned blocks_of(long total) { return (unsigned)((total + 255) / 256); }

int gemm_ksplits(int M, int N, int K, int batch) {
  // few output tiles and a long contraction (weight gradients: K = every row of the batch): split K so that the launch
  // has ~1000 workgroups; each split at least 512 deep
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (tiles >= 256 || K < 2048) return 1;
  int s = (int)std::min<long>((1024 + tiles - 1) / tiles, K / 512);
  return std::max(1, std::min(s, 256));
}

hipError_t launch_gemm_f32(bool ta, bool tb, const float* A, const float* B, float* C, const float* bias, const float* R, int M, int N,
                           int K, long lda, long ldb, long ldc, long sA, long sB, long sC, int batch, float alpha, float* part,
                           int ksplits, hipStream_t s) {
  if (!part) ksplits = 1;
  int kchunk = (K + ksplits - 1) / ksplits;
  kchunk = (kchunk + 15) / 16 * 16;
  ksplits = (K + kchunk - 1) / kchunk;
  const dim3 grid((N + 127) / 128, (M + 127) / 128, batch * ksplits);
#define GMF_GEMM(TA, TB) hipLaunchKernelGGL((k_gemm_f32<TA, TB>), grid, dim3(256), 0, s, A, B, C, bias, R, M, N, K, lda, ldb, ldc, sA, sB, sC, ksplits, kchunk, alpha, part)
  if (ta && tb) GMF_GEMM(true, true);
  else if (ta) GMF_GEMM(true, false);
  else if (tb) GMF_GEMM(false, true);
  else GMF_GEMM(false, false);
#undef GMF_GEMM
  if (ksplits > 1) {
    const long total = (long)batch * M * N;
    hipLaunchKernelGGL(k_gemm_reduce, dim3(blocks_of(total)), dim3(256), 0, s, part, C, bias, R, M, N, ldc, sC, ksplits, alpha, total);
  }
  return hipGetLastError();
}

hipError_t launch_lcpe(bool backward, const float* x, const float* w, const float* bias, float* y, int rows, int L, int C, hipStream_t s) {
  const long total = (long)rows * C;
  if (backward) hipLaunchKernelGGL(k_lcpe_bwd, dim3(blocks_of(total)), dim3(256), 0, s, x, w, y, L, C, total);
  else hipLaunchKernelGGL(k_lcpe_fwd, dim3(blocks_of(total)), dim3(256), 0, s, x, w, bias, y, L, C, total);
  return hipGetLastError();
}

hipError_t launch_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows, int C,
                         hipStream_t s) {
  hipLaunchKernelGGL(k_ln_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, C);
  return hipGetLastError();
}

hipError_t launch_ln_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dx_add,
                         float* dx, long rows, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx_add, dx, rows, C);
  return hipGetLastError();
}

hipError_t launch_softmax(bool backward, const float* a, const float* b, float* out, long rows, int T, float scale, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (backward) hipLaunchKernelGGL(k_softmax_bwd, grid, dim3(256), 0, s, a, b, out, rows, T, scale);
  else hipLaunchKernelGGL(k_softmax_fwd, grid, dim3(256), 0, s, a, out, rows, T, scale);
  return hipGetLastError();
}

hipError_t launch_geglu(bool backward, const float* hdn, const float* dg, float* out, long rows, int H, hipStream_t s) {
  const long total = rows * H;
  if (backward) hipLaunchKernelGGL(k_geglu_bwd, dim3(blocks_of(total)), dim3(256), 0, s, hdn, dg, out, H, total);
  else hipLaunchKernelGGL(k_geglu_fwd, dim3(blocks_of(total)), dim3(256), 0, s, hdn, out, H, total);
  return hipGetLastError();
}

int colsum_chunks(long rows) { return (int)std::max<long>(1, std::min<long>(1024, (rows + 63) / 64)); }

hipError_t launch_colsum(const float* x, const float* y, const float* mean, const float* rstd, int shift, int L, long rows, int C,
                         float* part, float* out, hipStream_t s) {
  const int chunks = colsum_chunks(rows);
  const int rpc = (int)((rows + chunks - 1) / chunks);
  const int used = (int)((rows + rpc - 1) / rpc);
  hipLaunchKernelGGL(k_colsum_partial, dim3(used), dim3(256), 0, s, x, y, mean, rstd, shift, L, rows, C, rpc, part);
  hipLaunchKernelGGL(k_colsum_final, dim3((C + 255) / 256), dim3(256), 0, s, part, used, C, out);
  return hipGetLastError();
}

}  // namespace gmf
