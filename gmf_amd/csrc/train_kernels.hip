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
#include <type_traits>

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
template <bool TA, bool TB, int RB>
__global__ void __launch_bounds__(256, 2)
k_gemm_f32(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, const float* __restrict__ bias,
           const float* __restrict__ R, int M, int N, int K, long lda, long ldb, long ldc, long sA, long sB, long sC,
           int ksplits, int kchunk, float alpha, float* __restrict__ part, int relu) {
  // RB = 2: 128 x 128 workgroup tile (a wave: 64 x 64); RB = 1: 64 x 128 (a wave: 32 x 64) - for tall-skinny products whose
  // 128-row tiling leaves half the chip without a workgroup
  constexpr int BM = 64 * RB;
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int b = blockIdx.z / ksplits, ks = blockIdx.z - b * ksplits;
  const int kbeg = ks * kchunk, kend = min(K, kbeg + kchunk);
  const int m0 = blockIdx.y * BM + 32 * RB * wr, n0 = blockIdx.x * 128 + 64 * wc;
  A += (size_t)b * sA;
  Bm += (size_t)b * sB;
  f32x16 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = zero16();
  // 16-byte loads for the operand whose contraction index is contiguous in memory (A when !TA, B when TB), when every
  // address is aligned; full k-steps take them, the ragged last one falls back to the masked scalar form
  const bool vecA = !TA && (lda % 4 == 0) && (sA % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool vecB = TB && (ldb % 4 == 0) && (sB % 4 == 0) && ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);
  auto load_step = [&](const int k0, float (&a)[RB][8], float (&bf)[2][8]) {
    const int kk = k0 + 8 * h;
    const bool fullk = (k0 + 16 <= kend);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int r = m0 + 32 * rb + i;
      if (vecA && fullk) {
        const float4* p = reinterpret_cast<const float4*>(A + (size_t)min(r, M - 1) * lda + kk);
        const float4 u = p[0], w = p[1];
        const float z = (r < M) ? 1.f : 0.f;
        a[rb][0] = u.x * z; a[rb][1] = u.y * z; a[rb][2] = u.z * z; a[rb][3] = u.w * z;
        a[rb][4] = w.x * z; a[rb][5] = w.y * z; a[rb][6] = w.z * z; a[rb][7] = w.w * z;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kk + e;
          const bool ok = (r < M) && (k < kend);
          a[rb][e] = ok ? (TA ? A[(size_t)k * lda + r] : A[(size_t)r * lda + k]) : 0.f;
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int c = n0 + 32 * cb + i;
      if (vecB && fullk) {
        const float4* p = reinterpret_cast<const float4*>(Bm + (size_t)min(c, N - 1) * ldb + kk);
        const float4 u = p[0], w = p[1];
        const float z = (c < N) ? 1.f : 0.f;
        bf[cb][0] = u.x * z; bf[cb][1] = u.y * z; bf[cb][2] = u.z * z; bf[cb][3] = u.w * z;
        bf[cb][4] = w.x * z; bf[cb][5] = w.y * z; bf[cb][6] = w.z * z; bf[cb][7] = w.w * z;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kk + e;
          const bool ok = (c < N) && (k < kend);
          bf[cb][e] = ok ? (TB ? Bm[(size_t)c * ldb + k] : Bm[(size_t)k * ldb + c]) : 0.f;
        }
      }
    }
  };
  auto mma_step = [&](const float (&a)[RB][8], const float (&bf)[2][8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mfma32(a[rb][e], bf[cb][e], acc[rb][cb]);
  };
  // Full k-steps take the fast loader: per-lane base pointers computed once (rows / columns past the edge are CLAMPED, not
  // masked - what they contribute lands in output rows / columns that are never stored), the k-dependent part of every
  // address uniform.  Only a ragged last k-step (K % 16 != 0) goes through the masked loader above.
  const float* pa[RB];
  const float* pb[2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int r = min(m0 + 32 * rb + i, M - 1);
    pa[rb] = TA ? A + r + (size_t)(8 * h) * lda : A + (size_t)r * lda + 8 * h;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int c = min(n0 + 32 * cb + i, N - 1);
    pb[cb] = TB ? Bm + (size_t)c * ldb + 8 * h : Bm + c + (size_t)(8 * h) * ldb;
  }
  auto load_fast = [&](const int k0, float (&a)[RB][8], float (&bf)[2][8]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      if (TA) {
        const float* q = pa[rb] + (size_t)k0 * lda;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[rb][e] = q[(size_t)e * lda];
      } else if (vecA) {
        const float4* q = reinterpret_cast<const float4*>(pa[rb] + k0);
        const float4 u = q[0], w = q[1];
        a[rb][0] = u.x; a[rb][1] = u.y; a[rb][2] = u.z; a[rb][3] = u.w;
        a[rb][4] = w.x; a[rb][5] = w.y; a[rb][6] = w.z; a[rb][7] = w.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[rb][e] = pa[rb][k0 + e];
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      if (!TB) {
        const float* q = pb[cb] + (size_t)k0 * ldb;
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[cb][e] = q[(size_t)e * ldb];
      } else if (vecB) {
        const float4* q = reinterpret_cast<const float4*>(pb[cb] + k0);
        const float4 u = q[0], w = q[1];
        bf[cb][0] = u.x; bf[cb][1] = u.y; bf[cb][2] = u.z; bf[cb][3] = u.w;
        bf[cb][4] = w.x; bf[cb][5] = w.y; bf[cb][6] = w.z; bf[cb][7] = w.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[cb][e] = pb[cb][k0 + e];
      }
    }
  };
  // register double buffering: the loads of k-step t + 1 are in flight while the MFMAs of k-step t run (a wave that first
  // waits for its operands and then multiplies leaves the matrix pipe idle for a memory round trip per step)
  const int kfull = kbeg + ((kend - kbeg) / 16) * 16;      // end of the full k-steps
  float a0[RB][8], b0[2][8], a1[RB][8], b1[2][8];
  if (kbeg < kfull) load_fast(kbeg, a0, b0);
  for (int k0 = kbeg; k0 < kfull; k0 += 32) {
    if (k0 + 16 < kfull) load_fast(k0 + 16, a1, b1);
    mma_step(a0, b0);
    if (k0 + 16 < kfull) {
      if (k0 + 32 < kfull) load_fast(k0 + 32, a0, b0);
      mma_step(a1, b1);
    }
  }
  if (kfull < kend) {
    load_step(kfull, a0, b0);
    mma_step(a0, b0);
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
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
          C[(size_t)b * sC + (size_t)row * ldc + col] = relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
}

// =========================================================================================
// k_gemm_lds: the same product as k_gemm_f32 for operands whose contiguous dimension is 16-byte friendly (see
// gemm_lds_ok): both 128 x 16 operand tiles of a k-step go global -> registers -> LDS with coalesced 16-byte loads
// (k_gemm_f32's lanes each read their own row: 64 cache lines per load instruction), double buffered, one barrier per
// k-step; the four waves read their MFMA fragments from the LDS (rows padded to 20 floats: conflict-free b128 reads).
//   LDS per stage: A_s [128][20] | B_s [128][20] floats (k contiguous per row / column), two stages = 40 KiB.
//   grid (ceil(N / 128), ceil(M / 128), batch * ksplits), block 256.
// =========================================================================================
constexpr int kGemmLd = 20;                          // floats per LDS row (16 + 4 of padding)

// Workgroup tile (64 RB) x (64 CB), a wave (32 RB) x (32 CB): 128 x 128 for large products, 64 x 128 and 64 x 64 for the
// tall-skinny ones (activations x a 128-column weight, the P V products): enough workgroups WITHOUT splitting K, i.e. without
// the partial tiles and their reduction.
template <bool TA, bool TB, int RB, int CB>
__global__ void __launch_bounds__(256, 2)
k_gemm_lds(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, const float* __restrict__ bias,
           const float* __restrict__ R, int M, int N, int K, long lda, long ldb, long ldc, long sA, long sB, long sC,
           int ksplits, int kchunk, float alpha, float* __restrict__ part, int relu) {
  constexpr int BM = 64 * RB, BN = 64 * CB;
  constexpr int kStage = (BM + BN) * kGemmLd;        // floats per stage: A_s [BM][20] | B_s [BN][20]
  __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
  const int wave = t >> 6, wr = wave >> 1, wc = wave & 1;
  const int b = blockIdx.z / ksplits, ks = blockIdx.z - b * ksplits;
  const int kbeg = ks * kchunk, kend = min(K, kbeg + kchunk);
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  A += (size_t)b * sA;
  Bm += (size_t)b * sB;
  f32x16 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = zero16();

  // global -> registers in 16-byte pieces, ROWS = rows (columns) of the operand tile, ROWS / 64 pieces per thread:
  //   k-contiguous operand (A when !TA, B when TB): piece p covers row (t >> 2) + 64 p, k = k0 + 4 (t & 3) .. + 3
  //   row-contiguous operand (A when TA, B when !TB), 128 rows: piece p covers k = k0 + (t >> 5) + 8 p, rows 4 (t & 31) .. + 3;
  //                                                    64 rows: the one piece covers k = k0 + (t >> 4), rows 4 (t & 15) .. + 3
  // Every fetch ISSUES its loads, whatever k0 (addresses past the end are clamped into the matrix and the values replaced by
  // zeros): a load under a branch would make the compiler's s_waitcnt vmcnt conservative at the join - it would wait for
  // the loads just issued instead of only for the ones of the previous step.
  auto fetch = [&](auto rows_tag, const float* base, long ld, bool kcontig, int r0, int rmax, int k0, float4 (&v)[2]) {
    constexpr int ROWS = decltype(rows_tag)::value, NP = ROWS / 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (kcontig) {
        const int row = min(r0 + (t >> 2) + 64 * p, rmax - 1), k = k0 + 4 * (t & 3);
        const float4 x = *reinterpret_cast<const float4*>(base + (size_t)row * ld + min(k, K - 4));
        const bool ok = k < kend;       // (component-wise selects: a select between two float4 OBJECTS goes through scratch)
        v[p].x = ok ? x.x : 0.f; v[p].y = ok ? x.y : 0.f; v[p].z = ok ? x.z : 0.f; v[p].w = ok ? x.w : 0.f;
      } else {
        const int k = (ROWS == 128) ? k0 + (t >> 5) + 8 * p : k0 + (t >> 4);
        const int row = r0 + 4 * ((ROWS == 128) ? (t & 31) : (t & 15));
        const float4 x = *reinterpret_cast<const float4*>(base + (size_t)min(k, K - 1) * ld + min(row, rmax - 4));
        const bool ok = k < kend && row < rmax;
        v[p].x = ok ? x.x : 0.f; v[p].y = ok ? x.y : 0.f; v[p].z = ok ? x.z : 0.f; v[p].w = ok ? x.w : 0.f;
      }
    }
  };
  auto stash = [&](auto rows_tag, float* dst, bool kcontig, const float4 (&v)[2]) {
    constexpr int ROWS = decltype(rows_tag)::value, NP = ROWS / 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (kcontig) {
        *reinterpret_cast<float4*>(dst + ((t >> 2) + 64 * p) * kGemmLd + 4 * (t & 3)) = v[p];
      } else {
        float* q = (ROWS == 128) ? dst + (4 * (t & 31)) * kGemmLd + (t >> 5) + 8 * p : dst + (4 * (t & 15)) * kGemmLd + (t >> 4);
        q[0] = v[p].x; q[kGemmLd] = v[p].y; q[2 * kGemmLd] = v[p].z; q[3 * kGemmLd] = v[p].w;
      }
    }
  };
  const std::integral_constant<int, BM> rowsA;
  const std::integral_constant<int, BN> rowsB;
  auto mma_stage = [&](const int stage) {
    const float* As = lds + stage * kStage + (32 * RB * wr + i) * kGemmLd + 8 * h;
    const float* Bs = lds + stage * kStage + BM * kGemmLd + (32 * CB * wc + i) * kGemmLd + 8 * h;
    float a[RB][8], bf[CB][8];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float4 u = *reinterpret_cast<const float4*>(As + 32 * rb * kGemmLd);
      const float4 w = *reinterpret_cast<const float4*>(As + 32 * rb * kGemmLd + 4);
      a[rb][0] = u.x; a[rb][1] = u.y; a[rb][2] = u.z; a[rb][3] = u.w; a[rb][4] = w.x; a[rb][5] = w.y; a[rb][6] = w.z; a[rb][7] = w.w;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const float4 u = *reinterpret_cast<const float4*>(Bs + 32 * cb * kGemmLd);
      const float4 w = *reinterpret_cast<const float4*>(Bs + 32 * cb * kGemmLd + 4);
      bf[cb][0] = u.x; bf[cb][1] = u.y; bf[cb][2] = u.z; bf[cb][3] = u.w; bf[cb][4] = w.x; bf[cb][5] = w.y; bf[cb][6] = w.z; bf[cb][7] = w.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = mfma32(a[rb][e], bf[cb][e], acc[rb][cb]);
  };
  // Prefetch distance TWO k-steps through two register sets: the operands of k-step t + 2 are requested at the top of
  // step t, held in registers across step t + 1's MFMAs and written to the LDS at its end - a full k-step (2048 matrix-pipe
  // cycles) plus a barrier more than one memory round trip.  One barrier per k-step, two LDS stages.
  float4 va0[2], vb0[2], va1[2], vb1[2];
  float* const A0 = lds;
  float* const B0 = lds + BM * kGemmLd;
  float* const A1 = lds + kStage;
  float* const B1 = lds + kStage + BM * kGemmLd;
  fetch(rowsA, A, lda, !TA, m0, M, kbeg, va0);
  fetch(rowsB, Bm, ldb, TB, n0, N, kbeg, vb0);
  fetch(rowsA, A, lda, !TA, m0, M, kbeg + 16, va1);
  fetch(rowsB, Bm, ldb, TB, n0, N, kbeg + 16, vb1);
  stash(rowsA, A0, !TA, va0);
  stash(rowsB, B0, TB, vb0);
  __syncthreads();
  // invariant at the top of a pair of steps: LDS stage 0 holds k0, register set 1 holds k0 + 16
  int k0 = kbeg;
  for (; k0 + 48 < kend; k0 += 32) {           // steady state: both fetches of the pair are needed - no branch around a load
    fetch(rowsA, A, lda, !TA, m0, M, k0 + 32, va0);
    fetch(rowsB, Bm, ldb, TB, n0, N, k0 + 32, vb0);
    mma_stage(0);
    stash(rowsA, A1, !TA, va1);
    stash(rowsB, B1, TB, vb1);
    __syncthreads();
    fetch(rowsA, A, lda, !TA, m0, M, k0 + 48, va1);
    fetch(rowsB, Bm, ldb, TB, n0, N, k0 + 48, vb1);
    mma_stage(1);
    stash(rowsA, A0, !TA, va0);
    stash(rowsB, B0, TB, vb0);
    __syncthreads();
  }
  for (; k0 < kend; k0 += 32) {                // the last (at most three) steps: nothing fetched that is not used
    if (k0 + 32 < kend) {
      fetch(rowsA, A, lda, !TA, m0, M, k0 + 32, va0);
      fetch(rowsB, Bm, ldb, TB, n0, N, k0 + 32, vb0);
    }
    mma_stage(0);
    if (k0 + 16 >= kend) break;
    stash(rowsA, A1, !TA, va1);
    stash(rowsB, B1, TB, vb1);
    __syncthreads();
    mma_stage(1);
    if (k0 + 32 < kend) {
      stash(rowsA, A0, !TA, va0);
      stash(rowsB, B0, TB, vb0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int col = n0 + 32 * CB * wc + 32 * cb + i;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * RB * wr + 32 * rb + 8 * (r >> 2) + 4 * h + (r & 3);
        if (row >= M) continue;
        if (ksplits > 1) {
          part[((size_t)blockIdx.z * M + row) * N + col] = acc[rb][cb][r];
        } else {
          float v = alpha * acc[rb][cb][r];
          if (bias) v += bias[col];
          if (R) v += R[(size_t)b * sC + (size_t)row * ldc + col];
          C[(size_t)b * sC + (size_t)row * ldc + col] = relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
}

// Sum of the split-K partial tiles in a FIXED order (the result does not depend on scheduling): 64 outputs per workgroup,
// G = blockDim.x / 64 wave-groups (1 for a few splits of a large output, up to 16 for the ~100 splits of a weight gradient);
// group g adds the splits z = g, g + G, ... in increasing z, the G group sums are added in index order.
__global__ void __launch_bounds__(1024)
k_gemm_reduce(const float* __restrict__ part, float* __restrict__ C, const float* __restrict__ bias, const float* __restrict__ R,
              int M, int N, long ldc, long sC, int ksplits, float alpha, long total, int relu) {
  __shared__ float red[16][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6, G = blockDim.x >> 6;
  const long idx = (long)blockIdx.x * 64 + o;
  const long mn = (long)M * N;
  const bool ok = idx < total;
  const int b = ok ? (int)(idx / mn) : 0;
  const long rem = ok ? idx - (long)b * mn : 0;
  float s = 0.f;
  if (ok)
    for (int z = g; z < ksplits; z += G) s += part[((size_t)b * ksplits + z) * mn + rem];
  red[g][o] = s;
  __syncthreads();
  if (g != 0 || !ok) return;
  float t = red[0][o];
  for (int q = 1; q < G; ++q) t += red[q][o];
  const int row = (int)(rem / N), col = (int)(rem - (long)row * N);
  float v = alpha * t;
  if (bias) v += bias[col];
  if (R) v += R[(size_t)b * sC + (size_t)row * ldc + col];
  C[(size_t)b * sC + (size_t)row * ldc + col] = relu ? fmaxf(v, 0.f) : v;
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
k_softmax_fwd(const float* __restrict__ S, const float* __restrict__ mul, float* __restrict__ P, long rows, int T, float scale) {
  // mul (optional, same shape as S): the logits are mul * (S * scale) - the spatial-consistency attention's
  // softmax(compat * QK^T / sqrt(C)) (PointDSC.py:60-62)
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* s = S + row * T;
  const float* m = mul ? mul + row * T : nullptr;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) mx = fmaxf(mx, m ? m[j] * (s[j] * scale) : s[j] * scale);
  mx = wave_max_f(mx);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += expf((m ? m[j] * (s[j] * scale) : s[j] * scale) - mx);
  const float inv = 1.0f / wave_sum_f(sum);
  for (int j = lane; j < T; j += 64) P[row * T + j] = expf((m ? m[j] * (s[j] * scale) : s[j] * scale) - mx) * inv;
}

__global__ void __launch_bounds__(256)
k_softmax_bwd(const float* __restrict__ P, const float* __restrict__ dP, const float* __restrict__ mul, float* __restrict__ dS,
              long rows, int T, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float dot = 0.f;
  for (int j = lane; j < T; j += 64) dot = fmaf(P[row * T + j], dP[row * T + j], dot);
  dot = wave_sum_f(dot);
  for (int j = lane; j < T; j += 64)
    dS[row * T + j] = scale * (mul ? mul[row * T + j] : 1.0f) * P[row * T + j] * (dP[row * T + j] - dot);
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
                 const float* __restrict__ rstd, const float* __restrict__ cmean, const float* __restrict__ crstd, int center_x,
                 int shift, int L, long rows, int C, int rows_per_chunk, float* __restrict__ part,
                 const float* __restrict__ relu_y, int dual) {
  // relu_y: x is masked by the saved output of a ReLU (x' = x where relu_y > 0, else 0: the gradient entering a fused
  // Linear / BatchNorm + ReLU).  dual: the plain column sums of x' are produced too, as columns C .. 2C - 1 of the
  // partial rows (which are then 2C wide): sum x' y' and sum x' in ONE pass over x (a LayerNorm's dgamma / dbeta, a
  // BatchNorm's, an LCPE tap with its bias).
  // cmean / crstd (per COLUMN: BatchNorm's statistics): y' = (y - cmean[c]) * crstd[c]; center_x: x' = x - cmean[c]
  // block: 4 waves; lane = column within the 64-column block blockIdx.y, wave w takes rows r0 + w, r0 + w + 4, ...;
  // the four wave sums are added in wave order through LDS
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + (long)rows_per_chunk);
  float s = 0.f, s1 = 0.f;
  if (c < C) {
    const float cm = cmean ? cmean[c] : 0.f, cr = crstd ? crstd[c] : 1.f;
    // four rows of this wave at a time: their loads are all requested before the first is used (a chunk is 16 rows per wave -
    // one load per round trip made the kernel 16 dependent round trips long); the additions keep the row order
    for (long rb = r0 + wave; rb < r1; rb += 16) {
      float v[4], yv[4], mu[4], rs[4], ry_mask[4];
      bool ok[4], yok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long rr = rb + 4 * u;
        ok[u] = rr < r1;
        const long r = ok[u] ? rr : rb;
        v[u] = x[r * C + c];
        ry_mask[u] = relu_y ? relu_y[r * C + c] : 1.f;
        yok[u] = true; yv[u] = 1.f; mu[u] = 0.f; rs[u] = 1.f;
        if (y) {
          const int l = (int)(r % L) + shift;
          yok[u] = l >= 0 && l < L;
          const long ry = yok[u] ? r + shift : r;
          yv[u] = y[ry * C + c];
          if (mean) { mu[u] = mean[ry]; rs[u] = rstd[ry]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) break;
        float vv = v[u];
        if (relu_y && !(ry_mask[u] > 0.f)) vv = 0.f;
        if (center_x) vv -= cm;
        s1 += vv;
        if (y) {
          if (!yok[u]) continue;
          float t = yv[u];
          if (mean) t = (t - mu[u]) * rs[u];
          if (cmean) t = (t - cm) * cr;
          vv *= t;
        }
        s += vv;
      }
    }
  }
  red[0][wave][lane] = s;
  red[1][wave][lane] = s1;
  __syncthreads();
  const int Cw = dual ? 2 * C : C;
  if (wave == 0 && c < C) part[(size_t)blockIdx.x * Cw + c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
  if (dual && wave == 1 && c < C)
    part[(size_t)blockIdx.x * Cw + C + c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
}

// one wave per column: lane l adds chunks l, l + 64, ... in order, then a fixed butterfly over the lanes (deterministic)
__global__ void __launch_bounds__(256)
k_colsum_final(const float* __restrict__ part, int chunks, int C, float* __restrict__ out, float* __restrict__ out2, int c_split) {
  // columns c_split .. C - 1 go to out2 (the plain sums of a dual pass) when it is given
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float s = 0.f;
  for (int k = lane; k < chunks; k += 64) s += part[(size_t)k * C + c];
  s = wave_sum_f(s);
  if (lane == 0) {
    if (out2 && c >= c_split) out2[c - c_split] = s;
    else out[c] = s;
  }
}

// =========================================================================================
// F.normalize(x, p = 2, dim = -1) (PointDSC.py:229): y = x / max(||x||, 1e-12), one wave per row; the norm is saved.
// Backward from the saved OUTPUT: dx = (dy - y <dy, y>) / max(||x||, 1e-12).
// =========================================================================================
__global__ void __launch_bounds__(256)
k_normalize_fwd(const float* __restrict__ x, float* __restrict__ nrm, float* __restrict__ y, long rows, int C) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[row * C + c]; s = fmaf(v, v, s); }
  const float n = fmaxf(sqrtf(wave_sum_f(s)), 1e-12f);
  for (int c = lane; c < C; c += 64) y[row * C + c] = x[row * C + c] / n;
  if (lane == 0) nrm[row] = n;
}

__global__ void __launch_bounds__(256)
k_normalize_bwd(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ nrm, float* __restrict__ dx,
                long rows, int C) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float d = 0.f;
  for (int c = lane; c < C; c += 64) d = fmaf(dy[row * C + c], y[row * C + c], d);
  d = wave_sum_f(d);
  const float inv = 1.0f / nrm[row];
  for (int c = lane; c < C; c += 64) dx[row * C + c] = (dy[row * C + c] - y[row * C + c] * d) * inv;
}

// =========================================================================================
// BatchNorm1d in TRAINING mode over the rows of x [rows, C] (the reference's BatchNorm1d on [B, C, N], PointDSC.py:13-21,
// 104-109: statistics per channel over batch and positions): the column statistics come from k_colsum_*; here the
// element-wise halves.  forward: y = (x - mean[c]) * rstd[c] * gamma[c] + beta[c] (-> ReLU).
// backward (dy already masked by the ReLU): dx = gamma rstd (dy - sdy / R - xhat sdyx / R), sdy = sum_r dy, sdyx = sum_r dy xhat.
// =========================================================================================
__global__ void __launch_bounds__(256)
k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
           const float* __restrict__ beta, float* __restrict__ y, int C, int relu, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const float v = fmaf((x[idx] - mean[c]) * rstd[c], gamma[c], beta[c]);
  y[idx] = relu ? fmaxf(v, 0.f) : v;
}

__global__ void __launch_bounds__(256)
k_bn_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
         const float* __restrict__ gamma, const float* __restrict__ sdy, const float* __restrict__ sdyx, float* __restrict__ dx, int C,
         float inv_rows, long total, const float* __restrict__ relu_y) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const float xh = (x[idx] - mean[c]) * rstd[c];
  const float g = (relu_y && !(relu_y[idx] > 0.f)) ? 0.f : dy[idx];      // the ReLU that followed the BatchNorm
  dx[idx] = gamma[c] * rstd[c] * (g - sdy[c] * inv_rows - xh * sdyx[c] * inv_rows);
}

// mean[c] = s[c] / rows ; rstd[c] = rsqrt(ss[c] / rows + eps) ; running statistics as torch updates them (momentum, unbiased var)
__global__ void k_bn_finish(const float* __restrict__ s, const float* __restrict__ ss, float* __restrict__ mean, float* __restrict__ rstd,
                            float* __restrict__ run_mean, float* __restrict__ run_var, int C, float rows, float eps, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (ss == nullptr) { mean[c] = s[c] / rows; return; }
  const float var = ss[c] / rows;
  rstd[c] = rsqrtf(var + eps);
  if (run_mean) {
    run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean[c];
    run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * (rows / fmaxf(rows - 1.0f, 1.0f));
  }
}

// out = y > 0 ? dy : 0   (ReLU backward from the saved OUTPUT)
__global__ void __launch_bounds__(256)
k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  out[idx] = y[idx] > 0.f ? dy[idx] : 0.f;
}

// =========================================================================================
// Loss backward halves (libs/loss.py) for an upstream gradient of 1.
//   k_bce_bwd: d/dpred of mean BCE-with-logits with pos_weight pw (pw = 1: the unbalanced form; weight: optional per-element):
//              (sigmoid(x) (1 - gt + pw gt) - pw gt) * w / count                                        loss.py:85-93
//   k_sm_dense_bwd: dL/dM of SpectralMatchingLoss(M, gt) with the per-pair constants cP, cN of k_sm_bwd_prep (consts[b] = {cP, cN, ..}):
//              gtM (M - 1) cP + (1 - gtM) M cN, zero on the diagonal                                     loss.py:116-140
//   k_sim_bwd_G: from S = Fn Fn^T and an upstream dM: G = dM [0 <= u <= 1] / sigma^2 off the diagonal (clamp's gradient mask,
//              u = 1 - (1 - S) / sigma^2), and rowdsig[r] = sum_j dM [..] 2 (1 - S) / sigma^3            PointDSC.py:231-234
// =========================================================================================
// pw[0] = (relu(sum(1 - gt) - 1) + 1) / (relu(sum(gt) - 1) + 1) over the whole batch (loss.py:85-86,93), one workgroup
__global__ void __launch_bounds__(1024)
k_bce_posweight(const float* __restrict__ gt, long total, float* __restrict__ pw) {
  __shared__ float red[16];
  float n1 = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) n1 += gt[i];
  n1 = wave_sum_f(n1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n1;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red[w];
    const double pos = fmax(s - 1.0, 0.0) + 1.0, neg = fmax((double)total - s - 1.0, 0.0) + 1.0;
    pw[0] = (float)(neg / pos);
  }
}

__global__ void __launch_bounds__(256)
k_bce_bwd(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ weight, float* __restrict__ dpred,
          const float* __restrict__ pw_dev, float inv_count, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float pos_weight = pw_dev ? pw_dev[0] : 1.0f;
  const float x = pred[idx], g = gt[idx];
  const float sg = 1.0f / (1.0f + expf(-x));
  float d = sg * (1.0f - g + pos_weight * g) - pos_weight * g;
  if (weight) d *= weight[idx];
  dpred[idx] = d * inv_count;
}

__global__ void __launch_bounds__(256)
k_sm_dense_bwd(const float* __restrict__ M, long ldm, const float* __restrict__ gt, const float* __restrict__ consts,
               float* __restrict__ dM, int N, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long nn = (long)N * N;
  const int b = (int)(idx / nn);
  const long rem = idx - (long)b * nn;
  const int i = (int)(rem / N), j = (int)(rem - (long)i * N);
  const float m = M[((size_t)b * N + i) * ldm + j];
  const bool both = (i != j) && gt[(size_t)b * N + i] == 1.0f && gt[(size_t)b * N + j] == 1.0f;
  dM[idx] = (i == j) ? 0.f : (both ? (m - 1.0f) * consts[4 * b] : m * consts[4 * b + 1]);
}

__global__ void __launch_bounds__(256)
k_sim_bwd_G(const float* __restrict__ S, const float* __restrict__ dM, float* __restrict__ G, float* __restrict__ rowdsig, int N,
            float inv_sig2, float two_inv_sig3, long rows, const float* __restrict__ sigma_dev) {
  inv_sig2 = sigma_inv2(inv_sig2, sigma_dev);
  two_inv_sig3 = sigma_two_inv3(two_inv_sig3, sigma_dev);
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int i = (int)(row % N);
  float ds = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float s = S[row * N + j], u = 1.0f - (1.0f - s) * inv_sig2;
    const bool inside = (u >= 0.f) && (u <= 1.f) && (j != i);
    const float d = inside ? dM[row * N + j] : 0.f;
    G[row * N + j] = d * inv_sig2;
    ds = fmaf(d, (1.0f - s) * two_inv_sig3, ds);
  }
  ds = wave_sum_f(ds);
  if (lane == 0) rowdsig[row] = ds;
}

// per pair consts {cP, cN, 0, 0} of the spectral-matching loss (the balanced form or the MSE form), as k_sm_bwd_prep
__global__ void __launch_bounds__(256)
k_sm_consts(const float* __restrict__ gt, float* __restrict__ consts, int B, int N, int balanced) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float n1 = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) n1 += (gt[(size_t)b * N + i] == 1.0f) ? 1.f : 0.f;
  n1 = wave_sum_f(n1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n1;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = (double)red[0] + red[1] + red[2] + red[3];
    const double np = n * n - n, nn = (double)N * N - np;
    double cP, cN;
    if (balanced) { cP = 1.0 / ((double)B * (fmax(np - 1.0, 0.0) + 1.0)); cN = 1.0 / ((double)B * (fmax(nn - 1.0, 0.0) + 1.0)); }
    else cP = cN = 2.0 / ((double)B * (double)N * (double)N);
    consts[4 * b] = (float)cP; consts[4 * b + 1] = (float)cN; consts[4 * b + 2] = 0.f; consts[4 * b + 3] = 0.f;
  }
}

// -----------------------------------------------------------------------------------------
static inline unsigned blocks_of(long total) { return (unsigned)((total + 255) / 256); }

// Workgroup tile of a product: 128 x 128 when that alone gives the chip two workgroups per CU; else 64-row tiles, and 64-column
// tiles too (LDS-staged kernel only) when the 64 x 128 tiling would still have to split a contraction it can do whole
static inline void gemm_tile(int M, int N, int K, int batch, bool lds_ok, int* bm, int* bn) {
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  *bm = (t128 < 512 && M > 64) ? 64 : 128;
  *bn = 128;
  const long t64 = (long)((M + *bm - 1) / *bm) * ((N + 127) / 128) * batch;
  if (lds_ok && *bm == 64 && t64 < 384 && K >= 256 && N > 64) *bn = 64;
}

static inline bool gemm_lds_ok(bool ta, bool tb, const float* A, const float* B, int M, int N, int K, long lda, long ldb, long sA,
                               long sB) {
  // the LDS-staged kernel needs 16-byte pieces along each operand's contiguous dimension
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool a_ok = al16(A) && lda % 4 == 0 && sA % 4 == 0 && (ta ? M % 4 == 0 : K % 4 == 0);
  const bool b_ok = al16(B) && ldb % 4 == 0 && sB % 4 == 0 && (tb ? K % 4 == 0 : N % 4 == 0);
  return a_ok && b_ok && K >= 4 && M >= 4 && N >= 4;
}

int gemm_ksplits(bool ta, bool tb, const float* A, const float* B, int M, int N, int K, long lda, long ldb, long sA, long sB,
                 int batch) {
  // few output tiles and a long contraction (weight gradients: K = every row of the batch): split K so that the launch has
  // ~768 workgroups, each split at least 64 deep, the partial tiles at most ~48 MB (written and read once by k_gemm_reduce)
  int bm, bn;
  gemm_tile(M, N, K, batch, gemm_lds_ok(ta, tb, A, B, M, N, K, lda, ldb, sA, sB), &bm, &bn);
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
  if (tiles >= 384 || K < 256) return 1;
  long s = std::min<long>((768 + tiles - 1) / tiles, K / 64);
  const long out_bytes = (long)M * N * batch * 4;
  s = std::min<long>(s, std::max<long>(1, (48L << 20) / out_bytes));
  return (int)std::max<long>(1, std::min<long>(s, 128));
}

hipError_t launch_gemm_f32(bool ta, bool tb, const float* A, const float* B, float* C, const float* bias, const float* R, int M, int N,
                           int K, long lda, long ldb, long ldc, long sA, long sB, long sC, int batch, float alpha, float* part,
                           int ksplits, int relu, hipStream_t s) {
  if (!part) ksplits = 1;
  int kchunk = (K + ksplits - 1) / ksplits;
  kchunk = (kchunk + 15) / 16 * 16;
  ksplits = (K + kchunk - 1) / kchunk;
  const bool lds_ok = gemm_lds_ok(ta, tb, A, B, M, N, K, lda, ldb, sA, sB);    // anything else takes the register-direct kernel
  int bm, bn;
  gemm_tile(M, N, K, batch, lds_ok, &bm, &bn);
  const dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm, batch * ksplits);
#define GMF_GEMM_ARGS grid, dim3(256), 0, s, A, B, C, bias, R, M, N, K, lda, ldb, ldc, sA, sB, sC, ksplits, kchunk, alpha, part, relu
#define GMF_GEMM_LDS(TA, TB)                                                                         \
  do {                                                                                               \
    if (bm == 64 && bn == 64) hipLaunchKernelGGL((k_gemm_lds<TA, TB, 1, 1>), GMF_GEMM_ARGS);         \
    else if (bm == 64) hipLaunchKernelGGL((k_gemm_lds<TA, TB, 1, 2>), GMF_GEMM_ARGS);                \
    else hipLaunchKernelGGL((k_gemm_lds<TA, TB, 2, 2>), GMF_GEMM_ARGS);                              \
  } while (0)
#define GMF_GEMM_DIRECT(TA, TB)                                                                      \
  do {                                                                                               \
    if (bm == 64) hipLaunchKernelGGL((k_gemm_f32<TA, TB, 1>), GMF_GEMM_ARGS);                        \
    else hipLaunchKernelGGL((k_gemm_f32<TA, TB, 2>), GMF_GEMM_ARGS);                                 \
  } while (0)
  if (lds_ok) {
    if (ta && tb) GMF_GEMM_LDS(true, true);
    else if (ta) GMF_GEMM_LDS(true, false);
    else if (tb) GMF_GEMM_LDS(false, true);
    else GMF_GEMM_LDS(false, false);
  } else {
    if (ta && tb) GMF_GEMM_DIRECT(true, true);
    else if (ta) GMF_GEMM_DIRECT(true, false);
    else if (tb) GMF_GEMM_DIRECT(false, true);
    else GMF_GEMM_DIRECT(false, false);
  }
#undef GMF_GEMM_LDS
#undef GMF_GEMM_DIRECT
#undef GMF_GEMM_ARGS
  if (ksplits > 1) {
    const long total = (long)batch * M * N;
    const int groups = ksplits > 64 ? 16 : ksplits > 32 ? 8 : ksplits > 16 ? 4 : ksplits > 8 ? 2 : 1;
    hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((total + 63) / 64)), dim3(64 * groups), 0, s, part, C, bias, R, M, N, ldc, sC, ksplits, alpha, total, relu);
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

hipError_t launch_softmax(bool backward, const float* a, const float* b, const float* mul, float* out, long rows, int T, float scale,
                          hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (backward) hipLaunchKernelGGL(k_softmax_bwd, grid, dim3(256), 0, s, a, b, mul, out, rows, T, scale);
  else hipLaunchKernelGGL(k_softmax_fwd, grid, dim3(256), 0, s, a, mul, out, rows, T, scale);
  return hipGetLastError();
}

hipError_t launch_geglu(bool backward, const float* hdn, const float* dg, float* out, long rows, int H, hipStream_t s) {
  const long total = rows * H;
  if (backward) hipLaunchKernelGGL(k_geglu_bwd, dim3(blocks_of(total)), dim3(256), 0, s, hdn, dg, out, H, total);
  else hipLaunchKernelGGL(k_geglu_fwd, dim3(blocks_of(total)), dim3(256), 0, s, hdn, out, H, total);
  return hipGetLastError();
}

int colsum_chunks(long rows) { return (int)std::max<long>(1, std::min<long>(1024, (rows + 63) / 64)); }

hipError_t launch_colsum(const float* x, const float* y, const float* mean, const float* rstd, const float* cmean, const float* crstd,
                         int center_x, int shift, int L, long rows, int C, float* part, float* out, hipStream_t s,
                         const float* relu_y, bool dual, float* out2) {
  const int chunks = colsum_chunks(rows);
  const int rpc = (int)((rows + chunks - 1) / chunks);
  const int used = (int)((rows + rpc - 1) / rpc);
  hipLaunchKernelGGL(k_colsum_partial, dim3(used, (C + 63) / 64), dim3(256), 0, s, x, y, mean, rstd, cmean, crstd, center_x, shift, L,
                     rows, C, rpc, part, relu_y, dual ? 1 : 0);
  const int Cw = dual ? 2 * C : C;                  // out: [sum x' y' (C) | sum x' (C)] when dual
  hipLaunchKernelGGL(k_colsum_final, dim3((Cw + 3) / 4), dim3(256), 0, s, part, used, Cw, out, dual ? out2 : (float*)nullptr, C);
  return hipGetLastError();
}

hipError_t launch_bn_finish(const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean, float* run_var, int C,
                            long rows, float eps, float momentum, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_finish, dim3((C + 255) / 256), dim3(256), 0, s, sum, sumsq, mean, rstd, run_mean, run_var, C, (float)rows, eps,
                     momentum);
  return hipGetLastError();
}

hipError_t launch_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                           long rows, int C, int relu, hipStream_t s) {
  const long total = rows * C;
  hipLaunchKernelGGL(k_bn_apply, dim3(blocks_of(total)), dim3(256), 0, s, x, mean, rstd, gamma, beta, y, C, relu, total);
  return hipGetLastError();
}

hipError_t launch_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* sdy,
                         const float* sdyx, float* dx, long rows, int C, hipStream_t s, const float* relu_y) {
  const long total = rows * C;
  hipLaunchKernelGGL(k_bn_bwd, dim3(blocks_of(total)), dim3(256), 0, s, dy, x, mean, rstd, gamma, sdy, sdyx, dx, C, 1.0f / (float)rows, total,
                     relu_y);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Dense spatial-consistency matrix for the trainable path (PointDSC.py:216-221, computed under no_grad by the reference):
//   c[b,i,j] = clamp(1 - (|p_i - p_j| - |q_i - q_j|)^2 / sigma_d^2, min = 0),  row-major [B,N,N].
// (The inference path never materialises it row-major: k_compat_build writes it in the attention kernel's fragment order.)
// grid (ceil(N/256), N, B), block 256: one row i per blockIdx.y, coalesced stores.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_compat_dense(const float* __restrict__ src, const float* __restrict__ tgt, float* __restrict__ out, int N, float inv_sigmad2) {
  const int b = blockIdx.z, i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const float* ps = src + (size_t)b * N * 3;
  const float* pt = tgt + (size_t)b * N * 3;
  const float ax = ps[3 * i] - ps[3 * j], ay = ps[3 * i + 1] - ps[3 * j + 1], az = ps[3 * i + 2] - ps[3 * j + 2];
  const float bx = pt[3 * i] - pt[3 * j], by = pt[3 * i + 1] - pt[3 * j + 1], bz = pt[3 * i + 2] - pt[3 * j + 2];
  const float d = sqrtf(ax * ax + ay * ay + az * az) - sqrtf(bx * bx + by * by + bz * bz);
  out[((size_t)b * N + i) * N + j] = fmaxf(1.0f - d * d * inv_sigmad2, 0.f);
}

hipError_t launch_relu_bwd(const float* dy, const float* y, float* out, long total, hipStream_t s) {
  hipLaunchKernelGGL(k_relu_bwd, dim3(blocks_of(total)), dim3(256), 0, s, dy, y, out, total);
  return hipGetLastError();
}

hipError_t launch_bce_bwd(const float* pred, const float* gt, const float* weight, float* dpred, float* pw_scratch, int balanced,
                          long total, hipStream_t s) {
  const bool use_pw = balanced && !weight;            // (a per-element weight replaces the balancing, loss.py:87-89)
  if (use_pw) hipLaunchKernelGGL(k_bce_posweight, dim3(1), dim3(1024), 0, s, gt, total, pw_scratch);
  hipLaunchKernelGGL(k_bce_bwd, dim3(blocks_of(total)), dim3(256), 0, s, pred, gt, weight, dpred, use_pw ? pw_scratch : (float*)nullptr,
                     1.0f / (float)total, total);
  return hipGetLastError();
}

hipError_t launch_sm_dense_bwd(const float* M, long ldm, const float* gt, float* consts, float* dM, int B, int N, int balanced,
                               hipStream_t s) {
  hipLaunchKernelGGL(k_sm_consts, dim3(B), dim3(256), 0, s, gt, consts, B, N, balanced);
  const long total = (long)B * N * N;
  hipLaunchKernelGGL(k_sm_dense_bwd, dim3(blocks_of(total)), dim3(256), 0, s, M, ldm, gt, consts, dM, N, total);
  return hipGetLastError();
}

hipError_t launch_sim_bwd_G(const float* S, const float* dM, float* G, float* rowdsig, int B, int N, float sigma, hipStream_t s,
                            const float* sigma_dev) {
  const long rows = (long)B * N;
  hipLaunchKernelGGL(k_sim_bwd_G, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, dM, G, rowdsig, N, 1.0f / (sigma * sigma),
                     2.0f / (sigma * sigma * sigma), rows, sigma_dev);
  return hipGetLastError();
}

hipError_t launch_compat_dense(const float* src, const float* tgt, float* out, int B, int N, float sigma_d, hipStream_t s) {
  hipLaunchKernelGGL(k_compat_dense, dim3((N + 255) / 256, N, B), dim3(256), 0, s, src, tgt, out, N, 1.0f / (sigma_d * sigma_d));
  return hipGetLastError();
}

hipError_t launch_normalize(bool backward, const float* a, const float* dy, float* nrm, float* out, long rows, int C, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (backward) hipLaunchKernelGGL(k_normalize_bwd, grid, dim3(256), 0, s, a, dy, nrm, out, rows, C);
  else hipLaunchKernelGGL(k_normalize_fwd, grid, dim3(256), 0, s, a, nrm, out, rows, C);
  return hipGetLastError();
}

}  // namespace gmf
