// Validation step (forward half): what libs/trainer.py:194-262 evaluates with the model
// in eval mode and no 'testing' key.  Replaces
//   GMF_PointDSC/models/PointDSC.py:231-234   M = clamp(1 - (1 - Fn Fn^T) / sigma^2, 0, 1), zero diagonal
//   GMF_PointDSC/libs/loss.py:116-140         SpectralMatchingLoss
//   GMF_PointDSC/libs/loss.py:67-113          ClassificationLoss
//   GMF_PointDSC/libs/loss.py:12-64           TransformationLoss
// M is an N x N x 128 product per pair: the normalised features are packed once into split-fp16 MFMA fragments
// (hi + lo planes, 16 KiB per 32 rows), a wave keeps its 32 rows in registers and the column tiles stream through LDS
// by DMA; three f16 MFMAs per multiply-add, fp32 accumulate (the arithmetic of the attention kernel).  Writing M is
// HBM-bound (4 N^2 bytes per pair).  The spectral-matching loss has a second form that never writes M: the same tiles,
// upper triangle only, reduced on the fly (`gmf_spectral_matching_loss_fused`).
// All sums are taken in fp64 in a fixed order (per-workgroup partials, then one ordered pass): results do not depend on
// scheduling.
#include <algorithm>

#include "mfma_core.hpp"
#include "launchers.hpp"

namespace gmf {

namespace {

constexpr int kFeat = 128;            // feature width of the path (PointDSC num_channels)
constexpr int kTileFloats = 4096;     // one 32 x 128 tile as two fp16 planes = 16 KiB

GMF_DEVINL double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the workgroup, valid in thread 0; `red` holds one double per wave
GMF_DEVINL double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

}  // namespace

// [B, N, 128] fp32 rows -> fragment image [B, tiles, plane(2), k-step(8), lane(64)] of 8 halves: lane (h, i) of k-step s
// holds features 16 s + 8 h .. + 7 of row 32 tile + i - the A (and, for F F^T, the B) operand of v_mfma_f32_32x32x16_f16.
__global__ void k_pack_rows_h2(const float* __restrict__ feat, float* __restrict__ img, int N, int tiles, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, s = (idx >> 6) & 7;
  const long bt = idx >> 9;
  const int tile = bt % tiles;
  const long b = bt / tiles;
  const int row = tile * 32 + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < N) {
    const float4* p = reinterpret_cast<const float4*>(feat + ((size_t)b * N + row) * kFeat + k0);
    const float4 x = p[0], y = p[1];
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  }
  f16x8 hi, lo;
  split8h(v, hi, lo);
  f16x8* o = reinterpret_cast<f16x8*>(img + (size_t)bt * kTileFloats);
  o[(0 * 8 + s) * 64 + lane] = hi;
  o[(1 * 8 + s) * 64 + lane] = lo;
}

// LOSS = false: writes M (row stride ldm floats).   LOSS = true: accumulates the two sums of the spectral-matching loss
// over the upper triangle of tiles (M is symmetric: off-diagonal tiles count twice) and writes one (S_pos, S_neg) pair
// per workgroup.
// grid = (row groups of 4 tiles, column chunks, B), 320 threads: four consumer waves that own one row tile each and one
// producer wave that only moves column tiles into a ring of LDS slots.  The split matters because stores count in vmcnt
// on this part: a wave that both waits for its DMA pieces and stores M would wait for its stores to be acknowledged
// every stage.  A store instruction writes two rows x 32 columns: two whole 128-byte lines when ldm is a multiple of 32
// (measured on MI355X at B = 32, N = 5000: 4.7 TB/s of stores with line-aligned rows against 2.7 TB/s with ldm = N,
// whose 128-byte pieces straddle two lines each).
constexpr int kSimRing = 3;    // LDS slots of 16 KiB: three workgroups (15 waves) per CU

template <bool LOSS>
__global__ void __launch_bounds__(320, 4)
k_similarity(const float* __restrict__ img, float* __restrict__ M, const float* __restrict__ gt, double* __restrict__ part,
             int N, int ldm, int tiles, int chunk, float inv_s2, const float* __restrict__ sigma_dev) {
  inv_s2 = sigma_inv2(inv_s2, sigma_dev);
  constexpr int NB = kSimRing;
  __shared__ __attribute__((aligned(16))) float lds[NB * kTileFloats];
  __shared__ unsigned colmask[64];
  __shared__ double red[5];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z;
  const float* pimg = img + (size_t)b * tiles * kTileFloats;
  int c0 = blockIdx.y * chunk;
  const int c1 = min(tiles, c0 + chunk);
  if (LOSS) c0 = max(c0, 4 * (int)blockIdx.x);
  const int n = c1 - c0;                           // stages of this workgroup (uniform)
  const float one_m = 1.0f - inv_s2;               // 1 - (1 - a) / s^2 = a / s^2 + (1 - 1 / s^2)
  double sp = 0.0, sn = 0.0;
  if (n > 0 && wave == 4) {
    // ---- producer: stage k = column tile c0 + k -> slot k % NB; NB - 1 stages in flight
    const float* src = pimg + (size_t)c0 * kTileFloats;
    auto issue = [&](int k) {
      const float* g = src + (size_t)k * kTileFloats;
      float* dst = lds + (k % NB) * kTileFloats;
#pragma unroll
      for (int p = 0; p < 16; ++p) dma_piece_1k(g + p * 256, dst + p * 256, lane);
    };
    if (LOSS) {                                    // inlier bits of the chunk's column tiles
      for (int k = 0; k < n; ++k) {
        const int col = (c0 + k) * 32 + i;
        const bool g = col < N && gt[(size_t)b * N + col] == 1.0f;
        const unsigned long long m = __ballot(g);
        if (lane == 0) colmask[k] = (unsigned)(m & 0xffffffffull);
      }
    }
    for (int k = 0; k < NB - 1 && k < n; ++k) issue(k);
    for (int k = 0; k < n; ++k) {
      const int younger = min(n - k - 1, NB - 2);  // stages issued after stage k
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                             // stage k visible; slot (k - 1) % NB free
      if (k + NB - 1 < n) issue(k + NB - 1);
    }
  } else if (n > 0) {
    // ---- consumers: register r of lane (h, i) = row 32 tile + 8 (r >> 2) + 4 h + (r & 3), column 32 t + i
    const int tile_raw = blockIdx.x * 4 + wave;
    const bool active = tile_raw < tiles;
    const int tile = active ? tile_raw : tiles - 1;
    f16x8 ah[8], al[8];
    {
      const f16x8* a = reinterpret_cast<const f16x8*>(pimg + (size_t)tile * kTileFloats) + lane;
#pragma unroll
      for (int s = 0; s < 8; ++s) { ah[s] = a[s * 64]; al[s] = a[(8 + s) * 64]; }
    }
    unsigned rowmask = 0;                          // bit 8 (r >> 2) + (r & 3) <-> register r
    if (LOSS) {
      const int row = tile * 32 + i;
      const bool g = row < N && gt[(size_t)b * N + row] == 1.0f;
      rowmask = (unsigned)(__ballot(g) & 0xffffffffull) >> (4 * h);
    }
    const bool rows_full = tile * 32 + 32 <= N;
    float* Mt = LOSS ? nullptr : M + ((size_t)b * N + tile * 32) * ldm;            // uniform
    const unsigned lane_off = (unsigned)(4 * h) * (unsigned)ldm + (unsigned)i;      // row 4 h of the tile, column i
    for (int k = 0; k < n; ++k) {
      __syncthreads();
      const int t = c0 + k;
      if (!active || (LOSS && t < tile)) continue; // wave-uniform
      const f16x8* lw = reinterpret_cast<const f16x8*>(lds + (k % NB) * kTileFloats) + lane;
      f32x16 acc = zero16();
#pragma unroll
      for (int s = 0; s < 8; ++s) mma3(acc, ah[s], al[s], lw[s * 64], lw[(8 + s) * 64]);
      float m[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) m[r] = __builtin_amdgcn_fmed3f(fmaf(acc[r], inv_s2, one_m), 0.0f, 1.0f);
      const bool full = rows_full && t * 32 + 32 <= N;
      if (LOSS) {
        // S_neg = sum_all m^2 - sum_both m^2,  S_pos = sum_both (m - 1)^2 = sum_both m^2 - 2 sum_both m + #both
        float all2 = 0.f, b2 = 0.f, b1 = 0.f, bc = 0.f;
        const unsigned g = ((colmask[k] >> i) & 1u) ? rowmask : 0u;
        if (full && t != tile) {                   // no diagonal, no padding
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = (g & (1u << (8 * (r >> 2) + (r & 3)))) ? m[r] : 0.0f;
            all2 = fmaf(m[r], m[r], all2);
            b2 = fmaf(x, x, b2);
            b1 += x;
          }
          bc = (float)__builtin_popcount(g & 0x0f0f0f0fu);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = 8 * (r >> 2) + 4 * h + (r & 3);
            const bool inside = tile * 32 + rr < N && t * 32 + i < N && !(t == tile && rr == i);
            const float v = inside ? m[r] : 0.0f;                  // diagonal and padding: M = 0, gt_M = 0
            const bool both = inside && (g & (1u << (8 * (r >> 2) + (r & 3))));
            const float x = both ? v : 0.0f;
            all2 = fmaf(v, v, all2);
            b2 = fmaf(x, x, b2);
            b1 += x;
            bc += both ? 1.0f : 0.0f;
          }
        }
        const float wgt = t > tile ? 2.0f : 1.0f;
        sp += (double)(wgt * (b2 - 2.0f * b1 + bc));
        sn += (double)(wgt * (all2 - b2));
      } else {
        if (t == tile) {                           // the diagonal is zero (PointDSC.py:234)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (8 * (r >> 2) + 4 * h + (r & 3) == i) m[r] = 0.0f;
        }
        float* p = Mt + t * 32;                    // uniform; lane_off selects row 4 h, column i
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) (p + (size_t)(8 * (r >> 2) + (r & 3)) * ldm)[lane_off] = m[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = 8 * (r >> 2) + 4 * h + (r & 3);
            if (tile * 32 + rr < N && t * 32 + i < N) (p + (size_t)(8 * (r >> 2) + (r & 3)) * ldm)[lane_off] = m[r];
          }
        }
      }
    }
  }
  if (LOSS) {
    const double tp = block_sum(sp, red), tn = block_sum(sn, red);
    if (threadIdx.x == 0) {
      const size_t wg = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      part[2 * wg] = tp;
      part[2 * wg + 1] = tn;
    }
  }
}

// the same two sums from an M the caller holds (libs/loss.py:127-136): grid = (G, B), rows strided over the G workgroups
__global__ void __launch_bounds__(256)
k_sm_partial(const float* __restrict__ M, const float* __restrict__ gt, double* __restrict__ part, int N, int ldm) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const float* g = gt + (size_t)b * N;
  double sp = 0.0, sn = 0.0;
  for (int row = blockIdx.x; row < N; row += gridDim.x) {
    const float* mr = M + ((size_t)b * N + row) * ldm;
    const bool gi = g[row] == 1.0f;
    float fp = 0.f, fn = 0.f;
    auto term = [&](float m, int col) {
      const bool both = gi && col != row && g[col] == 1.0f;
      const float d = both ? m - 1.0f : m;
      if (both) fp = fmaf(d, d, fp); else fn = fmaf(d, d, fn);
    };
    if ((ldm & 3) == 0) {                          // every row starts on a 16-byte boundary
      const f32x4* m4 = reinterpret_cast<const f32x4*>(mr);
      for (int c4 = threadIdx.x; c4 < (N >> 2); c4 += blockDim.x) {
        const f32x4 v = __builtin_nontemporal_load(m4 + c4);
        term(v[0], 4 * c4); term(v[1], 4 * c4 + 1); term(v[2], 4 * c4 + 2); term(v[3], 4 * c4 + 3);
      }
      for (int col = (N & ~3) + threadIdx.x; col < N; col += blockDim.x) term(mr[col], col);
    } else {
      for (int col = threadIdx.x; col < N; col += blockDim.x) term(mr[col], col);
    }
    sp += (double)fp;
    sn += (double)fn;
  }
  const double tp = block_sum(sp, red), tn = block_sum(sn, red);
  if (threadIdx.x == 0) {
    const size_t wg = (size_t)b * gridDim.x + blockIdx.x;
    part[2 * wg] = tp;
    part[2 * wg + 1] = tn;
  }
}

// one workgroup per pair: inlier count, the partial sums in a fixed tree order, the pair's term (loss.py:131-139)
__global__ void __launch_bounds__(256)
k_sm_pair(const double* __restrict__ part, const float* __restrict__ gt, int N, int parts_per_pair, int balanced,
          double* __restrict__ pair_loss) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  double c = 0.0, sp = 0.0, sn = 0.0;
  for (int k = threadIdx.x; k < N; k += blockDim.x) c += gt[(size_t)b * N + k] == 1.0f ? 1.0 : 0.0;
  for (int k = threadIdx.x; k < parts_per_pair; k += blockDim.x) {
    sp += part[2 * ((size_t)b * parts_per_pair + k)];
    sn += part[2 * ((size_t)b * parts_per_pair + k) + 1];
  }
  const double P = block_sum(c, red), SP = block_sum(sp, red), SN = block_sum(sn, red);
  if (threadIdx.x == 0) {
    if (balanced) {
      const float cp = (float)(P * (P - 1.0)), cn = (float)((double)N * N - P * (P - 1.0));
      const float lp = (float)SP / (fmaxf(cp - 1.0f, 0.0f) + 1.0f);
      const float ln = (float)SN / (fmaxf(cn - 1.0f, 0.0f) + 1.0f);
      pair_loss[b] = (double)(lp * 0.5f + ln * 0.5f);
    } else {
      pair_loss[b] = SP + SN;
    }
  }
}

// out[0] = (v[0] + v[1] + ... in index order) / denom
__global__ void __launch_bounds__(256)
k_mean_pairs(const double* __restrict__ v, int n, double denom, float* __restrict__ out) {
  __shared__ double buf[256];
  double acc = 0.0;
  for (int base = 0; base < n; base += 256) {
    __syncthreads();
    if (base + (int)threadIdx.x < n) buf[threadIdx.x] = v[base + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < min(256, n - base); ++k) acc += buf[k];
  }
  if (threadIdx.x == 0) out[0] = (float)(acc / denom);
}

// ClassificationLoss (loss.py:67-113).  BCE-with-logits as torch evaluates it:
// (1 - y) x + (1 + (pw - 1) y) (log1p(exp(-|x|)) + max(-x, 0)); nine sums per workgroup, combined by k_classification_final.
constexpr int kClsSums = 9;

__global__ void __launch_bounds__(256)
k_classification_partial(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ weight,
                         long total, int N, double* __restrict__ part) {
  __shared__ double red[4];
  double acc[kClsSums] = {0, 0, 0, 0, 0, 0, 0, 0, 0};    // A, B, W, n_pos, sum x y, sum x (1 - y), tp, fp, fn
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const float x = pred[k], y = gt[k];
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);
    const float base = (1.0f - y) * x + sp;
    acc[0] += (double)base;
    acc[1] += (double)(y * sp);
    if (weight) acc[2] += (double)(base * weight[k]);
    acc[3] += (double)y;
    acc[4] += (double)(x * y);
    acc[5] += (double)(x * (1.0f - y));
    if (k < N) {                                   // precision / recall / f1 are those of pair 0 (loss.py:99-101)
      const bool g = y > 0.5f, p = x > 0.0f;
      acc[6] += (g && p) ? 1.0 : 0.0;
      acc[7] += (!g && p) ? 1.0 : 0.0;
      acc[8] += (g && !p) ? 1.0 : 0.0;
    }
  }
#pragma unroll
  for (int j = 0; j < kClsSums; ++j) {
    const double t = block_sum(acc[j], red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * kClsSums + j] = t;
  }
}

// out = loss, precision, recall, f1, logit_true, logit_false
__global__ void __launch_bounds__(256)
k_classification_final(const double* __restrict__ part, int G, long total, int has_weight, int balanced,
                       float* __restrict__ out) {
  __shared__ double red[4];
  double sum[kClsSums];
#pragma unroll
  for (int j = 0; j < kClsSums; ++j) {
    double a = 0.0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) a += part[(size_t)g * kClsSums + j];
    sum[j] = block_sum(a, red);
  }
  if (threadIdx.x == 0) {
    const double A = sum[0], Bs = sum[1], W = sum[2], NY = sum[3], PY = sum[4], PN = sum[5], TP = sum[6], FP = sum[7], FN = sum[8];
    const float num_pos = fmaxf((float)NY - 1.0f, 0.0f) + 1.0f;
    const float num_neg = fmaxf((float)((double)total - NY) - 1.0f, 0.0f) + 1.0f;
    double loss;
    if (has_weight) loss = W / (double)total;
    else if (!balanced) loss = A / (double)total;
    else loss = (A + ((double)(num_neg / num_pos) - 1.0) * Bs) / (double)total;
    out[0] = (float)loss;
    out[1] = (TP + FP > 0.0) ? (float)(TP / (TP + FP)) : 0.0f;
    out[2] = (TP + FN > 0.0) ? (float)(TP / (TP + FN)) : 0.0f;
    out[3] = (2.0 * TP + FP + FN > 0.0) ? (float)(2.0 * TP / (2.0 * TP + FP + FN)) : 0.0f;
    out[4] = (float)(PY / fmax(1.0, NY));
    out[5] = (float)(PN / fmax(1.0, (double)total - NY));
  }
}

// TransformationLoss (loss.py:12-64), grid = (slices, B): pair i's points n = slice, slice + S, ...  As the reference,
// pair i's warped source points are compared with the target points of EVERY pair b (its `warp_src_keypts - tgt_keypts`
// broadcasts [N,3] against [bs,N,3]).  part[i][slice] = sum |d|, sum |d|^2, any(prob > 0).
__global__ void __launch_bounds__(256)
k_transformation_partial(const float* __restrict__ trans, const float* __restrict__ src, const float* __restrict__ tgt,
                         const float* __restrict__ probs, int B, int N, double* __restrict__ part) {
  __shared__ double red[4];
  const int p = blockIdx.y;
  const float* T = trans + (size_t)p * 16;
  const float r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3], r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
  const float r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
  double sd = 0.0, sq = 0.0, any = 0.0;
  for (int nn = blockIdx.x * blockDim.x + threadIdx.x; nn < N; nn += gridDim.x * blockDim.x) {
    const float* s = src + ((size_t)p * N + nn) * 3;
    const float wx = r00 * s[0] + r01 * s[1] + r02 * s[2] + t0;
    const float wy = r10 * s[0] + r11 * s[1] + r12 * s[2] + t1;
    const float wz = r20 * s[0] + r21 * s[1] + r22 * s[2] + t2;
    float fd = 0.f, fq = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* q = tgt + ((size_t)b * N + nn) * 3;
      const float dx = wx - q[0], dy = wy - q[1], dz = wz - q[2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      fd += sqrtf(d2);
      fq += d2;
      if ((b & 15) == 15) { sd += (double)fd; sq += (double)fq; fd = 0.f; fq = 0.f; }
    }
    sd += (double)fd;
    sq += (double)fq;
    if (probs[(size_t)p * N + nn] > 0.0f) any = 1.0;
  }
  const double SD = block_sum(sd, red), SQ = block_sum(sq, red), ANY = block_sum(any, red);
  if (threadIdx.x == 0) {
    double* o = part + ((size_t)p * gridDim.x + blockIdx.x) * 3;
    o[0] = SD; o[1] = SQ; o[2] = ANY;
  }
}

// one workgroup: per-pair errors (thread p), then the batch means in pair order.  out = loss, recall (%), RE, TE, RMSE
__global__ void __launch_bounds__(256)
k_transformation_final(const float* __restrict__ trans, const float* __restrict__ gt_trans, const double* __restrict__ part,
                       int B, int N, int S, float re_thre, float te_thre, float* __restrict__ out) {
  __shared__ double buf[256][5];
  double acc[5] = {0, 0, 0, 0, 0};
  for (int base = 0; base < B; base += 256) {
    const int p = base + threadIdx.x;
    __syncthreads();
    if (p < B) {
      const float* T = trans + (size_t)p * 16;
      const float* G = gt_trans + (size_t)p * 16;
      float tr = 0.f;
      for (int j = 0; j < 3; ++j) {
        float d = 0.f;
        for (int k = 0; k < 3; ++k) d += T[4 * k + j] * G[4 * k + j];      // (R^T G)[j][j]
        tr += d;
      }
      const float c = fminf(fmaxf((tr - 1.0f) / 2.0f, -1.0f), 1.0f);
      const float re = acosf(c) * 180.0f / 3.14159265358979323846f;
      const float e0 = T[3] - G[3], e1 = T[7] - G[7], e2 = T[11] - G[11];
      const float te = sqrtf(e0 * e0 + e1 * e1 + e2 * e2) * 100.0f;
      double SD = 0.0, SQ = 0.0, ANY = 0.0;
      for (int s = 0; s < S; ++s) {
        const double* o = part + ((size_t)p * S + s) * 3;
        SD += o[0]; SQ += o[1]; ANY += o[2];
      }
      const double cnt = (double)B * N;
      buf[threadIdx.x][0] = ANY > 0.0 ? SQ / cnt : 0.0;
      buf[threadIdx.x][1] = (te < te_thre && re < re_thre) ? 1.0 : 0.0;
      buf[threadIdx.x][2] = re;
      buf[threadIdx.x][3] = te;
      buf[threadIdx.x][4] = SD / cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < min(256, B - base); ++k)
        for (int j = 0; j < 5; ++j) acc[j] += buf[k][j];
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(acc[0] / B);
    out[1] = (float)(acc[1] * 100.0 / B);
    out[2] = (float)(acc[2] / B);
    out[3] = (float)(acc[3] / B);
    out[4] = (float)(acc[4] / B);
  }
}

// =========================================================================================
// Training path, first backward slice: gradient of the spectral-matching loss through the
// feature-similarity matrix,
//     M = clamp(1 - (1 - Fn Fn^T) / sigma^2, 0, 1), zero diagonal            GMF_PointDSC/models/PointDSC.py:231-234
//     loss = SpectralMatchingLoss(M, gt_labels)                              GMF_PointDSC/libs/loss.py:116-140
// with respect to the unit features Fn [B, N, 128] and the learned bandwidth sigma (PointDSC.py:164) - what
// `loss.backward()` in libs/trainer.py:158 pushes into the encoder through this term.  As in the forward, M never exists:
//     dL/dM_ij = gtM_ij (M_ij - 1) cP + (1 - gtM_ij) M_ij cN       cP = 1 / (B Dp), cN = 1 / (B Dn)   (balanced form;
//                                                                  MSE form: cP = cN = 2 / (B N^2))
//     G_ij     = dL/dM_ij [0 <= u_ij <= 1] / sigma^2               u = 1 - (1 - s) / sigma^2, s = <f_i, f_j>  (clamp's gradient mask)
//     dL/dFn   = (G + G^T) Fn = 2 G Fn                             (G is symmetric)
//     dL/dsigma = sum_ij dL/dM_ij [0 <= u_ij <= 1] 2 (1 - s_ij) / sigma^3
// i.e. the structure of the attention kernel with K = V = Fn and P = G: per 32 x 32 tile S^T = F_J F_I^T and
// dF_I^T += F_J^T G^T on the f16 MFMA with split-fp16 operands; G is scaled by a power of two per pair so that its fp16
// planes stay in the normal range.  One launch: 1 024 N^2 MFMA flops per pair, the size of one attention launch.
// =========================================================================================
// T image of the features for the second product: unit ((plane * 8 + slot) * 64 + lane), slot = 2 db + s2, lane (h, d): the 8
// halves are Fn[32 t + 16 s2 + 4 h + e][32 db + d] (e < 4) followed by Fn[32 t + 16 s2 + 8 + 4 h + e][32 db + d] - the key order
// in which a lane of the kernel below holds its 16 values of G.
__global__ void k_pack_rows_t_h2(const float* __restrict__ feat, float* __restrict__ img, int N, int tiles, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, slot = (idx >> 6) & 7;
  const long bt = idx >> 9;
  const int tile = bt % tiles;
  const long b = bt / tiles;
  const int h = lane >> 5, d = 32 * (slot >> 1) + (lane & 31), s2 = slot & 1;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = 32 * tile + 16 * s2 + 8 * (e >> 2) + 4 * h + (e & 3);
    v[e] = row < N ? feat[((size_t)b * N + row) * kFeat + d] : 0.f;
  }
  f16x8 hi, lo;
  split8h(v, hi, lo);
  f16x8* o = reinterpret_cast<f16x8*>(img + (size_t)bt * kTileFloats);
  o[(0 * 8 + slot) * 64 + lane] = hi;
  o[(1 * 8 + slot) * 64 + lane] = lo;
}

// per pair: cP, cN (see above), and the power-of-two scale of G; consts[b] = {cP, cN, scale, 1 / scale}
__global__ void __launch_bounds__(256)
k_sm_bwd_prep(const float* __restrict__ gt, float* __restrict__ consts, int B, int N, int balanced, float inv_sig2,
              const float* __restrict__ sigma_dev) {
  inv_sig2 = sigma_inv2(inv_sig2, sigma_dev);
  __shared__ double red[4];
  const int b = blockIdx.x;
  double n1 = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) n1 += (gt[(size_t)b * N + i] == 1.0f) ? 1.0 : 0.0;
  n1 = block_sum(n1, red);
  if (threadIdx.x == 0) {
    const double np = n1 * n1 - n1, nn = (double)N * N - np;       // entries of gt_M that are 1 / 0 (the diagonal is 0)
    double cP, cN;
    if (balanced) { cP = 1.0 / ((double)B * (fmax(np - 1.0, 0.0) + 1.0)); cN = 1.0 / ((double)B * (fmax(nn - 1.0, 0.0) + 1.0)); }
    else cP = cN = 2.0 / ((double)B * (double)N * (double)N);
    const double gmax = fmax(cP, cN) * inv_sig2;                   // |G| <= gmax
    int e;
    frexp(gmax, &e);                                               // gmax = m 2^e, m in [0.5, 1)
    const float scale = ldexpf(1.0f, 8 - e);                       // scaled |G| <= 256
    consts[4 * b + 0] = (float)cP; consts[4 * b + 1] = (float)cN; consts[4 * b + 2] = scale; consts[4 * b + 3] = 1.0f / scale;
  }
}

// grid (ceil(tiles / 4), B), block 256: a wave owns 32 rows i of pair b and sweeps all key tiles.
// dsig_part[b * gridDim.x + blockIdx.x] = this workgroup's part of dL/dsigma (fp64).
__global__ void __launch_bounds__(256, 2)
k_sm_backward(const float* __restrict__ img, const float* __restrict__ timg, const float* __restrict__ gt,
              const float* __restrict__ consts, float* __restrict__ dF, double* __restrict__ dsig_part, int N, int tiles,
              float inv_sig2, float two_inv_sig3, const float* __restrict__ sigma_dev) {
  inv_sig2 = sigma_inv2(inv_sig2, sigma_dev);
  two_inv_sig3 = sigma_two_inv3(two_inv_sig3, sigma_dev);
  __shared__ __attribute__((aligned(16))) float lds[4 * kTileFloats];      // K ring [2] | V ring [2]
  __shared__ double red[4];
  float* const ldsK = lds;
  float* const ldsV = lds + 2 * kTileFloats;
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * 4 + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t pbase = (size_t)pair * tiles;
  const float cP = consts[4 * pair + 0], cN = consts[4 * pair + 1], scale = consts[4 * pair + 2], inv_scale = consts[4 * pair + 3];
  const int row_i = 32 * tile + i;
  const float* gtp = gt + (size_t)pair * N;
  const float gt_i = row_i < N ? gtp[row_i] : 0.f;

  f16x8 qh[8], ql[8];
  {
    const f16x8* qp = reinterpret_cast<const f16x8*>(img + (pbase + tile) * (size_t)kTileFloats) + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) { qh[s] = qp[(0 * 8 + s) * 64]; ql[s] = qp[(1 * 8 + s) * 64]; }
  }
  const float* gk = img + pbase * (size_t)kTileFloats;
  const float* gv = timg + pbase * (size_t)kTileFloats;
  auto issue = [&](int t) {
    dma_issue(gk + (size_t)t * kTileFloats, ldsK + (t & 1) * kTileFloats, 16, wave, 4, lane);
    dma_issue(gv + (size_t)t * kTileFloats, ldsV + (t & 1) * kTileFloats, 16, wave, 4, lane);
  };
  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  double dsig = 0.0;
  issue(0);
  for (int t = 0; t < tiles; ++t) {
    // the labels of this lane's 16 keys are requested BEFORE the stage wait (a load after it would queue behind the next
    // tile's DMA pieces)
    float gtj[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
      gtj[r] = j < N ? gtp[j] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < tiles) issue(t + 1);
    const f16x8* lk = reinterpret_cast<const f16x8*>(ldsK + (t & 1) * kTileFloats) + lane;
    const f16x8* lv = reinterpret_cast<const f16x8*>(ldsV + (t & 1) * kTileFloats) + lane;
    f32x16 sacc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(sacc, lk[(0 * 8 + s) * 64], lk[(1 * 8 + s) * 64], qh[s], ql[s]);
    float g[16];
    float ds_local = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
      const float sv = sacc[r];
      const float u = 1.0f - (1.0f - sv) * inv_sig2;
      const bool diag = (j == row_i);
      const float M = diag ? 0.f : fminf(fmaxf(u, 0.f), 1.f);
      const bool both = (gt_i == 1.0f) && (gtj[r] == 1.0f) && !diag;
      const float dM = both ? (M - 1.0f) * cP : M * cN;
      const bool inside = (u >= 0.f) && (u <= 1.f) && !diag && (j < N) && (row_i < N);
      const float gm = inside ? dM : 0.f;
      g[r] = gm * inv_sig2 * scale;
      ds_local = fmaf(gm, (1.0f - sv) * two_inv_sig3, ds_local);
    }
    dsig += (double)ds_local;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&g[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int slot = 2 * db + s2;
        mma3(oacc[db], lv[(0 * 8 + slot) * 64], lv[(1 * 8 + slot) * 64], ph, pl);
      }
    }
  }
  // dF_i = 2 G F: lane (h, i) register r of block db is feature 32 db + 8 (r >> 2) + 4 h + (r & 3) of row i
  if (active && row_i < N) {
    float4* o = reinterpret_cast<float4*>(dF + ((size_t)pair * N + row_i) * kFeat);
    const float c2 = 2.0f * inv_scale;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o[8 * db + 2 * q + h] = make_float4(oacc[db][4 * q] * c2, oacc[db][4 * q + 1] * c2, oacc[db][4 * q + 2] * c2, oacc[db][4 * q + 3] * c2);
  }
  if (!active) dsig = 0.0;
  const double tot = block_sum(dsig, red);
  if (threadIdx.x == 0) dsig_part[(size_t)pair * gridDim.x + blockIdx.x] = tot;
}

// dsigma = sum of the partials in index order (deterministic)
__global__ void k_sm_bwd_dsigma(const double* __restrict__ part, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += part[k];
    out[0] = (float)s;
  }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
static void similarity_grid(int B, int tiles, dim3& grid, int& chunk) {
  const int rg = (tiles + 3) / 4;
  int nchunks = (2048 + rg * B - 1) / (rg * B);           // >= ~4 rounds of the 512 workgroup slots
  nchunks = std::max(1, std::min(nchunks, std::max(1, tiles / 8)));
  chunk = std::min(64, (tiles + nchunks - 1) / nchunks);    // <= 64: the LDS table of column inlier bits
  nchunks = (tiles + chunk - 1) / chunk;
  grid = dim3(rg, nchunks, B);
}

size_t similarity_image_floats(int B, int N) { return (size_t)B * ((N + 31) / 32) * kTileFloats; }

int sm_fused_parts_per_pair(int B, int N) {
  dim3 g; int chunk;
  similarity_grid(B, (N + 31) / 32, g, chunk);
  return (int)(g.x * g.y);
}

hipError_t launch_similarity_matrix(const float* feat_n, float* img, float* M, int B, int N, int ldm, float sigma,
                                    hipStream_t s, const float* sigma_dev) {
  const int tiles = (N + 31) / 32;
  const long total = (long)B * tiles * 512;
  hipLaunchKernelGGL(k_pack_rows_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat_n, img, N, tiles, total);
  dim3 grid; int chunk;
  similarity_grid(B, tiles, grid, chunk);
  const float s2 = sigma * sigma;
  hipLaunchKernelGGL(k_similarity<false>, grid, dim3(320), 0, s, img, M, (const float*)nullptr, (double*)nullptr, N, ldm,
                     tiles, chunk, 1.0f / s2, sigma_dev);
  return hipGetLastError();
}

hipError_t launch_sm_loss_fused(const float* feat_n, const float* gt, float* img, double* part, double* pair_loss, int B,
                                int N, float sigma, int balanced, float* out, hipStream_t s, const float* sigma_dev) {
  const int tiles = (N + 31) / 32;
  const long total = (long)B * tiles * 512;
  hipLaunchKernelGGL(k_pack_rows_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat_n, img, N, tiles, total);
  dim3 grid; int chunk;
  similarity_grid(B, tiles, grid, chunk);
  const float s2 = sigma * sigma;
  hipLaunchKernelGGL(k_similarity<true>, grid, dim3(320), 0, s, img, (float*)nullptr, gt, part, N, N, tiles, chunk, 1.0f / s2, sigma_dev);
  hipLaunchKernelGGL(k_sm_pair, dim3(B), dim3(256), 0, s, part, gt, N, (int)(grid.x * grid.y), balanced, pair_loss);
  hipLaunchKernelGGL(k_mean_pairs, dim3(1), dim3(256), 0, s, pair_loss, B, balanced ? (double)B : (double)B * N * N, out);
  return hipGetLastError();
}

int sm_parts_per_pair(int B, int N) { return std::max(1, std::min(N, (2048 + B - 1) / B)); }

hipError_t launch_sm_loss(const float* M, int ldm, const float* gt, double* part, double* pair_loss, int B, int N,
                          int balanced, float* out, hipStream_t s) {
  const int G = sm_parts_per_pair(B, N);
  hipLaunchKernelGGL(k_sm_partial, dim3(G, B), dim3(256), 0, s, M, gt, part, N, ldm);
  hipLaunchKernelGGL(k_sm_pair, dim3(B), dim3(256), 0, s, part, gt, N, G, balanced, pair_loss);
  hipLaunchKernelGGL(k_mean_pairs, dim3(1), dim3(256), 0, s, pair_loss, B, balanced ? (double)B : (double)B * N * N, out);
  return hipGetLastError();
}

int classification_parts(int B, int N) { return (int)std::max<long>(1, std::min<long>(256, ((long)B * N + 1023) / 1024)); }

hipError_t launch_classification_loss(const float* pred, const float* gt, const float* weight, double* part, int B, int N,
                                      int balanced, float* out, hipStream_t s) {
  const int G = classification_parts(B, N);
  const long total = (long)B * N;
  hipLaunchKernelGGL(k_classification_partial, dim3(G), dim3(256), 0, s, pred, gt, weight, total, N, part);
  hipLaunchKernelGGL(k_classification_final, dim3(1), dim3(256), 0, s, part, G, total, weight ? 1 : 0, balanced, out);
  return hipGetLastError();
}

int transformation_slices(int B, int N) { return std::max(1, std::min((N + 255) / 256, (512 + B - 1) / B)); }

hipError_t launch_transformation_loss(const float* trans, const float* gt_trans, const float* src, const float* tgt,
                                      const float* probs, double* part, int B, int N, float re_thre, float te_thre,
                                      float* out, hipStream_t s) {
  const int S = transformation_slices(B, N);
  hipLaunchKernelGGL(k_transformation_partial, dim3(S, B), dim3(256), 0, s, trans, src, tgt, probs, B, N, part);
  hipLaunchKernelGGL(k_transformation_final, dim3(1), dim3(256), 0, s, trans, gt_trans, part, B, N, S, re_thre, te_thre, out);
  return hipGetLastError();
}

int sm_backward_parts(int B, int N) { return B * ((((N + 31) / 32) + 3) / 4); }

hipError_t launch_sm_backward(const float* feat_n, const float* gt, float* img, float* timg, float* consts, double* dsig_part,
                              int B, int N, float sigma, int balanced, float* dF, float* dsigma, hipStream_t s, const float* sigma_dev) {
  const int tiles = (N + 31) / 32;
  const long total = (long)B * tiles * 512;
  const float inv_sig2 = 1.0f / (sigma * sigma);
  hipLaunchKernelGGL(k_pack_rows_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat_n, img, N, tiles, total);
  hipLaunchKernelGGL(k_pack_rows_t_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat_n, timg, N, tiles, total);
  hipLaunchKernelGGL(k_sm_bwd_prep, dim3(B), dim3(256), 0, s, gt, consts, B, N, balanced, inv_sig2, sigma_dev);
  const dim3 grid((tiles + 3) / 4, B);
  hipLaunchKernelGGL(k_sm_backward, grid, dim3(256), 0, s, img, timg, gt, consts, dF, dsig_part, N, tiles, inv_sig2,
                     2.0f / (sigma * sigma * sigma), sigma_dev);
  hipLaunchKernelGGL(k_sm_bwd_dsigma, dim3(1), dim3(64), 0, s, dsig_part, (int)(grid.x * grid.y), dsigma);
  return hipGetLastError();
}

}  // namespace gmf
