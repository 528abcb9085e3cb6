// Pose head of GMF-PointDSC and the DGR weighted-Procrustes solve on gfx950.
//
// Replaces (file:line relative to /root/reference/):
//   k_nms_keys          GMF_PointDSC/models/PointDSC.py:268-285   parallel NMS -> score * is_local_max
//   k_sort_topk         GMF_PointDSC/models/PointDSC.py:246,286   argsort(descending)[:S]
//   k_knn_seeds         GMF_PointDSC/models/common.py:53-75 + PointDSC.py:327-329  kNN rows of the seeds only
//   k_seed_power        GMF_PointDSC/models/PointDSC.py:335-361,437-448  seed compatibility matrices + power iteration
//   k_seed_kabsch       GMF_PointDSC/models/PointDSC.py:364-365,405; models/common.py:10-50  weights + weighted Kabsch
//   k_score_hyp         GMF_PointDSC/models/PointDSC.py:413-417   inlier count of every hypothesis
//   k_finalize_pose     GMF_PointDSC/models/PointDSC.py:419-425,493-528  argmax, labels, IRLS post-refinement
//   k_rigid_transform   GMF_PointDSC/models/common.py:10-50       batched weighted Kabsch (public op)
//   k_weighted_procrustes  GMF_DeepGlobalRegistration/*/core/registration.py:91-113
//
// These stages are HBM/latency bound integer-and-float work (no MFMA).  The reference moves every
// 3x3 covariance to the CPU for LAPACK (common.py:40-41, registration.py:105); here the SVD is a
// one-sided Jacobi in fp64 registers, one lane per matrix, with no host round trip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <algorithm>
#include <type_traits>

#include "launchers_pose.hpp"
#include "mfma_core.hpp"
#include "kabsch.hpp"

namespace gmf {

#define GMF_DEVINL __device__ __forceinline__

// Ragged batches (PairTab, launchers.hpp): the rows of pair `pair` in the caller's packed tensors, its own N and seed count.
// Library-internal per-pair buffers keep the strides of the LARGEST pair (the kernel's N / S arguments).
GMF_DEVINL int pair_seeds(const PairTab* pt, int pair, int S) { return pt ? pt[pair].S : S; }

// T (row-major 4x4 fp32) from R (double) and centroids:  t = cB - R cA   (common.py:46-50)
GMF_DEVINL void write_T(float* T, const double* R, const double* ca, const double* cb) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float r0 = (float)R[3 * r], r1 = (float)R[3 * r + 1], r2 = (float)R[3 * r + 2];
    T[4 * r + 0] = r0; T[4 * r + 1] = r1; T[4 * r + 2] = r2;
    T[4 * r + 3] = (float)cb[r] - (r0 * (float)ca[0] + r1 * (float)ca[1] + r2 * (float)ca[2]);
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// ---- reductions --------------------------------------------------------------------------
GMF_DEVINL double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NV>
GMF_DEVINL void block_sum(double (&v)[NV], double* sh /* >= NV*16 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
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

// block_sum for the inner loop of k_global_registration: after the wave sums, lanes 0..NV-1 of wave 0 add the per-wave
// partials (NV parallel chains of nw LDS reads) and every thread then reads the NV totals - block_sum has every thread
// add all nw x NV partials itself (208 LDS reads per thread for 13 values and 16 waves).
template <int NV>
GMF_DEVINL void block_sum_tree(double (&v)[NV], double* sh /* >= NV*17 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  // butterfly over the wave with all NV exchanges of a step in flight together (NV independent ds_bpermute pairs, one wait)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double t[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) t[k] = __shfl_xor(v[k], o, 64);
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] += t[k];
  }
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) sh[k * 16 + wave] = v[k];
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0;
    for (int w = 0; w < nw; ++w) s += sh[threadIdx.x * 16 + w];
    sh[NV * 16 + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = sh[NV * 16 + k];
}

// ---------------------------------------------------------------------------------------
// NMS keys: key_i = score_i * [ for all j: score_i >= score_j  or  ||src_i - src_j|| >= R ]
// grid (ceil(N/256), B, JS): the candidates j are divided over JS workgroups so that small batches fill the chip (at B = 1
// a thread otherwise walks all N candidates alone: 0.4 ms at N = 5000).  With JS > 1 `keys` is pre-set to the scores by the
// launcher and a workgroup only writes zeros for the points its candidate range suppresses (the conjunction over ranges
// needs no ordering; score * 0 keeps the reference's sort order whatever the sign of the zero).
// ---------------------------------------------------------------------------------------
typedef float nms_f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256)
k_nms_keys(const float* __restrict__ src, const float* __restrict__ scores, float* __restrict__ keys, int N, float R2t,
           const PairTab* __restrict__ ptab) {
  // candidates of a block as structure-of-arrays so that two of them load as one register pair: the squared distance of a
  // point to TWO candidates is 6 packed-fp32 instructions (v_pk_add/mul/fma_f32) with the same roundings as the scalar form
  __shared__ __attribute__((aligned(8))) float sx[256], sy[256], sz[256], sw[256];
  const int pair = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const size_t row0 = pair_row0(ptab, pair, N);
  N = pair_rows(ptab, pair, N);
  if ((int)blockIdx.x * 256 >= N) return;
  const float* ps = src + row0 * 3;
  const float* sc = scores + row0;
  float xi = 0, yi = 0, zi = 0, si = 0;
  if (i < N) { xi = ps[3 * i]; yi = ps[3 * i + 1]; zi = ps[3 * i + 2]; si = sc[i]; }
  const nms_f2 xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
  bool is_max = true;
  const int nblk = (N + 255) / 256, js = gridDim.z;
  const int b0 = (nblk * (int)blockIdx.z) / js, b1 = (nblk * ((int)blockIdx.z + 1)) / js;
  for (int j0 = b0 * 256; j0 < min(N, b1 * 256); j0 += 256) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    // padding candidates: score -inf never suppresses anything
    sx[threadIdx.x] = (j < N) ? ps[3 * j] : 0.f;
    sy[threadIdx.x] = (j < N) ? ps[3 * j + 1] : 0.f;
    sz[threadIdx.x] = (j < N) ? ps[3 * j + 2] : 0.f;
    sw[threadIdx.x] = (j < N) ? sc[j] : -INFINITY;
    __syncthreads();
    const int lim = (min(256, N - j0) + 1) & ~1;
    for (int jj = 0; jj < lim; jj += 2) {
      const nms_f2 px = *reinterpret_cast<const nms_f2*>(&sx[jj]), py = *reinterpret_cast<const nms_f2*>(&sy[jj]);
      const nms_f2 pz = *reinterpret_cast<const nms_f2*>(&sz[jj]), pw = *reinterpret_cast<const nms_f2*>(&sw[jj]);
      const nms_f2 dx = xi2 - px, dy = yi2 - py, dz = zi2 - pz;
      const nms_f2 d2 = dx * dx + dy * dy + dz * dz;
      is_max = is_max && ((si >= pw[0]) || (d2[0] >= R2t)) && ((si >= pw[1]) || (d2[1] >= R2t));   // see launch_nms_keys
    }
  }
  if (i < N) {
    if (js == 1) keys[row0 + i] = si * (is_max ? 1.f : 0.f);
    else if (!is_max) keys[row0 + i] = si * 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// The same keys with the candidates of a point limited to its 27 neighbouring cells of a uniform grid (cell edge 1.01 R,
// coordinates wrapped to 16 x 16 x 16 cells): every j with ||src_i - src_j|| < R lies in one of them, the others pass the
// test through the distance term, so the conjunction - and every key - is exactly that of the all-pairs kernel.
//   k_nms_bin   (grid B, block 1024): counting sort of the pair's points by cell -> cell_start [4097], sorted (x, y, z, score).
//               A pair with a coordinate beyond 6e4 cells (where the float rounding of x / cell could separate neighbours by
//               two cells) or a non-finite coordinate is flagged: its points test every candidate.
//   k_nms_keys_binned (grid (ceil(N/256), B)): one point per thread.
// scratch per pair: 4 N floats (sorted points) + 4104 ints (cell_start | flag), see nms_scratch_floats().
// ---------------------------------------------------------------------------------------
constexpr int kNmsCells = 4096, kNmsHdr = 4104;

GMF_DEVINL int nms_cell_coord(float x, float inv_cell) { return ((int)floorf(x * inv_cell)) & 15; }

__global__ void __launch_bounds__(1024)
k_nms_bin(const float* __restrict__ src, const float* __restrict__ scores, float* __restrict__ scratch, int N, float inv_cell,
          const PairTab* __restrict__ ptab) {
  __shared__ unsigned cnt[kNmsCells];
  __shared__ unsigned wtot[16];
  __shared__ int bad;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4* sorted = reinterpret_cast<float4*>(scratch + (size_t)pair * ((size_t)4 * N + kNmsHdr));     // (slot of the largest pair)
  int* hdr = reinterpret_cast<int*>(sorted + N);
  const size_t row0 = pair_row0(ptab, pair, N);
  N = pair_rows(ptab, pair, N);
  const float* ps = src + row0 * 3;
  const float* sc = scores + row0;
  for (int c = tid; c < kNmsCells; c += 1024) cnt[c] = 0;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    const float x = ps[3 * i], y = ps[3 * i + 1], z = ps[3 * i + 2];
    const float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))) * inv_cell;
    if (!(m < 6.0e4f)) bad = 1;                     // also catches NaN / inf
    const int key = nms_cell_coord(x, inv_cell) | (nms_cell_coord(y, inv_cell) << 4) | (nms_cell_coord(z, inv_cell) << 8);
    atomicAdd(&cnt[key], 1u);
  }
  __syncthreads();
  // exclusive scan of the 4096 counts: 4 cells per thread
  unsigned c4[4], tsum = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { c4[q] = cnt[4 * tid + q]; tsum += c4[q]; }
  unsigned inc = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  unsigned run = base + inc - tsum;
#pragma unroll
  for (int q = 0; q < 4; ++q) { hdr[4 * tid + q] = (int)run; cnt[4 * tid + q] = run; run += c4[q]; }   // cnt becomes the fill cursor
  if (tid == 1023) hdr[kNmsCells] = (int)run;       // = N
  __syncthreads();
  if (tid == 0) hdr[kNmsCells + 1] = bad;
  for (int i = tid; i < N; i += 1024) {
    const float x = ps[3 * i], y = ps[3 * i + 1], z = ps[3 * i + 2];
    const int key = nms_cell_coord(x, inv_cell) | (nms_cell_coord(y, inv_cell) << 4) | (nms_cell_coord(z, inv_cell) << 8);
    const unsigned pos = atomicAdd(&cnt[key], 1u);
    sorted[pos] = make_float4(x, y, z, sc[i]);
  }
}

__global__ void __launch_bounds__(256)
k_nms_keys_binned(const float* __restrict__ src, const float* __restrict__ scores, const float* __restrict__ scratch,
                  float* __restrict__ keys, int N, float R2t, float inv_cell, const PairTab* __restrict__ ptab) {
  const int pair = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float4* sorted = reinterpret_cast<const float4*>(scratch + (size_t)pair * ((size_t)4 * N + kNmsHdr));
  const int* hdr = reinterpret_cast<const int*>(sorted + N);
  const size_t row0 = pair_row0(ptab, pair, N);
  N = pair_rows(ptab, pair, N);
  if (i >= N) return;
  const float* ps = src + row0 * 3;
  const float xi = ps[3 * i], yi = ps[3 * i + 1], zi = ps[3 * i + 2], si = scores[row0 + i];
  bool is_max = true;
  auto scan = [&](int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
      const float4 p = sorted[j];
      const float dx = xi - p.x, dy = yi - p.y, dz = zi - p.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      is_max = is_max && ((si >= p.w) || (d2 >= R2t));
    }
  };
  if (hdr[kNmsCells + 1]) {
    scan(0, N);
  } else {
    // [r4] The 27 cells, nine at a time: their 18 bounds are requested together, then the first TWO points of every cell (a cell holds
    // ~1.2 points at the reference's settings) - one memory round trip per group instead of one per candidate (the scan used to be a
    // chain of ~33 dependent loads per point: 76 us at 32 x 5000); cells with more than two points finish in the loop.  The conjunction
    // over candidates needs no order: every key is that of the all-pairs kernel.
    const int cx = nms_cell_coord(xi, inv_cell), cy = nms_cell_coord(yi, inv_cell), cz = nms_cell_coord(zi, inv_cell);
    const int last = N - 1;
#pragma unroll
    for (int oz = -1; oz <= 1; ++oz) {
      int j0[9], j1[9];
#pragma unroll
      for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) {
          const int c = ((cx + ox) & 15) | (((cy + oy) & 15) << 4) | (((cz + oz) & 15) << 8);
          j0[3 * (oy + 1) + ox + 1] = hdr[c];
          j1[3 * (oy + 1) + ox + 1] = hdr[c + 1];
        }
      float4 pa[9], pb[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) { pa[q] = sorted[min(j0[q], last)]; pb[q] = sorted[min(j0[q] + 1, last)]; }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (j0[q] < j1[q]) {
          const float dx = xi - pa[q].x, dy = yi - pa[q].y, dz = zi - pa[q].z;
          is_max = is_max && ((si >= pa[q].w) || (dx * dx + dy * dy + dz * dz >= R2t));
        }
        if (j0[q] + 1 < j1[q]) {
          const float dx = xi - pb[q].x, dy = yi - pb[q].y, dz = zi - pb[q].z;
          is_max = is_max && ((si >= pb[q].w) || (dx * dx + dy * dy + dz * dz >= R2t));
        }
        if (j0[q] + 2 < j1[q]) scan(j0[q] + 2, j1[q]);
      }
    }
  }
  keys[row0 + i] = si * (is_max ? 1.f : 0.f);
}

// ---------------------------------------------------------------------------------------
// Per pair: indices of the S largest keys, ordered (key descending, index ascending) - the order
// torch's stable CPU sort gives argsort(descending=True).  Bitonic sort in LDS, N <= 16384.
// grid (B), block 1024, dynamic LDS = M*8 bytes (M = next pow2 >= N)
// ---------------------------------------------------------------------------------------
GMF_DEVINL bool key_before(float ka, int ia, float kb, int ib) { return (ka > kb) || (ka == kb && ia < ib); }

__global__ void __launch_bounds__(1024)
k_sort_topk(const float* __restrict__ keys, int* __restrict__ out_idx, int N, int M, int S, const PairTab* __restrict__ ptab) {
  extern __shared__ unsigned char smem_raw[];
  float* sk = reinterpret_cast<float*>(smem_raw);
  int* si = reinterpret_cast<int*>(smem_raw + (size_t)M * 4);
  const int pair = blockIdx.x;
  const size_t row0 = pair_row0(ptab, pair, N);
  const int S_out = S;                                  // (stride of the output list: the largest pair's)
  N = pair_rows(ptab, pair, N);
  S = pair_seeds(ptab, pair, S);
  for (int t = threadIdx.x; t < M; t += blockDim.x) {
    sk[t] = (t < N) ? keys[row0 + t] : -INFINITY;
    si[t] = (t < N) ? t : 0x7fffffff;
  }
  __syncthreads();
  for (int size = 2; size <= M; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (M >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);      // ascending position order = "before" first
        const float ka = sk[lo], kb = sk[hi];
        const int ia = si[lo], ib = si[hi];
        const bool a_first = key_before(ka, ia, kb, ib);
        if (a_first != up) { sk[lo] = kb; sk[hi] = ka; si[lo] = ib; si[hi] = ia; }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < S; t += blockDim.x) out_idx[(size_t)pair * S_out + t] = si[t];
}

// ---------------------------------------------------------------------------------------
// The same result without sorting all N keys: radix-select the S-th element of the order (key descending, index
// ascending), compact the S winners in index order, then place each by counting the winners before it.
//   d = ~monotone(key) as uint32 (ascending d = descending key; -0 is folded onto +0 so that float equality and bit
//   equality agree); four 8-bit passes of an LDS histogram narrow the threshold; keys equal to the threshold are taken
//   lowest index first.  After NMS most keys are exactly zero (PointDSC.py:285), so zeros are counted per wave with a
//   popcount instead of one LDS atomic each.  ~10 passes over LDS with ~20 barriers instead of the 91 compare-exchange
//   passes of the bitonic network at N = 5000.
// grid (B), block 1024, dynamic LDS = N*4 + S*8 bytes
// ---------------------------------------------------------------------------------------
// KEYS_IN_LDS = false [r4]: the same selection for ANY N (the reference has no limit, PointDSC.py:268-286; its 3DMatch evaluation
// feeds num_node = 'all', evaluation/test_3DMatch.py:143): the transformed keys are not staged in the LDS but re-read from global
// memory (L2-resident: 4 N bytes per pair) in each of the four histogram passes and the two compaction passes; the counters of the
// ordered compaction are two 32-bit fields instead of 16 + 16 bits.  Every decision is the same function of the same keys, so both
// forms give identical seed lists (tests: test_topk_select_equals_full_sort, test_topk_select_any_size).
// dynamic LDS = S*8 bytes (+ N*4 + 8 with the keys in the LDS)
template <bool KEYS_IN_LDS>
__global__ void __launch_bounds__(1024)
k_select_topk(const float* __restrict__ keys, int* __restrict__ out_idx, int N, int S, const PairTab* __restrict__ ptab) {
  extern __shared__ unsigned char smem_raw[];
  unsigned* d = reinterpret_cast<unsigned*>(smem_raw);                                   // [N] (KEYS_IN_LDS)
  unsigned long long* sel = reinterpret_cast<unsigned long long*>(KEYS_IN_LDS ? smem_raw + (((size_t)N * 4 + 7) & ~(size_t)7) : smem_raw);   // [S]
  __shared__ unsigned hist[256];
  __shared__ unsigned wtot[16], wtot2[16];
  __shared__ unsigned s_prefix, s_rank;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr unsigned kZero = 0x7fffffffu;          // d of +0.0
  const size_t row0 = pair_row0(ptab, pair, N);
  const int S_out = S;                             // (stride of the output list: the largest pair's; the LDS was sized by it too)
  N = pair_rows(ptab, pair, N);
  S = pair_seeds(ptab, pair, S);
  auto to_d = [](float f) -> unsigned {
    if (f == 0.0f) f = 0.0f;                       // -0 -> +0
    const unsigned bits = __float_as_uint(f);
    const unsigned u = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    return ~u;
  };
  auto dval = [&](int t) -> unsigned { return KEYS_IN_LDS ? d[t] : to_d(keys[row0 + t]); };
  if (KEYS_IN_LDS) {
    for (int t = tid; t < N; t += 1024) d[t] = to_d(keys[row0 + t]);
    __syncthreads();
  }
  unsigned prefix = 0, mask = 0, r = (unsigned)(S - 1);                                  // 0-based rank still to resolve
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    unsigned zc = 0;
    for (int t = tid; t < N; t += 1024) {
      const unsigned v = dval(t);
      if ((v & mask) == prefix) {
        if (v == kZero) ++zc; else atomicAdd(&hist[(v >> shift) & 255u], 1u);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zc += __shfl_xor(zc, o, 64);
    if (lane == 0 && zc) atomicAdd(&hist[(kZero >> shift) & 255u], zc);
    __syncthreads();
    if (tid < 256) {                               // inclusive scan of the 256 bins: four waves of 64 bins
      unsigned inc = hist[tid];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      if (lane == 63) wtot[wave] = inc;
      hist[tid] = inc;                             // own bin only: no hazard
    }
    __syncthreads();
    if (tid < 256) {
      unsigned base = 0;
      for (int w = 0; w < wave; ++w) base += wtot[w];
      const unsigned own = (lane == 0) ? hist[tid] : hist[tid] - hist[tid - 1];         // count of bin tid
      const unsigned inc = hist[tid] + base, before = inc - own;
      if (before <= r && r < inc) { s_prefix = (unsigned)tid; s_rank = r - before; }
    }
    __syncthreads();
    prefix |= s_prefix << shift;
    mask |= 255u << shift;
    r = s_rank;
    __syncthreads();
  }
  const unsigned thr = prefix, need_eq = r + 1;    // winners: d < thr, and the first need_eq of d == thr by index
  // ordered compaction: thread t owns the contiguous indices [t * per, (t + 1) * per)
  const int per = (N + 1023) / 1024, lo = min(N, tid * per), hi = min(N, lo + per);
  unsigned c_lt = 0, c_eq = 0;
  for (int t = lo; t < hi; ++t) { const unsigned v = dval(t); c_lt += (v < thr) ? 1u : 0u; c_eq += (v == thr) ? 1u : 0u; }
  unsigned i_lt = c_lt, i_eq = c_eq;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned u1 = __shfl_up(i_lt, o, 64), u2 = __shfl_up(i_eq, o, 64);
    if (lane >= o) { i_lt += u1; i_eq += u2; }
  }
  if (lane == 63) { wtot[wave] = i_lt; wtot2[wave] = i_eq; }
  __syncthreads();
  unsigned b_lt = 0, b_eq = 0;
  for (int w = 0; w < wave; ++w) { b_lt += wtot[w]; b_eq += wtot2[w]; }
  unsigned lt_b = b_lt + i_lt - c_lt, eq_b = b_eq + i_eq - c_eq;
  for (int t = lo; t < hi; ++t) {
    const unsigned v = dval(t);
    if (v < thr) {
      sel[lt_b + min(eq_b, need_eq)] = ((unsigned long long)v << 32) | (unsigned)t;
      ++lt_b;
    } else if (v == thr) {
      if (eq_b < need_eq) sel[lt_b + eq_b] = ((unsigned long long)v << 32) | (unsigned)t;
      ++eq_b;
    }
  }
  __syncthreads();
  // rank of every winner among the winners = its output position
  for (int e = tid; e < S; e += 1024) {
    const unsigned long long me = sel[e];
    int rank = 0;
    for (int j = 0; j < S; ++j) rank += sel[j] < me ? 1 : 0;
    out_idx[(size_t)pair * S_out + rank] = (int)(unsigned)(me & 0xffffffffull);
  }
}

// ---------------------------------------------------------------------------------------
// kNN of the seed rows only: for seed s, the k+1 smallest of d_j = 2 - 2 <f_s, f_j>, ascending
// (ties: lower index first), first one dropped (common.py:70-74).  feat_n row-major [B,N,128].
// grid (S, B), block 256, dynamic LDS = N*4 bytes
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_knn_seeds(const float* __restrict__ feat_n, const int* __restrict__ seeds, const float* __restrict__ dist_in,
            int* __restrict__ knn_idx, int N, int S, int k, const PairTab* __restrict__ ptab, int ld) {
  extern __shared__ float dist[];
  __shared__ float fs[128];
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  const int pair = blockIdx.y, s = blockIdx.x;
  if (s >= pair_seeds(ptab, pair, S)) return;
  const float* fb = feat_n + pair_row0(ptab, pair, N) * 128;
  const float* dr0 = dist_in ? dist_in + ((size_t)pair * S + s) * ld : nullptr;    // (strides of the largest pair, rows padded to ld)
  N = pair_rows(ptab, pair, N);
  if (dist_in) {        // distances precomputed by k_seed_dist (MFMA): just stage the row in LDS
    const float* dr = dr0;
    for (int j = threadIdx.x; j < N; j += 256) dist[j] = dr[j];
  } else {
  const int seed = seeds[(size_t)pair * S + s];
  if (threadIdx.x < 128) fs[threadIdx.x] = fb[(size_t)seed * 128 + threadIdx.x];
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += 256) {
    const float4* pj = reinterpret_cast<const float4*>(fb + (size_t)j * 128);
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const float4 v = pj[c];
      acc = fmaf(v.x, fs[4 * c], acc); acc = fmaf(v.y, fs[4 * c + 1], acc);
      acc = fmaf(v.z, fs[4 * c + 2], acc); acc = fmaf(v.w, fs[4 * c + 3], acc);
    }
    dist[j] = 2.0f - 2.0f * acc;
  }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it <= k; ++it) {
    float bv = INFINITY; int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < N; j += 256) {
      const float v = dist[j];
      if (v < bv) { bv = v; bi = j; }          // strided scan keeps the lowest index among equal values per thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = red_v[0]; int ix = red_i[0];
      for (int w = 1; w < 4; ++w) if (red_v[w] < v || (red_v[w] == v && red_i[w] < ix)) { v = red_v[w]; ix = red_i[w]; }
      if (ix >= N) ix = min(it, N - 1);      // (a row of NaNs never wins a comparison: the index list must stay in range all the same)
      if (it > 0) knn_idx[((size_t)pair * S + s) * k + (it - 1)] = ix;
      dist[ix] = INFINITY;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Top-(k+1) selection with the seed's distance row held in registers (EPT values per thread, N <= 256*EPT):
// each thread caches its local minimum; a round is one block-wide argmin over the 256 cached minima and a rescan by
// the single owner of the extracted element.  Same order as k_knn_seeds: ascending distance, lower index first.
// grid (S, B), block 256.
// ---------------------------------------------------------------------------------------
template <int EPT>
__global__ void __launch_bounds__(256)
k_knn_select(const float* __restrict__ dist_in, int* __restrict__ knn_idx, int N, int S, int k, const PairTab* __restrict__ ptab, int ld) {
  __shared__ float red_v[2][4];
  __shared__ int red_i[2][4];
  const int pair = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (s >= pair_seeds(ptab, pair, S)) return;
  const float* dr = dist_in + ((size_t)pair * S + s) * ld;
  N = pair_rows(ptab, pair, N);
  float v[EPT];
#pragma unroll
  for (int m = 0; m < EPT; ++m) { const int j = tid + 256 * m; v[m] = (j < N) ? dr[j] : INFINITY; }
  float bv = INFINITY; int bm = 0;
#pragma unroll
  for (int m = 0; m < EPT; ++m) if (v[m] < bv) { bv = v[m]; bm = m; }
  for (int it = 0; it <= k; ++it) {
    float rv = bv; int ri = (bv < INFINITY) ? tid + 256 * bm : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(rv, o, 64); const int oi = __shfl_xor(ri, o, 64);
      if (ov < rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
    }
    const int buf = it & 1;
    if (lane == 0) { red_v[buf][wave] = rv; red_i[buf][wave] = ri; }
    __syncthreads();
    float wv = red_v[buf][0]; int wi = red_i[buf][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = red_v[buf][w]; const int oi = red_i[buf][w];
      if (ov < wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    if (it > 0 && tid == 0) knn_idx[((size_t)pair * S + s) * k + (it - 1)] = (wi < N) ? wi : min(it, N - 1);   // (NaN rows: in range all the same)
    if ((wi & 255) == tid) {               // owner: drop the element, rescan the registers
      const int mm = wi >> 8;
      bv = INFINITY; bm = 0;
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        if (m == mm) v[m] = INFINITY;
        if (v[m] < bv) { bv = v[m]; bm = m; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Top-(k+1) selection in O(1) barriers.  T = the (k+1)-th smallest of the 256 per-thread minima bounds the
// (k+1)-th smallest element overall, so every wanted element is among the c elements <= T (c ~ k+1 for random
// rows); those are gathered into LDS and ranked by counting under the order (distance, index).  Falls back to the
// round-based loop of k_knn_select in the (adversarial) case c > kCandMax.   grid (S, B), block 256.
// ---------------------------------------------------------------------------------------
constexpr int kCandMax = 1024;

template <int EPT>
__global__ void __launch_bounds__(256, EPT <= 20 ? 6 : EPT <= 32 ? 5 : 3)      // (a latency chain per row: 5-6 rows per SIMD in flight)
k_knn_select_fast(const float* __restrict__ dist_in, int* __restrict__ knn_idx, int N, int S, int k, const PairTab* __restrict__ ptab, int ld) {
  __shared__ float mins[256];
  __shared__ __attribute__((aligned(16))) float cand_v[kCandMax + 4];
  __shared__ __attribute__((aligned(16))) int cand_i[kCandMax + 4];
  __shared__ float T_sh;
  __shared__ int count;
  __shared__ float red_v[2][4];
  __shared__ int red_i[2][4];
  const int pair = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
  if (s >= pair_seeds(ptab, pair, S)) return;
  const float* dr = dist_in + ((size_t)pair * S + s) * ld;                // (strides of the largest pair; rows padded to ld floats)
  int* out = knn_idx + ((size_t)pair * S + s) * k;
  N = pair_rows(ptab, pair, N);
  // a row with NaN distances yields fewer than k + 1 candidates: every slot holds an in-range index before the ranks are
  // written (the barriers below order this store before theirs), so no consumer ever gathers through an uninitialised index
  if (tid < k) out[tid] = min(tid + 1, N - 1);
  float v[EPT];
#pragma unroll
  for (int m = 0; m < EPT; ++m) { const int j = tid + 256 * m; v[m] = (j < N) ? dr[j] : INFINITY; }
  float bv = INFINITY; int bm = 0;
#pragma unroll
  for (int m = 0; m < EPT; ++m) if (v[m] < bv) { bv = v[m]; bm = m; }
  if (tid == 0) count = 0;
  if (k + 1 <= 64) {
    // T from 64 GROUP minima (a group = 4 neighbouring threads = 4 * EPT elements): the (k + 1)-th smallest of 64 distinct
    // elements also bounds the (k + 1)-th smallest overall, at a 16th of the comparisons (every thread used to rank its own
    // minimum among all 256 - 65 536 comparisons per seed row, half of this kernel's instructions); the candidate count grows
    // from ~k + 4 to ~1.5 k, far below kCandMax.
    float gv = fminf(bv, __shfl_xor(bv, 1, 64));
    gv = fminf(gv, __shfl_xor(gv, 2, 64));
    if ((tid & 3) == 0) mins[tid >> 2] = gv;
    __syncthreads();
    if (tid < 64) {
      const float mine = mins[tid];
      int rank = 0;
      const float4* m4 = reinterpret_cast<const float4*>(mins);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float4 q = m4[u];
        rank += (q.x < mine || (q.x == mine && 4 * u + 0 < tid)) ? 1 : 0;
        rank += (q.y < mine || (q.y == mine && 4 * u + 1 < tid)) ? 1 : 0;
        rank += (q.z < mine || (q.z == mine && 4 * u + 2 < tid)) ? 1 : 0;
        rank += (q.w < mine || (q.w == mine && 4 * u + 3 < tid)) ? 1 : 0;
      }
      if (rank == k) T_sh = mine;          // exactly one of the 64 has this rank
    }
  } else {
    mins[tid] = bv;
    __syncthreads();
    int rank = 0;
    const float4* m4 = reinterpret_cast<const float4*>(mins);
#pragma unroll 8
    for (int u = 0; u < 64; ++u) {
      const float4 q = m4[u];
      rank += (q.x < bv || (q.x == bv && 4 * u + 0 < tid)) ? 1 : 0;
      rank += (q.y < bv || (q.y == bv && 4 * u + 1 < tid)) ? 1 : 0;
      rank += (q.z < bv || (q.z == bv && 4 * u + 2 < tid)) ? 1 : 0;
      rank += (q.w < bv || (q.w == bv && 4 * u + 3 < tid)) ? 1 : 0;
    }
    if (rank == k) T_sh = bv;            // k <= 64 < 256: exactly one thread has this rank
  }
  __syncthreads();
  const float T = T_sh;
  // candidates (elements <= T) into the LDS: a thread counts its own, a wave-wide prefix sum places them behind ONE LDS atomic
  // per wave (there used to be one atomic per candidate)
  {
    const int lane = tid & 63;
    int mine = 0;
#pragma unroll
    for (int m = 0; m < EPT; ++m) mine += (v[m] <= T) ? 1 : 0;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    int base = 0;
    if (lane == 63 && inc) base = atomicAdd(&count, inc);
    base = __shfl(base, 63, 64);
    int pos = base + inc - mine;
    if (mine) {
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        if (v[m] <= T) {
          if (pos < kCandMax) { cand_v[pos] = v[m]; cand_i[pos] = tid + 256 * m; }
          ++pos;
        }
      }
    }
  }
  __syncthreads();
  const int c = count;
  if (c <= kCandMax) {
    // rank of every candidate among the candidates under (distance, index), four candidates per LDS read (the tail of the last
    // group of four is padded with +inf / the largest index, which ranks behind everything)
    const int c4 = (c + 3) & ~3;
    if (tid < c4 - c) { cand_v[c + tid] = INFINITY; cand_i[c + tid] = 0x7fffffff; }
    __syncthreads();
    const float4* cv4 = reinterpret_cast<const float4*>(cand_v);
    const int4* ci4 = reinterpret_cast<const int4*>(cand_i);
    for (int p = tid; p < c; p += 256) {
      const float pv = cand_v[p]; const int pi = cand_i[p];
      int rank = 0;
      for (int q = 0; q < c4 / 4; ++q) {
        const float4 qv = cv4[q]; const int4 qi = ci4[q];
        rank += (qv.x < pv || (qv.x == pv && qi.x < pi)) ? 1 : 0;
        rank += (qv.y < pv || (qv.y == pv && qi.y < pi)) ? 1 : 0;
        rank += (qv.z < pv || (qv.z == pv && qi.z < pi)) ? 1 : 0;
        rank += (qv.w < pv || (qv.w == pv && qi.w < pi)) ? 1 : 0;
      }
      if (rank >= 1 && rank <= k) out[rank - 1] = pi;     // rank 0 (the row itself) is dropped, common.py:74
    }
    return;
  }
  // fallback: k+1 rounds of block-wide argmin over the cached per-thread minima
  const int lane = tid & 63, wave = tid >> 6;
  for (int it = 0; it <= k; ++it) {
    float rv = bv; int ri = (bv < INFINITY) ? tid + 256 * bm : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(rv, o, 64); const int oi = __shfl_xor(ri, o, 64);
      if (ov < rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
    }
    const int buf = it & 1;
    if (lane == 0) { red_v[buf][wave] = rv; red_i[buf][wave] = ri; }
    __syncthreads();
    float wv = red_v[buf][0]; int wi = red_i[buf][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = red_v[buf][w]; const int oi = red_i[buf][w];
      if (ov < wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    if (it > 0 && tid == 0) out[it - 1] = (wi < N) ? wi : min(it, N - 1);
    if ((wi & 255) == tid) {
      const int mm = wi >> 8;
      bv = INFINITY; bm = 0;
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        if (m == mm) v[m] = INFINITY;
        if (v[m] < bv) { bv = v[m]; bm = m; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// [r4] The same selection for rows of ANY length (N > 16 384: the reference has no limit, common.py:53-75): the distance row is not held
// in registers but streamed from global memory twice - once for the per-thread minima that give the threshold T (the (k + 1)-th
// smallest of the 64 group minima, exactly as above), once to collect the candidates <= T (one LDS atomic per candidate: ~1.5 k of
// them per row) - and the candidates are ranked under (distance, index) as above.  The output is a function of the row alone, so it
// equals k_knn_select_fast's wherever both apply (test_knn_stream_equals_register_form).  Adversarial rows (more than kCandMax
// candidates) take k + 1 streamed rounds: the next element in (distance, index) order behind the last one extracted.
// grid (S, B), block 256.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_knn_select_stream(const float* __restrict__ dist_in, int* __restrict__ knn_idx, int N, int S, int k, const PairTab* __restrict__ ptab, int ld) {
  __shared__ float mins[256];
  __shared__ __attribute__((aligned(16))) float cand_v[kCandMax + 4];
  __shared__ __attribute__((aligned(16))) int cand_i[kCandMax + 4];
  __shared__ float T_sh;
  __shared__ int count;
  __shared__ float red_v[2][4];
  __shared__ int red_i[2][4];
  const int pair = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
  if (s >= pair_seeds(ptab, pair, S)) return;
  const float* dr = dist_in + ((size_t)pair * S + s) * ld;
  int* out = knn_idx + ((size_t)pair * S + s) * k;
  N = pair_rows(ptab, pair, N);
  if (tid < k) out[tid] = min(tid + 1, N - 1);     // (NaN rows: every slot holds an in-range index, see k_knn_select_fast)
  float bv = INFINITY;
  for (int j = tid; j < N; j += 256) bv = fminf(bv, dr[j]);
  if (tid == 0) count = 0;
  if (k + 1 <= 64) {
    float gv = fminf(bv, __shfl_xor(bv, 1, 64));
    gv = fminf(gv, __shfl_xor(gv, 2, 64));
    if ((tid & 3) == 0) mins[tid >> 2] = gv;
    __syncthreads();
    if (tid < 64) {
      const float mine = mins[tid];
      int rank = 0;
      for (int u = 0; u < 64; ++u) { const float q = mins[u]; rank += (q < mine || (q == mine && u < tid)) ? 1 : 0; }
      if (rank == k) T_sh = mine;
    }
  } else {
    mins[tid] = bv;
    __syncthreads();
    int rank = 0;
    for (int u = 0; u < 256; ++u) { const float q = mins[u]; rank += (q < bv || (q == bv && u < tid)) ? 1 : 0; }
    if (rank == k) T_sh = bv;
  }
  __syncthreads();
  const float T = T_sh;
  for (int j = tid; j < N; j += 256) {
    const float v = dr[j];
    if (v <= T) {
      const int pos = atomicAdd(&count, 1);
      if (pos < kCandMax) { cand_v[pos] = v; cand_i[pos] = j; }
    }
  }
  __syncthreads();
  const int c = count;
  if (c <= kCandMax) {
    for (int p = tid; p < c; p += 256) {
      const float pv = cand_v[p]; const int pi = cand_i[p];
      int rank = 0;
      for (int q = 0; q < c; ++q) { const float qv = cand_v[q]; const int qi = cand_i[q]; rank += (qv < pv || (qv == pv && qi < pi)) ? 1 : 0; }
      if (rank >= 1 && rank <= k) out[rank - 1] = pi;     // rank 0 (the row itself) is dropped, common.py:74
    }
    return;
  }
  // fallback: k + 1 streamed rounds, each the smallest element behind (last_v, last_i) in (distance, index) order
  const int lane = tid & 63, wave = tid >> 6;
  float last_v = -INFINITY; int last_i = -1;
  for (int it = 0; it <= k; ++it) {
    float rv = INFINITY; int ri = 0x7fffffff;
    for (int j = tid; j < N; j += 256) {
      const float v = dr[j];
      const bool behind = v > last_v || (v == last_v && j > last_i);
      if (behind && (v < rv || (v == rv && j < ri))) { rv = v; ri = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(rv, o, 64); const int oi = __shfl_xor(ri, o, 64);
      if (ov < rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
    }
    const int buf = it & 1;
    if (lane == 0) { red_v[buf][wave] = rv; red_i[buf][wave] = ri; }
    __syncthreads();
    float wv = red_v[buf][0]; int wi = red_i[buf][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = red_v[buf][w]; const int oi = red_i[buf][w];
      if (ov < wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    if (it > 0 && tid == 0) out[it - 1] = (wi < N) ? wi : min(it, N - 1);
    last_v = wv; last_i = wi;
  }
}

// ---------------------------------------------------------------------------------------
// Per seed: M = clamp(1-(1-F F^T)/sigma^2,0) * clamp(1-(ds-dt)^2/sigma_d^2,0), diag 0; power iteration
// with every iterate stored (the reference's early exit is a GLOBAL allclose over all seeds of the
// pair, resolved by k_seed_kabsch).  One wave per seed, lane a = neighbour a.  k <= 64.
// grid (S, B), block 64
// snaps [B,S,iters,k], conv [B,S,iters] (1 if allclose(v_it, v_{it-1}))
// ---------------------------------------------------------------------------------------
constexpr int kKMax = 64;

// The k x k feature Gram matrix comes from the f16 MFMA on split-fp16 operands (fp32-equivalent, mfma_core.hpp): the wave gathers its
// neighbours' unit features as two 32-row fragments from the row-major features (no LDS staging) and forms G00, G01, G11
// (G10 = G01^T) with 24 MFMAs each; lane (h, b) then owns column b of each tile and turns the dot products into
// M entries together with the spatial term.
__global__ void __launch_bounds__(64, 3)     // [r4] <= 168 registers: three waves per SIMD for a kernel that is one latency chain per wave (180 registers, two waves: 195 us)
k_seed_power(const float* __restrict__ feat_n, const float* __restrict__ src, const float* __restrict__ tgt,
             const int* __restrict__ knn_idx, float* __restrict__ snaps, unsigned char* __restrict__ conv,
             double* __restrict__ hsum, int N, int S, int k, int iters, float inv_sigma2, float inv_sigmad2,
             const PairTab* __restrict__ ptab, const float* __restrict__ sigma_dev) {
  inv_sigma2 = sigma_inv2(inv_sigma2, sigma_dev);
  // LDS sized by k (dynamic): the k x k matrix with row stride k + 1 - at k = 40 9 KiB per seed instead of 19, i.e. 17 instead
  // of 8 resident seeds per CU for a kernel that is one latency chain per seed
  extern __shared__ __attribute__((aligned(16))) float seed_smem[];
  float* const P = seed_smem;                      // [64][8]: xyz of the neighbour in the two clouds
  float* const vec = P + kKMax * 8;                // [64]
  float* const Mx = vec + kKMax;                   // [k][k + 1]
  const int ld = k + 1;
  const int pair = blockIdx.y, s = blockIdx.x, a = threadIdx.x;
  const int lane = a, h = lane >> 5, i = lane & 31;
  if (s >= pair_seeds(ptab, pair, S)) return;                // (ragged batch; the per-seed buffers keep the stride S)
  const int* nb = knn_idx + ((size_t)pair * S + s) * k;
  const size_t row0 = pair_row0(ptab, pair, N);
  N = gmf::pair_rows(ptab, pair, N);
  const float* pair_rows = feat_n + row0 * 128;
  {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* Pz = reinterpret_cast<float4*>(P);
    Pz[2 * a] = z; Pz[2 * a + 1] = z;
  }
  if (a < k) {
    const int j = nb[a];
    const float* ps = src + (row0 + j) * 3;
    const float* pt = tgt + (row0 + j) * 3;
    P[a * 8 + 0] = ps[0]; P[a * 8 + 1] = ps[1]; P[a * 8 + 2] = ps[2];
    P[a * 8 + 4] = pt[0]; P[a * 8 + 5] = pt[1]; P[a * 8 + 6] = pt[2];
  }
  // the neighbours' unit features as two 32-row fragments, gathered from the ROW-MAJOR features: a row is 4 cache lines there;
  // in the P32 image (what the MFMA kernels stream) a row's 16-byte pieces lie 512 bytes apart - 32 lines per row, and this
  // kernel's 16 000 gathering waves were bound by that line traffic (4 GB per launch at 32 x 5000)
  auto row_frag = [&](float (&x)[64], const int row) {
    if (row >= 0 && row < N) {
      const float4* p = reinterpret_cast<const float4*>(pair_rows + (size_t)row * 128) + h;
#pragma unroll
      for (int g = 0; g < 16; ++g) {                 // fragment group g = features 32 (g >> 2) + 8 (g & 3) + 4 h .. + 3
        const float4 t = p[8 * (g >> 2) + 2 * (g & 3)];
        x[4 * g + 0] = t.x; x[4 * g + 1] = t.y; x[4 * g + 2] = t.z; x[4 * g + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 64; ++e) x[e] = 0.f;
    }
  };
  float fa[64], fb[64];
  row_frag(fa, (i < k) ? nb[i] : -1);
  if (k > 32) row_frag(fb, (32 + i < k) ? nb[32 + i] : -1);
  else {
#pragma unroll
    for (int e = 0; e < 64; ++e) fb[e] = 0.f;
  }
  __syncthreads();

  // tile (ta, tb): rows 32*ta + a', cols 32*tb + (lane & 31).  Branch-free per element (rows / columns beyond k read the zeroed
  // padding of P and are not stored), square roots by sqrt_cr (bit-identical to sqrtf on squared distances, 6 instructions
  // instead of ~20): the 48 elements of a seed used to be 48 basic blocks with two full sqrtf expansions each - 60 % of the
  // kernel's instructions.
  auto emit_tile = [&](const gmf::f32x16& g, int ta, int tb, bool mirror) {
    const int b = 32 * tb + i;
    const float bsx = P[b * 8], bsy = P[b * 8 + 1], bsz = P[b * 8 + 2];
    const float btx = P[b * 8 + 4], bty = P[b * 8 + 5], btz = P[b * 8 + 6];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // [r4] register groups whose rows all lie beyond k hold nothing (uniform: at k = 40 the tiles of rows 32.. keep one group of four)
      if ((r & 3) == 0 && 32 * ta + 8 * (r >> 2) >= k) break;
      const int ar = 32 * ta + 8 * (r >> 2) + 4 * h + (r & 3);
      const float mf = fmaxf(1.0f - (1.0f - g[r]) * inv_sigma2, 0.f);
      const float ax = P[ar * 8] - bsx, ay = P[ar * 8 + 1] - bsy, az = P[ar * 8 + 2] - bsz;
      const float bx = P[ar * 8 + 4] - btx, by = P[ar * 8 + 5] - bty, bz = P[ar * 8 + 6] - btz;
      const float d = gmf::sqrt_cr(ax * ax + ay * ay + az * az) - gmf::sqrt_cr(bx * bx + by * by + bz * bz);
      const float ms = fmaxf(1.0f - d * d * inv_sigmad2, 0.f);
      const float m = (ar == b) ? 0.f : mf * ms;
      if (ar < k && b < k) {
        Mx[ar * ld + b] = m;
        if (mirror) Mx[b * ld + ar] = m;
      }
    }
  };
  // Gram tiles on the f16 MFMA with split-fp16 operands (x = hi + lo, products hl + lh + hh, fp32 accumulate: fp32-equivalent,
  // mfma_core.hpp): 24 MFMAs of 32 cycles per tile instead of 64 of 64
  gmf::f16x8 ah[8], al[8], bh[8], bl[8];
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) { gmf::split8h(fa + 8 * s8, ah[s8], al[s8]); gmf::split8h(fb + 8 * s8, bh[s8], bl[s8]); }
  {
    gmf::f32x16 g = gmf::zero16();
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) gmf::mma3(g, ah[s8], al[s8], ah[s8], al[s8]);
    emit_tile(g, 0, 0, false);
  }
  if (k > 32) {
    // [r4] the off-diagonal block as G10 = F_b F_a^T (rows 32 .., columns 0 .. 31) instead of G01: its k - 32 valid rows are one or
    // two register groups on ALL 64 lanes, where G01's k - 32 valid columns were 16 registers on a quarter of the lanes - at
    // k = 40 a seed's 48 matrix entries per lane (~38 vector instructions each) become 24.  Same products in the same order (the
    // operands of each MFMA change places, the three partial products keep their sequence): every entry is bit-identical.
    gmf::f32x16 g = gmf::zero16();
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      g = gmf::mfma_h16(bh[s8], al[s8], g);
      g = gmf::mfma_h16(bl[s8], ah[s8], g);
      g = gmf::mfma_h16(bh[s8], ah[s8], g);
    }
    emit_tile(g, 1, 0, true);
    g = gmf::zero16();
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) gmf::mma3(g, bh[s8], bl[s8], bh[s8], bl[s8]);
    emit_tile(g, 1, 1, false);
  }
  vec[a] = 1.0f;
  __syncthreads();
  float last = 1.0f;
  float* sn = snaps + ((size_t)pair * S + s) * iters * k;
  unsigned char* cv = conv + ((size_t)pair * S + s) * iters;
  // v <- M v / (|M v| + 1e-6), `iters` times.  Lane a keeps ROW a of M in registers for all iterations (the matrix used to be
  // re-read from the LDS in every one of them: 2 LDS reads per multiply-add); the iterate is read back as broadcast float4s.
  // Same products in the same order (b ascending), so every iterate is bit-identical to the two-read form.
  auto power = [&](auto kk_tag) {
    constexpr int KK = decltype(kk_tag)::value;          // compile-time row length >= k (entries beyond k are zero)
    float mrow[KK];
#pragma unroll
    for (int b = 0; b < KK; ++b) mrow[b] = (a < k && b < k) ? Mx[a * ld + b] : 0.f;
    const float4* v4 = reinterpret_cast<const float4*>(vec);
    for (int it = 0; it < iters; ++it) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < KK / 4; ++q) {
        const float4 t = v4[q];
        v = fmaf(mrow[4 * q + 0], t.x, v); v = fmaf(mrow[4 * q + 1], t.y, v);
        v = fmaf(mrow[4 * q + 2], t.z, v); v = fmaf(mrow[4 * q + 3], t.w, v);
      }
      float n2 = (a < k) ? v * v : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 64);
      v = v / (sqrtf(n2) + 1e-6f);
      const bool close = (a >= k) || (fabsf(v - last) <= 1e-8f + 1e-5f * fabsf(last));
      const bool all_close = __all(close);
      __syncthreads();
      if (a < k) { vec[a] = v; sn[it * k + a] = v; }
      if (a == 0) cv[it] = all_close ? 1 : 0;
      last = v;
      __syncthreads();
    }
  };
  if (k <= 40) power(std::integral_constant<int, 40>{});
  else power(std::integral_constant<int, 64>{});
  if (!hsum) return;
  // Weights w = v / (sum v + 1e-6) (PointDSC.py:364-365) and the sums of the weighted Kabsch problem (common.py:10-50:
  // centroids and H) for the LAST iterate, wave-parallel over the k neighbours: hsum [B,S,15] = ca, cb, H.  k_seed_kabsch
  // (one lane per seed) then only runs the 3x3 SVD - unless every seed of the pair passed allclose earlier, in which case it
  // redoes the pair's sums from the iterate the reference stopped at.
  {
    float sv = 0.f;
    for (int r = 0; r < k; ++r) sv += vec[r];            // (the order of k_seed_kabsch's sum)
    const float inv = 1.0f / (sv + 1e-6f);
    float wf = (a < k) ? last * inv : 0.f;
    wf = (wf < 0.f) ? 0.f : wf;                          // weights[weights < 0] = 0 (common.py:24)
    const double w = wf;
    double pa[3], pb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { pa[c] = (a < k) ? P[a * 8 + c] : 0.0; pb[c] = (a < k) ? P[a * 8 + 4 + c] : 0.0; }
    const double sw = wave_sum(w);
    double ca[3], cb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { ca[c] = wave_sum(w * pa[c]) / (sw + 1e-6); cb[c] = wave_sum(w * pb[c]) / (sw + 1e-6); }
    double H[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) H[3 * r + c] = wave_sum(w * (pa[r] - ca[r]) * (pb[c] - cb[c]));
    if (a < 15) {
      double* o = hsum + ((size_t)pair * S + s) * 15;
      double val = 0.0;
#pragma unroll
      for (int e = 0; e < 15; ++e) if (a == e) val = (e < 3) ? ca[e] : (e < 6) ? cb[e - 3] : H[e - 6];
      o[a] = val;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Per seed: pick the iterate the reference would have stopped at, normalise to weights
// w = v/(sum v + 1e-6) (PointDSC.py:364-365) and solve the weighted Kabsch problem on the k neighbours.
// One thread per seed.  grid (ceil(S/64), B), block 64.  seed_T [B,S,16]
// ---------------------------------------------------------------------------------------
// The iteration at which the reference's power iteration stops (PointDSC.py:444): the first at which EVERY seed of EVERY pair
// of the batch passed allclose (the test is over the whole [bs * S, k] tensor), else the last.  One workgroup; only launched
// for batches (for one pair k_seed_kabsch finds it itself).
__global__ void __launch_bounds__(1024)
k_stop_iteration(const unsigned char* __restrict__ conv, int total_seeds, int iters, int* __restrict__ stop_out) {
  __shared__ unsigned long long red[16];
  // [r4] conv [total_seeds, iters] bytes is read as ONE flat array in coalesced 16-byte pieces (it was read seed by seed, one byte
  // per load: 27 us for 160 KB, 3 % of the pose head); byte p belongs to iteration p % iters.
  // bit `it` of `bad`: some seed did NOT pass at iteration `it`
  unsigned long long bad = 0;
  const size_t total = (size_t)total_seeds * iters;
  const size_t n16 = total >> 4;                   // (the workspace slot is 256-byte aligned)
  const uint4* c16 = reinterpret_cast<const uint4*>(conv);
  for (size_t c = threadIdx.x; c < n16; c += blockDim.x) {
    const uint4 v = c16[c];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    int it = (int)((c << 4) % (size_t)iters);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (((w[q] >> (8 * e)) & 0xffu) == 0) bad |= 1ull << it;
        it = (it + 1 == iters) ? 0 : it + 1;
      }
  }
  for (size_t p = (n16 << 4) + threadIdx.x; p < total; p += blockDim.x)
    if (conv[p] == 0) bad |= 1ull << (int)(p % (size_t)iters);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) bad |= red[w];
    int first = iters - 1;
    for (int it = 0; it < iters - 1; ++it)
      if (!((bad >> it) & 1ull)) { first = it; break; }
    *stop_out = first;
  }
}

__global__ void __launch_bounds__(64)
k_seed_kabsch(const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ knn_idx,
              const float* __restrict__ snaps, const unsigned char* __restrict__ conv, float* __restrict__ seed_T,
              int N, int S, int k, int iters, const double* __restrict__ hsum, const int* __restrict__ stop_batch,
              const PairTab* __restrict__ ptab) {
  __shared__ int stop_it;
  const int pair = blockIdx.y;
  const int S_p = pair_seeds(ptab, pair, S);                // (ragged batch; the per-seed buffers keep the stride S)
  if ((int)blockIdx.x * 64 >= S_p) return;
  // global early exit: first iteration at which EVERY seed passed allclose (PointDSC.py:444) - of the whole batch
  // (stop_batch, from k_stop_iteration) or, for a single pair, found here
  if (threadIdx.x == 0) stop_it = iters - 1;
  __syncthreads();
  if (stop_batch) {
    if (threadIdx.x == 0) stop_it = *stop_batch;
  } else {
    const unsigned char* cv = conv + (size_t)pair * S * iters;
    int first = iters - 1;
    for (int it = 0; it < iters - 1; ++it) {
      bool ok = true;
      for (int s2 = threadIdx.x; s2 < S_p; s2 += 64) ok = ok && (cv[(size_t)s2 * iters + it] != 0);
      if (__all(ok)) { first = it; break; }
    }
    if (threadIdx.x == 0) stop_it = first;
  }
  __syncthreads();
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= S_p) return;
  if (hsum && stop_it == iters - 1) {                // k_seed_power already summed the last iterate: only the SVD is left
    const double* hs = hsum + ((size_t)pair * S + s) * 15;
    double ca[3], cb[3], H[9], R[9];
#pragma unroll
    for (int e = 0; e < 3; ++e) { ca[e] = hs[e]; cb[e] = hs[3 + e]; }
#pragma unroll
    for (int e = 0; e < 9; ++e) H[e] = hs[6 + e];
    kabsch_rotation_from_H(H, R);
    write_T(seed_T + ((size_t)pair * S + s) * 16, R, ca, cb);
    return;
  }
  const int* nb = knn_idx + ((size_t)pair * S + s) * k;
  const float* v = snaps + (((size_t)pair * S + s) * iters + stop_it) * k;
  float sv = 0.f;
  for (int r = 0; r < k; ++r) sv += v[r];
  const float inv = 1.0f / (sv + 1e-6f);
  const float* ps = src + pair_row0(ptab, pair, N) * 3;
  const float* pt = tgt + pair_row0(ptab, pair, N) * 3;
  double sw = 0, ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int r = 0; r < k; ++r) {
    float w = v[r] * inv;
    w = (w < 0.f) ? 0.f : w;                      // weights[weights < 0] = 0 (common.py:24)
    const int j = nb[r];
    sw += w;
    for (int c = 0; c < 3; ++c) { ca[c] += (double)w * ps[3 * j + c]; cb[c] += (double)w * pt[3 * j + c]; }
  }
  for (int c = 0; c < 3; ++c) { ca[c] /= (sw + 1e-6); cb[c] /= (sw + 1e-6); }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < k; ++r) {
    float w = v[r] * inv;
    w = (w < 0.f) ? 0.f : w;
    const int j = nb[r];
    double am[3], bm[3];
    for (int c = 0; c < 3; ++c) { am[c] = ps[3 * j + c] - ca[c]; bm[c] = pt[3 * j + c] - cb[c]; }
    for (int r2 = 0; r2 < 3; ++r2) for (int c = 0; c < 3; ++c) H[3 * r2 + c] += w * am[r2] * bm[c];
  }
  double R[9];
  kabsch_rotation_from_H(H, R);
  write_T(seed_T + ((size_t)pair * S + s) * 16, R, ca, cb);
}

// ---------------------------------------------------------------------------------------
// Inlier count of each hypothesis over all N correspondences.  grid (ceil(S / kHypPerWG), B), block 256.
// A workgroup scores kHypPerWG hypotheses per pass over the points (one hypothesis per workgroup re-read the pair's
// 24 N bytes from L2 for each of its S hypotheses: 1.9 GB per launch at B = 32, N = 5000, S = 500 - L2-bandwidth bound).
// tau2t: smallest float t with sqrtf(t) >= tau, so that d2 < t  <=>  sqrtf(d2) < tau  (the reference's test, PointDSC.py:415).
// ---------------------------------------------------------------------------------------
constexpr int kHypPerWG = 8;

__global__ void __launch_bounds__(256)
k_score_hyp(const float* __restrict__ src, const float* __restrict__ tgt, const float* __restrict__ seed_T,
            int* __restrict__ counts, int N, int S, float tau2t, const PairTab* __restrict__ ptab) {
  __shared__ int red[4][kHypPerWG];
  const int pair = blockIdx.y, s0 = blockIdx.x * kHypPerWG;
  const int S_p = pair_seeds(ptab, pair, S);                // (ragged batch; the per-seed buffers keep the stride S)
  if (s0 >= S_p) return;
  // two hypotheses per packed-fp32 instruction (v_pk_mul / v_pk_fma / v_pk_add_f32: the same operations with the same roundings
  // as the scalar form, half the instructions of a kernel that is nothing but vector arithmetic)
  nms_f2 Tm[kHypPerWG / 2][12];
#pragma unroll
  for (int q = 0; q < kHypPerWG / 2; ++q) {
    const float* Ta = seed_T + ((size_t)pair * S + min(s0 + 2 * q, S_p - 1)) * 16;
    const float* Tb = seed_T + ((size_t)pair * S + min(s0 + 2 * q + 1, S_p - 1)) * 16;
#pragma unroll
    for (int e = 0; e < 12; ++e) Tm[q][e] = nms_f2{Ta[e], Tb[e]};
  }
  const float* ps = src + pair_row0(ptab, pair, N) * 3;
  const float* pt = tgt + pair_row0(ptab, pair, N) * 3;
  N = pair_rows(ptab, pair, N);
  int cnt[kHypPerWG];
#pragma unroll
  for (int q = 0; q < kHypPerWG; ++q) cnt[q] = 0;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float xs = ps[3 * j], ys = ps[3 * j + 1], zs = ps[3 * j + 2];
    const float us = pt[3 * j], vs = pt[3 * j + 1], ws = pt[3 * j + 2];
    const nms_f2 x = {xs, xs}, y = {ys, ys}, z = {zs, zs}, u = {us, us}, v = {vs, vs}, w = {ws, ws};
#pragma unroll
    for (int q = 0; q < kHypPerWG / 2; ++q) {
      const nms_f2 dx = (Tm[q][0] * x + Tm[q][1] * y + Tm[q][2] * z) + Tm[q][3] - u;
      const nms_f2 dy = (Tm[q][4] * x + Tm[q][5] * y + Tm[q][6] * z) + Tm[q][7] - v;
      const nms_f2 dz = (Tm[q][8] * x + Tm[q][9] * y + Tm[q][10] * z) + Tm[q][11] - w;
      const nms_f2 d2 = dx * dx + dy * dy + dz * dz;
      cnt[2 * q] += (d2[0] < tau2t) ? 1 : 0;
      cnt[2 * q + 1] += (d2[1] < tau2t) ? 1 : 0;
    }
  }
#pragma unroll
  for (int q = 0; q < kHypPerWG; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt[q] += __shfl_xor(cnt[q], o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = cnt[q];
  }
  __syncthreads();
  if (threadIdx.x < kHypPerWG && s0 + threadIdx.x < S_p)
    counts[(size_t)pair * S + s0 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------
// Weighted Kabsch over a (masked) point set by a whole workgroup; T updated in shared memory.
// ---------------------------------------------------------------------------------------
GMF_DEVINL void block_kabsch(const float* ps, const float* pt, int N, const float* Tin, float thr, bool use_mask,
                             float* Tout /*shared, 16*/, double* sh) {
  const float r00 = Tin[0], r01 = Tin[1], r02 = Tin[2], t0 = Tin[3], r10 = Tin[4], r11 = Tin[5], r12 = Tin[6],
              t1 = Tin[7], r20 = Tin[8], r21 = Tin[9], r22 = Tin[10], t2 = Tin[11];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const float x = ps[3 * j], y = ps[3 * j + 1], z = ps[3 * j + 2];
    const float dx = (r00 * x + r01 * y + r02 * z) + t0 - pt[3 * j];
    const float dy = (r10 * x + r11 * y + r12 * z) + t1 - pt[3 * j + 1];
    const float dz = (r20 * x + r21 * y + r22 * z) + t2 - pt[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!use_mask || d < thr) {
      const float q = d / thr;
      const double w = 1.0f / (1.0f + q * q);
      acc[0] += w;
      acc[1] += w * x; acc[2] += w * y; acc[3] += w * z;
      acc[4] += w * pt[3 * j]; acc[5] += w * pt[3 * j + 1]; acc[6] += w * pt[3 * j + 2];
    }
  }
  block_sum<7>(acc, sh);
  const double den = acc[0] + 1e-6;
  const double ca[3] = {acc[1] / den, acc[2] / den, acc[3] / den};
  const double cb[3] = {acc[4] / den, acc[5] / den, acc[6] / den};
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const float x = ps[3 * j], y = ps[3 * j + 1], z = ps[3 * j + 2];
    const float dx = (r00 * x + r01 * y + r02 * z) + t0 - pt[3 * j];
    const float dy = (r10 * x + r11 * y + r12 * z) + t1 - pt[3 * j + 1];
    const float dz = (r20 * x + r21 * y + r22 * z) + t2 - pt[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!use_mask || d < thr) {
      const float q = d / thr;
      const double w = 1.0f / (1.0f + q * q);
      const double am[3] = {x - ca[0], y - ca[1], z - ca[2]};
      const double bm[3] = {pt[3 * j] - cb[0], pt[3 * j + 1] - cb[1], pt[3 * j + 2] - cb[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[3 * r + c] += w * am[r] * bm[c];
    }
  }
  block_sum<9>(H, sh);
  __syncthreads();
  if (threadIdx.x == 0) {
    double R[9];
    kabsch_rotation_from_H(H, R);
    write_T(Tout, R, ca, cb);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// post_refinement (PointDSC.py:493-528) by a whole workgroup: up to `iters` IRLS steps from the transform in Tcur (shared),
// with the reference's exit on an unchanged inlier count.  ONE pass over the points and ONE block reduction per step: the
// inlier count and the weighted sums W, sum w a, sum w b, sum w a b^T (fp64) come from the same sweep, and
//   H = sum w (a - ca)(b - cb)^T = Sab - ca Sb^T - Sa cb^T + W ca cb^T,   ca = Sa / (W + 1e-6), cb = Sb / (W + 1e-6)
// (exact algebra of common.py:25-37; the step used to take a count pass, a centroid pass and an H pass, each with its own
// reduction).  sh: >= 17 * 17 doubles.
// ---------------------------------------------------------------------------------------
GMF_DEVINL void irls_refine(const float* ps, const float* pt, int N, float* Tcur /*shared, 16*/, float thr, int iters,
                            double* sh) {
  int prev = 0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    float Tl[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Tl[e] = Tcur[e];
    double acc[17];
#pragma unroll
    for (int e = 0; e < 17; ++e) acc[e] = 0.0;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const float x = ps[3 * j], y = ps[3 * j + 1], z = ps[3 * j + 2];
      const float u = pt[3 * j], v = pt[3 * j + 1], q = pt[3 * j + 2];
      const float dx = (Tl[0] * x + Tl[1] * y + Tl[2] * z) + Tl[3] - u;
      const float dy = (Tl[4] * x + Tl[5] * y + Tl[6] * z) + Tl[7] - v;
      const float dz = (Tl[8] * x + Tl[9] * y + Tl[10] * z) + Tl[11] - q;
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      if (d < thr) {
        const float r = d / thr;
        const double w = 1.0f / (1.0f + r * r);
        const double wx = w * x, wy = w * y, wz = w * z;
        acc[0] += 1.0; acc[1] += w;
        acc[2] += wx; acc[3] += wy; acc[4] += wz;
        acc[5] += w * u; acc[6] += w * v; acc[7] += w * q;
        acc[8] += wx * u; acc[9] += wx * v; acc[10] += wx * q;
        acc[11] += wy * u; acc[12] += wy * v; acc[13] += wy * q;
        acc[14] += wz * u; acc[15] += wz * v; acc[16] += wz * q;
      }
    }
    block_sum_tree<17>(acc, sh);
    const int n_inl = (int)acc[0];
    if (abs(n_inl - prev) < 1) break;
    prev = n_inl;
    if (threadIdx.x == 0) {
      const double den = acc[1] + 1e-6;
      const double ca[3] = {acc[2] / den, acc[3] / den, acc[4] / den};
      const double cb[3] = {acc[5] / den, acc[6] / den, acc[7] / den};
      double H[9], R[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          H[3 * r + c] = acc[8 + 3 * r + c] - ca[r] * acc[5 + c] - acc[2 + r] * cb[c] + acc[1] * ca[r] * cb[c];
      kabsch_rotation_from_H(H, R);
      write_T(Tcur, R, ca, cb);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// Per pair: best hypothesis (first argmax of the inlier count), its inlier labels, and the
// reference's <=20-step IRLS refinement with early exit on an unchanged inlier count.
// grid (B), block 1024.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_finalize_pose(const float* __restrict__ src, const float* __restrict__ tgt, const float* __restrict__ seed_T,
                const int* __restrict__ counts, float* __restrict__ fitness, float* __restrict__ final_T,
                float* __restrict__ labels, int* __restrict__ best_out, int N, int S, float tau, float refine_thr,
                int refine_iters, const PairTab* __restrict__ ptab) {
  __shared__ double sh[17 * 17];
  __shared__ float Tcur[16];
  __shared__ int s_best;
  __shared__ int red_c[16], red_i[16];
  const int pair = blockIdx.x;
  const size_t row0 = pair_row0(ptab, pair, N);              // (ragged batch: labels are packed [sum n]; per-seed buffers keep the stride S)
  const int S_p = pair_seeds(ptab, pair, S);
  N = pair_rows(ptab, pair, N);
  const float* ps = src + row0 * 3;
  const float* pt = tgt + row0 * 3;
  // argmax (first maximal index)
  int bc = -1, bi = 0x7fffffff;
  for (int s = threadIdx.x; s < S_p; s += blockDim.x) {
    const int c = counts[(size_t)pair * S + s];
    fitness[(size_t)pair * S + s] = (float)c / (float)N;
    if (c > bc) { bc = c; bi = s; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int oc = __shfl_xor(bc, o, 64), oi = __shfl_xor(bi, o, 64);
    if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { red_c[threadIdx.x >> 6] = bc; red_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = red_c[0], ix = red_i[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (red_c[w] > c || (red_c[w] == c && red_i[w] < ix)) { c = red_c[w]; ix = red_i[w]; }
    s_best = ix;
    best_out[pair] = ix;
  }
  __syncthreads();
  if (threadIdx.x < 16) Tcur[threadIdx.x] = seed_T[((size_t)pair * S + s_best) * 16 + threadIdx.x];
  __syncthreads();
  // labels under the un-refined best hypothesis (PointDSC.py:423-425)
  {
    const float r00 = Tcur[0], r01 = Tcur[1], r02 = Tcur[2], t0 = Tcur[3], r10 = Tcur[4], r11 = Tcur[5], r12 = Tcur[6],
                t1 = Tcur[7], r20 = Tcur[8], r21 = Tcur[9], r22 = Tcur[10], t2 = Tcur[11];
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const float x = ps[3 * j], y = ps[3 * j + 1], z = ps[3 * j + 2];
      const float dx = (r00 * x + r01 * y + r02 * z) + t0 - pt[3 * j];
      const float dy = (r10 * x + r11 * y + r12 * z) + t1 - pt[3 * j + 1];
      const float dz = (r20 * x + r21 * y + r22 * z) + t2 - pt[3 * j + 2];
      labels[row0 + j] = (sqrtf(dx * dx + dy * dy + dz * dz) < tau) ? 1.f : 0.f;
    }
  }
  irls_refine(ps, pt, N, Tcur, refine_thr, refine_iters, sh);
  if (threadIdx.x < 16) final_T[(size_t)pair * 16 + threadIdx.x] = Tcur[threadIdx.x];
}

// ---------------------------------------------------------------------------------------
// Stand-alone post_refinement (PointDSC.py:493-528) for B pairs.  grid (B), block 1024.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_post_refine(const float* __restrict__ T_in, const float* __restrict__ src, const float* __restrict__ tgt,
              float* __restrict__ T_out, int N, float thr, int iters) {
  __shared__ double sh[17 * 17];
  __shared__ float Tcur[16];
  const int pair = blockIdx.x;
  const float* ps = src + (size_t)pair * N * 3;
  const float* pt = tgt + (size_t)pair * N * 3;
  if (threadIdx.x < 16) Tcur[threadIdx.x] = T_in[(size_t)pair * 16 + threadIdx.x];
  irls_refine(ps, pt, N, Tcur, thr, iters, sh);
  if (threadIdx.x < 16) T_out[(size_t)pair * 16 + threadIdx.x] = Tcur[threadIdx.x];
}

// ---------------------------------------------------------------------------------------
// Public batched op: rigid_transform_3d(A, B, weights, weight_threshold) (common.py:10-50).
// One wave per problem; A,B [n,k,3], w [n,k] or null.  grid (n), block 64.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_rigid_transform(const float* __restrict__ A, const float* __restrict__ Bp, const float* __restrict__ w,
                  float* __restrict__ T, int k, float weight_threshold) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const float* a = A + (size_t)n * k * 3;
  const float* b = Bp + (size_t)n * k * 3;
  const float* ww = w ? w + (size_t)n * k : nullptr;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = lane; j < k; j += 64) {
    float wj = ww ? ww[j] : 1.f;
    wj = (wj < weight_threshold) ? 0.f : wj;
    acc[0] += wj;
    for (int c = 0; c < 3; ++c) { acc[1 + c] += (double)wj * a[3 * j + c]; acc[4 + c] += (double)wj * b[3 * j + c]; }
  }
#pragma unroll
  for (int e = 0; e < 7; ++e) acc[e] = wave_sum(acc[e]);
  const double den = acc[0] + 1e-6;
  const double ca[3] = {acc[1] / den, acc[2] / den, acc[3] / den};
  const double cb[3] = {acc[4] / den, acc[5] / den, acc[6] / den};
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = lane; j < k; j += 64) {
    float wj = ww ? ww[j] : 1.f;
    wj = (wj < weight_threshold) ? 0.f : wj;
    double am[3], bm[3];
    for (int c = 0; c < 3; ++c) { am[c] = a[3 * j + c] - ca[c]; bm[c] = b[3 * j + c] - cb[c]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[3 * r + c] += wj * am[r] * bm[c];
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) H[e] = wave_sum(H[e]);
  if (lane == 0) {
    double R[9];
    kabsch_rotation_from_H(H, R);
    write_T(T + (size_t)n * 16, R, ca, cb);
  }
}

// ---------------------------------------------------------------------------------------
// DGR weighted_procrustes (core/registration.py:91-113), batched over pairs with ragged N.
//   w~ = w/(sum|w| + eps); mu = sum w~ p;  Sxy = sum (y-muy)(w~ (x-mux))^T;  Sxy = U D V^T (fp64);
//   R = U diag(1,1,det(U)det(V)) V^T;  t = muy - R mux.
// grid (B), block 1024.  offsets [B+1] row offsets into X,Y,w.
// ONE pass over the points: the raw moments sum|w|, sum w, sum w x, sum w y, sum w y x^T in fp64 (17 sums), from which
//   Sxy = sum w~ y x^T - (2 - sum w~) muy mux^T
// exactly as the centred sum expands; in fp64 the cancellation costs (|mu| / spread)^2 of 2^-53 - 1e-12 of the entries for a KITTI
// scene 100 m from the origin - where the reference's own fp32 sums carry 1e-7.  (Two passes, as the formula is written, are two
// dependent sweeps of a latency-bound kernel: 9 us more per call at 32 x 8000.)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_weighted_procrustes(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ w,
                      const int* __restrict__ offsets, float eps, float* __restrict__ Rout, float* __restrict__ tout) {
  __shared__ double sh[17 * 17];
  const int pair = blockIdx.x;
  const int o0 = offsets[pair], n = offsets[pair + 1] - o0;
  const float* x = X + (size_t)o0 * 3;
  const float* y = Y + (size_t)o0 * 3;
  const float* ww = w + o0;
  double acc[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) acc[k] = 0.0;
#pragma unroll 2
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double wj = ww[j];
    const double xj[3] = {x[3 * j], x[3 * j + 1], x[3 * j + 2]};
    const double yj[3] = {y[3 * j], y[3 * j + 1], y[3 * j + 2]};
    acc[0] += fabs(wj);
    acc[1] += wj;
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc[2 + c] += wj * xj[c]; acc[5 + c] += wj * yj[c]; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double wy = wj * yj[r];
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[8 + 3 * r + c] += wy * xj[c];
    }
  }
  block_sum_tree<17>(acc, sh);
  if (threadIdx.x == 0) {
    const double inv = 1.0 / ((double)(float)acc[0] + (double)eps);
    const double mx[3] = {acc[2] * inv, acc[3] * inv, acc[4] * inv};
    const double my[3] = {acc[5] * inv, acc[6] * inv, acc[7] * inv};
    const double k2 = 2.0 - acc[1] * inv;
    double Sm[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Sm[3 * r + c] = acc[8 + 3 * r + c] * inv - k2 * my[r] * mx[c];
    // kabsch_rotation_from_H(H) returns V D U^T for H = U S V^T; the DGR rotation is its transpose.
    double Rt[9];
    kabsch_rotation_from_H(Sm, Rt);
    float Rf[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rf[3 * r + c] = (float)Rt[3 * c + r];
    for (int e = 0; e < 9; ++e) Rout[(size_t)pair * 9 + e] = Rf[e];
    for (int r = 0; r < 3; ++r)
      tout[(size_t)pair * 3 + r] = (float)my[r] - (Rf[3 * r] * (float)mx[0] + Rf[3 * r + 1] * (float)mx[1] + Rf[3 * r + 2] * (float)mx[2]);
  }
}

// =========================================================================================
// k_global_registration: DGR robust pose refinement, one workgroup per pair.
//   GlobalRegistration (core/registration.py:135-194): weighted-Procrustes initialisation (:91-113), then up to max_iter
//   Adam steps (lr 0.1 * 0.999^i, torch defaults) on a 6-D rotation parameter (ortho2rotation :16-63) and a translation
//   under HighDimSmoothL1Loss (core/loss.py:42-61), with the reference's stopping rule (:173-186).
//   The reference spends 17-55 ms per step in autograd and optimizer dispatch for ~100 flops per point; here the whole
//   optimisation is ONE persistent kernel: per step every thread evaluates its points (fp32, as the reference), the
//   13 sums (loss, dL/dR, dL/dt) are reduced in fp64 across the workgroup, and every thread then repeats the identical
//   scalar tail (backward through the Gram-Schmidt map in fp64, Adam in fp32 as torch's single-tensor path, stopping
//   rule on the fp32-rounded loss) - no broadcast, no host round trip, no launch per step.
//   w == nullptr is the unweighted form (argmin_se3_squared_dist :66-88 initialisation, loss.mean()).
//   stats[pair] = {iterations, loss, break_count} (the reference's opt_result).
// =========================================================================================
GMF_DEVINL void cross3f(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// ortho2rotation in fp32, as the reference evaluates it.  aux = x[3], y[3], |a|, |v|, x.x, f
GMF_DEVINL void rot_from_6d(const float* a, float* R /* row-major */, float* aux) {
  const float ax = a[0], ay = a[1], az = a[2], bx = a[3], by = a[4], bz = a[5];
  const float na = fmaxf(sqrtf((ax * ax + ay * ay) + az * az), 1e-8f);
  const float x[3] = {ax / na, ay / na, az / na};
  const float n2 = fmaxf((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2], 1e-8f);
  const float f = ((x[0] * bx + x[1] * by) + x[2] * bz) / n2;
  const float v[3] = {bx - f * x[0], by - f * x[1], bz - f * x[2]};
  const float nv = fmaxf(sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]), 1e-8f);
  const float y[3] = {v[0] / nv, v[1] / nv, v[2] / nv};
  float z[3];
  cross3f(x, y, z);
#pragma unroll
  for (int r = 0; r < 3; ++r) { R[3 * r] = x[r]; R[3 * r + 1] = y[r]; R[3 * r + 2] = z[r]; }
#pragma unroll
  for (int r = 0; r < 3; ++r) { aux[r] = x[r]; aux[3 + r] = y[r]; }
  aux[6] = na; aux[7] = nv; aux[8] = n2; aux[9] = f;
}

__global__ void __launch_bounds__(1024)
k_global_registration(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ w,
                      const int* __restrict__ offsets, float eps, float qsize, int max_iter, int max_break, double ratio,
                      float* __restrict__ Rout, float* __restrict__ tout, float* __restrict__ stats) {
  __shared__ double sh[13 * 17];
  __shared__ float init_rt[12];
  const int pair = blockIdx.x;
  const int o0 = offsets[pair], n = offsets[pair + 1] - o0;
  const float* x = X + (size_t)o0 * 3;
  const float* y = Y + (size_t)o0 * 3;
  const float* ww = w ? w + o0 : nullptr;

  // ---- initialisation: (weighted) Procrustes, as k_weighted_procrustes ----
  double w1d;
  {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const double wj = ww ? (double)ww[j] : 1.0;
      acc[0] += fabs(wj);
      acc[7] += wj;
      for (int c = 0; c < 3; ++c) { acc[1 + c] += wj * x[3 * j + c]; acc[4 + c] += wj * y[3 * j + c]; }
    }
    block_sum<8>(acc, sh);
    w1d = ww ? (double)(float)acc[7] : (double)n;
    const double inv = 1.0 / ((double)(float)acc[0] + (ww ? (double)eps : 0.0));
    const double mx[3] = {acc[1] * inv, acc[2] * inv, acc[3] * inv};
    const double my[3] = {acc[4] * inv, acc[5] * inv, acc[6] * inv};
    double Sm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const double wn = (ww ? (double)ww[j] : 1.0) * inv;
      double xm[3], ym[3];
      for (int c = 0; c < 3; ++c) { xm[c] = x[3 * j + c] - mx[c]; ym[c] = y[3 * j + c] - my[c]; }
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Sm[3 * r + c] += ym[r] * (wn * xm[c]);
    }
    __syncthreads();
    block_sum<9>(Sm, sh);
    if (threadIdx.x == 0) {
      double Rt[9];
      kabsch_rotation_from_H(Sm, Rt);          // V D U^T; the DGR rotation is its transpose
      float Rf[9];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rf[3 * r + c] = (float)Rt[3 * c + r];
      for (int e = 0; e < 9; ++e) init_rt[e] = Rf[e];
      for (int r = 0; r < 3; ++r)
        init_rt[9 + r] = (float)my[r] - (Rf[3 * r] * (float)mx[0] + Rf[3 * r + 1] * (float)mx[1] + Rf[3 * r + 2] * (float)mx[2]);
    }
    __syncthreads();
  }
  // parameters: rot6d = first two COLUMNS of R (registration.py:122-124), trans = t.  Every thread keeps a copy.
  float a[6] = {init_rt[0], init_rt[3], init_rt[6], init_rt[1], init_rt[4], init_rt[7]};
  float tr[3] = {init_rt[9], init_rt[10], init_rt[11]};
  float m1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, m2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

  // The first kGrPts strided points of every thread stay in registers for the whole optimisation (N <= 16 x blockDim:
  // all of them); only the remainder is re-read from L2 each step.
  constexpr int kGrPts = 16;
  float rx[kGrPts][3], ry[kGrPts][3], rw[kGrPts];
#pragma unroll
  for (int q = 0; q < kGrPts; ++q) {
    const int j = threadIdx.x + q * blockDim.x;
    const bool in = j < n;
#pragma unroll
    for (int c = 0; c < 3; ++c) { rx[q][c] = in ? x[3 * j + c] : 0.f; ry[q][c] = in ? y[3 * j + c] : 0.f; }
    rw[q] = in ? (ww ? ww[j] : 1.0f) : 0.f;      // weight 0: a padded slot adds nothing to any sum
  }
  float Rf[9], aux[10];
  // loss sum and (optionally) the gradient sums for the current parameters
  auto evaluate = [&](bool grad, double (&red)[13]) {
    rot_from_6d(a, Rf, aux);
#pragma unroll
    for (int e = 0; e < 13; ++e) red[e] = 0.0;
    auto point = [&](const float px, const float py, const float pz, const float qx, const float qy, const float qz, const float wj) {
      const float yv[3] = {qx, qy, qz};
      float d[3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        d[r] = (((px * Rf[3 * r] + py * Rf[3 * r + 1]) + pz * Rf[3 * r + 2]) + tr[r] - yv[r]) / qsize;
      const float sq = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
      const bool small = sq < 1.0f;
      const float rt = sqrtf(sq + eps);
      const float li = small ? 0.5f * sq : 0.5f * (rt - 0.5f);
      red[0] += (double)(li * wj);
      if (grad) {
        // dL/dp = w * dl/dsq * 2 d / q   (the common 1/w1 is applied after the reduction)
        const float dls = small ? 0.5f : 0.25f / rt;
        const float cf = wj * dls * 2.0f / qsize;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double g = (double)(cf * d[r]);
          red[1 + 3 * r] += g * px; red[2 + 3 * r] += g * py; red[3 + 3 * r] += g * pz;
          red[10 + r] += g;
        }
      }
    };
#pragma unroll
    for (int q = 0; q < kGrPts; ++q) point(rx[q][0], rx[q][1], rx[q][2], ry[q][0], ry[q][1], ry[q][2], rw[q]);
    for (int j = threadIdx.x + kGrPts * blockDim.x; j < n; j += blockDim.x)
      point(x[3 * j], x[3 * j + 1], x[3 * j + 2], y[3 * j], y[3 * j + 1], y[3 * j + 2], ww ? ww[j] : 1.0f);
    __syncthreads();
    block_sum_tree<13>(red, sh);
  };

  double red[13];
  evaluate(false, red);
  float loss_prev = (float)(red[0] / w1d), lossf = loss_prev;
  int brk = 0, it = 0;
  double lr = 0.1, b1p = 1.0, b2p = 1.0;
  bool broke = false;
  for (it = 0; it < max_iter; ++it) {
    evaluate(true, red);
    lossf = (float)(red[0] / w1d);
    if (lossf < 1e-7f) { broke = true; break; }
    // ---- backward through new_points = points @ R^T + trans and R = ortho2rotation(rot6d), fp32 like autograd ----
    float G[9];   // dL/dR, row-major
#pragma unroll
    for (int e = 0; e < 9; ++e) G[e] = (float)(red[1 + e] / w1d);
    const float* xv = aux;
    const float* yv = aux + 3;
    const float na = aux[6], nv = aux[7], n2 = aux[8], f = aux[9];
    float gx[3] = {G[0], G[3], G[6]}, gy[3] = {G[1], G[4], G[7]}, gz[3] = {G[2], G[5], G[8]};
    float c1[3], c2[3];
    cross3f(yv, gz, c1);         // z = x cross y:  gx += y cross gz,  gy += gz cross x
    cross3f(gz, xv, c2);
#pragma unroll
    for (int r = 0; r < 3; ++r) { gx[r] += c1[r]; gy[r] += c2[r]; }
    const float ygy = (yv[0] * gy[0] + yv[1] * gy[1]) + yv[2] * gy[2];
    float gv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) gv[r] = (gy[r] - yv[r] * ygy) / nv;           // y = v / |v|
    const float b[3] = {a[3], a[4], a[5]};
    const float xgv = (xv[0] * gv[0] + xv[1] * gv[1]) + xv[2] * gv[2];
    const float xb = f * n2;
    float g9[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      g9[3 + r] = gv[r] - xv[r] * xgv / n2;                                    // v = b - f x, f = (x.b) / n2
      gx[r] += -(b[r] * xgv + xb * gv[r]) / n2 + 2.0f * xv[r] * xb * xgv / (n2 * n2);
    }
    const float xgx = (xv[0] * gx[0] + xv[1] * gx[1]) + xv[2] * gx[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      g9[r] = (gx[r] - xv[r] * xgx) / na;                                      // x = a / |a|
      g9[6 + r] = (float)(red[10 + r] / w1d);
    }
    // ---- Adam (torch.optim.Adam single-tensor path, betas 0.9 / 0.999, eps 1e-8), then lr *= 0.999 ----
    b1p *= 0.9; b2p *= 0.999;
    const double bc1 = 1.0 - b1p, bc2 = 1.0 - b2p;
    const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
    const float w_lerp = (float)(1.0 - 0.9), beta2 = (float)0.999, omb2 = (float)(1.0 - 0.999);
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      m1[e] = m1[e] + w_lerp * (g9[e] - m1[e]);
      m2[e] = m2[e] * beta2 + omb2 * g9[e] * g9[e];
      const float den = sqrtf(m2[e]) / bc2s + 1e-8f;
      const float upd = -step_size * (m1[e] / den);
      if (e < 6) a[e] += upd; else tr[e - 6] += upd;
    }
    lr *= 0.999;
    if (fabs((double)loss_prev - (double)lossf) < (double)loss_prev * ratio) {
      if (++brk >= max_break) { broke = true; break; }
    }
    loss_prev = lossf;
  }
  if (threadIdx.x == 0) {
    rot_from_6d(a, Rf, aux);
    for (int e = 0; e < 9; ++e) Rout[(size_t)pair * 9 + e] = Rf[e];
    for (int r = 0; r < 3; ++r) tout[(size_t)pair * 3 + r] = tr[r];
    stats[(size_t)pair * 3 + 0] = (float)(broke ? it : max_iter - 1);
    stats[(size_t)pair * 3 + 1] = lossf;
    stats[(size_t)pair * 3 + 2] = (float)brk;
  }
}

// =========================================================================================
// launchers
// =========================================================================================
static inline int next_pow2(int n) { int m = 1; while (m < n) m <<= 1; return m; }

size_t nms_scratch_floats(int B, int N) { return (size_t)B * ((size_t)4 * N + kNmsHdr); }

// tune.nms_binned: 1 = grid-binned candidates for N >= 1024 on grids of >= 128 workgroups (default), 2 = whenever N >= 1024, 0 = all pairs
hipError_t launch_nms_keys(const Tuning& tune, const float* src, const float* scores, float* keys, int B, int N, float R, hipStream_t s,
                           float* scratch, const PairTab* ptab, long total_rows, bool keys_preset) {
  // The reference tests sqrt(d2) >= R (PointDSC.py:283).  sqrtf is monotone, so that is d2 >= t for the smallest float t
  // with sqrtf(t) >= R; finding t on the host removes the square root from the N^2 loop without changing one decision.
  float t = R * R;
  if (R > 0.f) {
    while (sqrtf(t) >= R && t > 0.f) t = nextafterf(t, 0.f);
    while (sqrtf(t) < R) t = nextafterf(t, INFINITY);
  } else {
    t = 0.f;
  }
  const int nblk = (N + 255) / 256;
  // (on small grids - B = 1 - the two binned launches cost more latency than the candidate-split all-pairs kernel saves)
  if (scratch && tune.nms_binned && N >= 1024 && (nblk * B >= 128 || tune.nms_binned == 2) && R > 0.f && std::isfinite(R)) {
    const float inv_cell = 1.0f / (1.01f * R);
    hipLaunchKernelGGL(k_nms_bin, dim3(B), dim3(1024), 0, s, src, scores, scratch, N, inv_cell, ptab);
    hipLaunchKernelGGL(k_nms_keys_binned, dim3(nblk, B), dim3(256), 0, s, src, scores, scratch, keys, N, t, inv_cell, ptab);
    return hipGetLastError();
  }
  int js = 1;
  if (nblk * B < 256) js = std::min(std::min(16, nblk), (512 + nblk * B - 1) / (nblk * B));
  if (js > 1 && !keys_preset) {
    hipError_t e = hipMemcpyAsync(keys, scores, (size_t)(ptab ? total_rows : (long)B * N) * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_nms_keys, dim3(nblk, B, js), dim3(256), 0, s, src, scores, keys, N, t, ptab);
  return hipGetLastError();
}

// tune.topk_select: radix select + rank placement (default) or the full bitonic sort
hipError_t launch_sort_topk(const Tuning& tune, const float* keys, int* out_idx, int B, int N, int S, hipStream_t s, const PairTab* ptab) {
  const int M = next_pow2(N < 2 ? 2 : N);
  constexpr size_t kSelectLds = 156 * 1024;        // dynamic LDS the select kernel may use (160 KiB minus its static arrays)
  const size_t need = (size_t)N * 4 + 8 + (size_t)S * 8;
  if (S < 1 || S > N) return hipErrorInvalidValue;
  if (tune.topk_select && need <= kSelectLds) {
    {   // per device and cheap: no process-wide "already set" flag
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_select_topk<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)kSelectLds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_select_topk<true>, dim3(B), dim3(1024), need, s, keys, out_idx, N, S, ptab);
    return hipGetLastError();
  }
  if (M > 16384 || (tune.topk_select && need > kSelectLds)) {
    // [r4] any N: the keys stay in global memory, only the S winners live in the LDS (S <= 19 968, i.e. N < 200 000 at ratio 0.1)
    if ((size_t)S * 8 > kSelectLds) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_select_topk<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kSelectLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select_topk<false>, dim3(B), dim3(1024), (size_t)S * 8, s, keys, out_idx, N, S, ptab);
    return hipGetLastError();
  }
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_topk), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_sort_topk, dim3(B), dim3(1024), (size_t)M * 8, s, keys, out_idx, N, M, S, ptab);
  return hipGetLastError();
}

hipError_t launch_knn_seeds(const float* feat_n, const int* seeds, const float* dist_in, int* knn_idx, int B, int N, int S,
                            int k, hipStream_t s, const PairTab* ptab) {
  const int ld = ((N + 31) / 32) * 32;             // row stride of dist_in (launch_seed_dist pads the rows to whole tiles)
  if (dist_in && N > 256 * 64) {                   // [r4] rows of any length: streamed selection
    hipLaunchKernelGGL(k_knn_select_stream, dim3(S, B), dim3(256), 0, s, dist_in, knn_idx, N, S, k, ptab, ld);
    return hipGetLastError();
  }
  if ((size_t)N * 4 > 150 * 1024) return hipErrorInvalidValue;
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_seeds), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
  }
  // EPT = row elements per thread (registers): the smallest instantiation that holds the row - fewer registers, more rows in flight
  if (dist_in && N <= 256 * 8) {
    hipLaunchKernelGGL(k_knn_select_fast<8>, dim3(S, B), dim3(256), 0, s, dist_in, knn_idx, N, S, k, ptab, ld);
    return hipGetLastError();
  }
  if (dist_in && N <= 256 * 20) {
    hipLaunchKernelGGL(k_knn_select_fast<20>, dim3(S, B), dim3(256), 0, s, dist_in, knn_idx, N, S, k, ptab, ld);
    return hipGetLastError();
  }
  if (dist_in && N <= 256 * 32) {
    hipLaunchKernelGGL(k_knn_select_fast<32>, dim3(S, B), dim3(256), 0, s, dist_in, knn_idx, N, S, k, ptab, ld);
    return hipGetLastError();
  }
  if (dist_in && N <= 256 * 64) {
    hipLaunchKernelGGL(k_knn_select_fast<64>, dim3(S, B), dim3(256), 0, s, dist_in, knn_idx, N, S, k, ptab, ld);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_knn_seeds, dim3(S, B), dim3(256), (size_t)N * 4, s, feat_n, seeds, dist_in, knn_idx, N, S, k, ptab, ld);
  return hipGetLastError();
}

hipError_t launch_seed_power(const float* feat_n, const float* src, const float* tgt, const int* knn_idx, float* snaps,
                             unsigned char* conv, double* hsum, int B, int N, int S, int k, int iters, float sigma,
                             float sigma_d, hipStream_t s, const PairTab* ptab, const float* sigma_dev) {
  if (k > kKMax) return hipErrorInvalidValue;
  const size_t lds_bytes = (size_t)(kKMax * 8 + kKMax + k * (k + 1)) * sizeof(float);
  hipLaunchKernelGGL(k_seed_power, dim3(S, B), dim3(64), lds_bytes, s, feat_n, src, tgt, knn_idx, snaps, conv, hsum, N, S, k,
                     iters, 1.0f / (sigma * sigma), 1.0f / (sigma_d * sigma_d), ptab, sigma_dev);
  return hipGetLastError();
}

hipError_t launch_seed_kabsch(const float* src, const float* tgt, const int* knn_idx, const float* snaps,
                              const unsigned char* conv, float* seed_T, int B, int N, int S, int k, int iters,
                              const double* hsum, int* stop_scratch, hipStream_t s, const PairTab* ptab) {
  const int* stop_batch = nullptr;
  if (B > 1 && stop_scratch && !ptab) { // the reference's allclose spans the batch (a ragged batch is B independent B = 1 calls)
    hipLaunchKernelGGL(k_stop_iteration, dim3(1), dim3(1024), 0, s, conv, B * S, iters, stop_scratch);
    stop_batch = stop_scratch;
  }
  hipLaunchKernelGGL(k_seed_kabsch, dim3((S + 63) / 64, B), dim3(64), 0, s, src, tgt, knn_idx, snaps, conv, seed_T, N, S, k, iters,
                     hsum, stop_batch, ptab);
  return hipGetLastError();
}

hipError_t launch_stop_iteration(const unsigned char* conv, int B, int S, int iters, int* stop_out, hipStream_t s) {
  hipLaunchKernelGGL(k_stop_iteration, dim3(1), dim3(1024), 0, s, conv, B * S, iters, stop_out);
  return hipGetLastError();
}

hipError_t launch_score_hyp(const float* src, const float* tgt, const float* seed_T, int* counts, int B, int N, int S,
                            float tau, hipStream_t s, const PairTab* ptab) {
  // sqrtf(d2) < tau  <=>  d2 < t for the smallest float t with sqrtf(t) >= tau (sqrtf is monotone): no square root per pair
  float t = tau * tau;
  if (tau > 0.f) {
    while (sqrtf(t) >= tau && t > 0.f) t = nextafterf(t, 0.f);
    while (sqrtf(t) < tau) t = nextafterf(t, INFINITY);
  } else {
    t = 0.f;
  }
  hipLaunchKernelGGL(k_score_hyp, dim3((S + kHypPerWG - 1) / kHypPerWG, B), dim3(256), 0, s, src, tgt, seed_T, counts, N, S, t, ptab);
  return hipGetLastError();
}

hipError_t launch_finalize_pose(const float* src, const float* tgt, const float* seed_T, const int* counts, float* fitness,
                                float* final_T, float* labels, int* best, int B, int N, int S, float tau, float refine_thr,
                                int refine_iters, hipStream_t s, const PairTab* ptab) {
  hipLaunchKernelGGL(k_finalize_pose, dim3(B), dim3(N <= 2048 ? 256 : 1024), 0, s, src, tgt, seed_T, counts, fitness, final_T, labels, best, N, S,
                     tau, refine_thr, refine_iters, ptab);
  return hipGetLastError();
}

hipError_t launch_post_refine(const float* T_in, const float* src, const float* tgt, float* T_out, int B, int N, float thr,
                              int iters, hipStream_t s) {
  hipLaunchKernelGGL(k_post_refine, dim3(B), dim3(N <= 2048 ? 256 : 1024), 0, s, T_in, src, tgt, T_out, N, thr, iters);
  return hipGetLastError();
}

hipError_t launch_rigid_transform(const float* A, const float* Bp, const float* w, float* T, int n, int k,
                                  float weight_threshold, hipStream_t s) {
  hipLaunchKernelGGL(k_rigid_transform, dim3(n), dim3(64), 0, s, A, Bp, w, T, k, weight_threshold);
  return hipGetLastError();
}

hipError_t launch_weighted_procrustes(const float* X, const float* Y, const float* w, const int* offsets, int B, float eps,
                                      float* R, float* t, hipStream_t s) {
  hipLaunchKernelGGL(k_weighted_procrustes, dim3(B), dim3(1024), 0, s, X, Y, w, offsets, eps, R, t);
  return hipGetLastError();
}

hipError_t launch_global_registration(const float* X, const float* Y, const float* w, const int* offsets, int B, float eps,
                                      float qsize, int max_iter, int max_break, double ratio, float* R, float* t,
                                      float* stats, int max_n, hipStream_t s) {
  // 16 register-resident points per thread cover N <= 16 x threads; fewer waves make the per-step reduction cheaper
  const int threads = max_n <= 4096 ? 256 : max_n <= 8192 ? 512 : 1024;
  hipLaunchKernelGGL(k_global_registration, dim3(B), dim3(threads), 0, s, X, Y, w, offsets, eps, qsize, max_iter, max_break, ratio,
                     R, t, stats);
  return hipGetLastError();
}

}  // namespace gmf
