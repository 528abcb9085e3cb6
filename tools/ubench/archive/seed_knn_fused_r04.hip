// ARCHIVED (round 4, measured and dropped; not part of the library, not compiled): the k nearest neighbours of the seed rows
// WITHOUT the S x N distance rows in HBM ("fuse k_seed_dist + k_knn_select_fast, 276 us; the distance rows never
// need to reach HBM").  Built into pose_kernels.hip behind gmf_set_tuning("knn_fused"), correct at the first run - neighbour lists,
// seed hypotheses and poses IDENTICAL to the two-kernel form on (B, N) = (1, 45) ... (32, 5000), (2, 16384), ragged batches, and a
// batch with 800 duplicated feature rows that takes the overflow fall-back - and NOT faster (us per launch at 32 x 500 seeds x 5000 keys):
//
//   two-kernel form                       k_seed_dist 128-130   + k_knn_select_fast 148-155                      = 277-285
//   fused, 32 seeds per wave              k_seed_scan<1> 100    + k_seed_scan<2> 119   + k_knn_rank 46  (+ thresh) = 270+ ; step 17.27 vs 17.29 ms
//   fused, 64 seeds per wave, pooled      k_seed_scan<1>  81    + k_seed_scan<2> 149   + k_knn_rank 56  (+ thresh) = 290+ ; step 17.27 vs 17.20 ms
//   B = 1, N = 5000                       1.191 ms against 1.152 ms (six launches instead of two)
//
// The premise was wrong: k_seed_dist is NOT bound by writing its 320 MB (its scan without any store still takes 100 us, 81 us with
// every key tile shared by 256 seeds) - it is a 20-tile latency chain per workgroup (seed-fragment gather, ring start, a barrier per
// 24 / 48 MFMAs) at 40-50 % matrix-pipe use, and the fused form runs that chain TWICE (group minima -> threshold, then candidates)
// plus a candidate pass whose per-tile LDS atomics and predicated stores cost as much again.  What it saves (640 MB of HBM traffic)
// was never the bottleneck.  Kept here for the design (exact selection from 64 group minima, deterministic candidate slots, the
// overflow fall-back); the pose head's remaining 0.76 ms is a chain of latency-bound kernels, not a bandwidth problem.
//
// ---- kernels (namespace gmf, pose_kernels.hip) ----------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// [r4] kNN of the seed rows WITHOUT the S x N distance matrix in HBM (common.py:53-75 + PointDSC.py:327-329).
// k_seed_dist wrote 320 MB of distance rows per batch (32 x 500 x 5000 fp32) which k_knn_select_fast read back: 276 us of the
// pose head, both bound by that traffic.  The selection only needs, per seed row, (a) a threshold T that bounds its (k + 1)-th
// smallest distance and (b) the few elements <= T.  k_seed_scan computes the SAME distances as k_seed_dist (the same MFMA
// sequence on the same operands: bit-identical values) twice and keeps them in registers:
//   MODE 1: per seed 64 GROUP minima (group = chunk x lane half x a slice of the lane's (tile parity, register) positions);
//           k_seed_thresh takes T = the (k + 1)-th smallest of them - k + 1 different groups each hold an element <= T, so the
//           (k + 1)-th smallest distance of the row is <= T (the argument of k_knn_select_fast, with other groups);
//   MODE 2: elements <= T are collected per (seed, chunk) in the LDS (one LDS atomic per lane and tile that has any; ~8 per
//           list at 8 chunks, room for kSeedSubCap) and leave as that chunk's fixed slot of the seed's candidate list - no
//           global atomic, a deterministic layout; k_knn_rank orders a seed's candidates by (distance, index) and writes
//           ranks 1 .. k (rank 0 = the row itself is dropped).
// The result is a function of the distance row alone, i.e. identical to the two-kernel form.  A (seed, chunk) list that
// overflows (massive exact ties: duplicated feature rows) sets a flag, and
//   MODE 0: = k_seed_dist (rows to HBM) followed by k_knn_select_fast, both launched behind the flag, redo the whole batch.
// grid (ceil(S / 128), B * chunks), block 256; dist rows [B, S, 32 tiles] (MODE 0), gmin [B, S, 64], thr [B, S],
// nsub [B, S, chunks] + 1 flag word, cand [B, S, chunks, kSeedSubCap] x (distance, index)
// ---------------------------------------------------------------------------------------
constexpr int kSeedSubCap = 30;                  // slots of a (seed, chunk) sub-list in global memory
constexpr int kScanRing = 3;                     // MODE 2: 3 ring slots (48 KiB) + the candidate pool (28 KiB) + 256 counters: two workgroups per CU
constexpr int kScanPool = 3584;                  // candidates a workgroup (256 seeds x its chunk of keys) can hold: ~1000-2000 arrive

// A wave owns 64 seeds (two fragments, lane (h, i): seeds i and 32 + i of the wave): every key fragment read from the LDS feeds
// six MFMAs and a key tile (16 KiB through L2 -> LDS) serves 256 seeds - the 32-seed form was bound by that stream (328 MB per
// pass at 32 x 500 x 5000: 100 us per pass).
template <int MODE>
__global__ void __launch_bounds__(256, 2)
k_seed_scan(const float* __restrict__ featn_img, const int* __restrict__ seeds, float* __restrict__ dist, float* __restrict__ gmin,
            const float* __restrict__ thr, int* __restrict__ nsub, float2* __restrict__ cand, int* __restrict__ flag, int N, int tiles,
            int S, int chunks, const PairTab* __restrict__ ptab) {
  constexpr int NB = MODE == 2 ? kScanRing : 4;
  __shared__ __attribute__((aligned(16))) float lds[NB * kStageFloats + (MODE == 2 ? kScanPool * 2 + 256 + 4 : 0)];
  if (MODE == 0 && *flag == 0) return;                        // (no overflow: nothing to redo)
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y / chunks, chunk = blockIdx.y - pair * chunks;   // key tiles are split over `chunks` workgroups
  const int Smax = S;
  N = pair_rows(ptab, pair, N);
  S = ptab ? ptab[pair].S : S;
  const int tiles_p = (N + 31) >> 5;
  if ((int)blockIdx.x * 256 >= S) return;                     // (uniform per workgroup, before any barrier)
  const int per = (tiles_p + chunks - 1) / chunks;
  const int t0 = chunk * per, t1 = min(tiles_p, t0 + per);
  const int seed_base = blockIdx.x * 256 + wave * 64;
  const float* pair_img = featn_img + (size_t)pair * tiles * (32 * 128);
  const int GPL = 32 / chunks;                                // group slots per lane and seed (chunks is a power of two <= 16)
  const int GW = min(GPL, 16);                                // ... of which the lane fills GW from its 16 register positions
  int my[2]; size_t sidx[2]; int row[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    my[f] = seed_base + 32 * f + i;
    sidx[f] = (size_t)pair * Smax + my[f];
    row[f] = (my[f] < S) ? seeds[sidx[f]] : 0;
  }
  if (t0 >= t1) {                                             // an empty chunk (uniform over the workgroup): its groups hold nothing
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (MODE == 1 && my[f] < S)
        for (int g = 0; g < GPL; ++g) gmin[sidx[f] * 64 + chunk * (2 * GPL) + h * GPL + g] = INFINITY;
      if (MODE == 2 && my[f] < S && h == 0) nsub[sidx[f] * chunks + chunk] = 0;
    }
    return;
  }
  f16x8 sh[2][8], sl[2][8];                                   // the seeds' rows as the two fp16 planes of the image (k_pack_rows_h2)
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const f16x8* rp = reinterpret_cast<const f16x8*>(pair_img + (size_t)(row[f] >> 5) * kStageFloats) + h * 32 + (row[f] & 31);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) { sh[f][s8] = rp[(0 * 8 + s8) * 64]; sl[f][s8] = rp[(1 * 8 + s8) * 64]; }
  }
  float T[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) T[f] = (MODE == 2 && my[f] < S) ? thr[sidx[f]] : -INFINITY;
  float2* const pool = reinterpret_cast<float2*>(lds + NB * kStageFloats);                    // [kScanPool] (distance, key | seed << 14 | position << 22)
  int* const lcnt = reinterpret_cast<int*>(lds + NB * kStageFloats + kScanPool * 2);          // [256] per-seed counts, then [1] the pool's fill
  if (MODE == 2) {                                            // (ordered before the first use by the ring's first barrier)
    lcnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) lcnt[256] = 0;
  }
  float m[2][16];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int q = 0; q < 16; ++q) m[f][q] = INFINITY;
  StageRing<NB> ss;
  ss.init(lds, wave, lane, pair_img + (size_t)t0 * kStageFloats, t1 - t0);
  ss.prime();
  const int ld = tiles * 32;
  bool over = false;
  for (int t = t0; t < t1; ++t) {
    const f16x8* lk = reinterpret_cast<const f16x8*>(ss.acquire());
    f32x16 acc[2];
    acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {            // (per seed the products in k_seed_dist's order: s_l k_h, s_h k_l, s_h k_h - bit-identical distances)
      const f16x8 kh = lk[(0 * 8 + s8) * 64], kl = lk[(1 * 8 + s8) * 64];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        acc[f] = mfma_h16(kh, sl[f][s8], acc[f]);
        acc[f] = mfma_h16(kl, sh[f][s8], acc[f]);
        acc[f] = mfma_h16(kh, sh[f][s8], acc[f]);
      }
    }
    // register r = key 32 t + 8 (r >> 2) + 4 h + (r & 3); keys >= N (the zero rows that pad the last tile) do not exist
    const int jb = 32 * t + 4 * h;
    const bool full = 32 * t + 32 <= N;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (MODE == 0) {
        if (my[f] < S) {
          float4* out = reinterpret_cast<float4*>(dist + sidx[f] * ld + 4 * h + (size_t)t * 32);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            out[2 * q] = make_float4(2.0f - 2.0f * acc[f][4 * q], 2.0f - 2.0f * acc[f][4 * q + 1], 2.0f - 2.0f * acc[f][4 * q + 2], 2.0f - 2.0f * acc[f][4 * q + 3]);
        }
      } else if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = 2.0f - 2.0f * acc[f][r];
          m[f][r] = fminf(m[f][r], (full || jb + 8 * (r >> 2) + (r & 3) < N) ? d : INFINITY);
        }
      } else {
        float d[16];
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          d[r] = 2.0f - 2.0f * acc[f][r];
          mask |= (d[r] <= T[f] && (full || jb + 8 * (r >> 2) + (r & 3) < N)) ? (1u << r) : 0u;
        }
        if (mask) {                               // (~13 of the 1024 elements of a seed fragment per tile)
          const int n = __popc(mask), ls = wave * 64 + 32 * f + i;
          int pos = atomicAdd(&lcnt[ls], n);      // position within the seed's (chunk) list
          int slot = atomicAdd(&lcnt[256], n);    // position within the workgroup's pool
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (mask & (1u << r)) {
              if (pos < kSeedSubCap && slot < kScanPool)
                pool[slot] = make_float2(d[r], __int_as_float((jb + 8 * (r >> 2) + (r & 3)) | (ls << 14) | (pos << 22)));
              else over = true;
              ++pos; ++slot;
            }
          }
        }
      }
    }
  }
  if (MODE == 1) {
    // fold the lane's 16 position minima to its GW groups (16 / GW consecutive positions each); a lane has GPL slots (GPL = 32 with
    // one chunk: the upper 16 stay empty)
    const int lw = __builtin_ctz(16 / GW);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (my[f] >= S) continue;
      for (int g = 0; g < GPL; ++g) {
        float v = INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) v = ((q >> lw) == g) ? fminf(v, m[f][q]) : v;
        gmin[sidx[f] * 64 + chunk * (2 * GPL) + h * GPL + g] = v;
      }
    }
  }
  if (MODE == 2) {
    if (__any(over)) { if (lane == 0) *flag = 1; }
    __syncthreads();                              // (every lane has appended)
    // the pool's entries go to their (seed, chunk) slot of the global candidate list; the per-seed counts beside them
    const int fill = min(lcnt[256], kScanPool);
    const size_t wg_seed0 = (size_t)pair * Smax + blockIdx.x * 256;
    for (int e = threadIdx.x; e < fill; e += 256) {
      const float2 v = pool[e];
      const int w = __float_as_int(v.y), j = w & 0x3fff, ls = (w >> 14) & 255, pos = w >> 22;
      cand[((wg_seed0 + ls) * chunks + chunk) * kSeedSubCap + pos] = make_float2(v.x, __int_as_float(j));
    }
    const int sd = blockIdx.x * 256 + threadIdx.x;
    if (sd < S) nsub[((size_t)pair * Smax + sd) * chunks + chunk] = min(lcnt[threadIdx.x], kSeedSubCap);
  }
}

// T[seed] = the (k + 1)-th smallest of the seed's 64 group minima under (value, group index); one wave per seed, lane = group.
// Also clears the overflow flag of the candidate pass that follows.   grid (ceil(S / 4), B), block 256
__global__ void __launch_bounds__(256)
k_seed_thresh(const float* __restrict__ gmin, float* __restrict__ thr, int* __restrict__ flag, int S, int k, const PairTab* __restrict__ ptab) {
  const int pair = blockIdx.y, s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *flag = 0;
  if (s >= pair_seeds(ptab, pair, S)) return;
  const float v = gmin[((size_t)pair * S + s) * 64 + lane];
  int rank = 0;
#pragma unroll
  for (int b = 0; b < 64; ++b) {
    const float o = __shfl(v, b, 64);
    rank += (o < v || (o == v && b < lane)) ? 1 : 0;
  }
  // NaN minima cannot occur (fminf drops NaNs); with +inf ties the index order still gives exactly one lane the rank k
  if (rank == k) thr[(size_t)pair * S + s] = v;
}

// ranks of a seed's candidates under (distance, index) -> its k neighbours; one wave per seed, four seeds per workgroup.
// grid (ceil(S / 4), B), block 256
__global__ void __launch_bounds__(256)
k_knn_rank(const int* __restrict__ nsub, const float2* __restrict__ cand, int* __restrict__ knn_idx, int N, int S, int k, int chunks,
           const PairTab* __restrict__ ptab) {
  __shared__ float cv[4][16 * kSeedSubCap];
  __shared__ int ci[4][16 * kSeedSubCap];
  const int pair = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, s = blockIdx.x * 4 + wave;
  const bool live = s < pair_seeds(ptab, pair, S);
  const size_t sidx = (size_t)pair * S + (live ? s : 0);
  int* out = knn_idx + sidx * k;
  N = pair_rows(ptab, pair, N);
  // a row with NaN distances yields fewer than k + 1 candidates: every slot holds an in-range index before the ranks are written
  if (live && lane < k) out[lane] = min(lane + 1, N - 1);
  // lane c < chunks: the length of sub-list c and, by a wave prefix sum, its offset in the seed's list
  const int n_c = (live && lane < chunks) ? min(nsub[sidx * chunks + lane], kSeedSubCap) : 0;
  int inc = n_c;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  const int c = __shfl(inc, 15, 64);              // (chunks <= 16)
  for (int ch = 0; ch < chunks; ++ch) {
    const int n = __shfl(n_c, ch, 64), o0 = __shfl(inc, ch, 64) - n;
    if (lane < n) {
      const float2 v = cand[(sidx * chunks + ch) * kSeedSubCap + lane];
      cv[wave][o0 + lane] = v.x; ci[wave][o0 + lane] = __float_as_int(v.y);
    }
  }
  __syncthreads();
  if (!live) return;
  for (int p = lane; p < c; p += 64) {
    const float pv = cv[wave][p]; const int pi = ci[wave][p];
    int rank = 0;
    for (int q = 0; q < c; ++q) { const float qv = cv[wave][q]; const int qi = ci[wave][q]; rank += (qv < pv || (qv == pv && qi < pi)) ? 1 : 0; }
    if (rank >= 1 && rank <= k) out[rank - 1] = pi;
  }
}


// ---- launcher ---------------------------------------------------------------------------------------------------------------
// [r4] the fused form of launch_seed_dist + launch_knn_seeds (k + 1 <= 64, N <= 16 384): see k_seed_scan.
// ws: seed_knn_ws_floats(B, S) floats; dist: the [B, S, 32 tiles] rows the fall-back writes (only touched when a list overflows).
size_t seed_knn_ws_floats(int B, int S) { return (size_t)B * S * (2 * 16 * kSeedSubCap + 64 + 1 + 16) + 64; }

hipError_t launch_seed_knn_fused(const float* featn_img, const int* seeds, float* dist, float* ws, int* knn_idx, int B, int N, int S, int k,
                                 hipStream_t s, const PairTab* ptab) {
  const int tiles = (N + 31) / 32;
  const int sblocks = (S + 255) / 256;
  int chunks = tiles >= 2 ? 2 : 1;                 // (>= 2 chunks x 2 lane halves x 16 positions = the 64 groups of a seed; one tile: N <= 32)
  while (chunks < 16 && sblocks * B * chunks < 1024 && tiles / (2 * chunks) >= 4) chunks *= 2;
  const size_t BS = (size_t)B * S;
  float2* cand = reinterpret_cast<float2*>(ws);                   // [B, S, chunks, kSeedSubCap] (distance, index)
  float* gmin = ws + BS * chunks * kSeedSubCap * 2;               // [B, S, 64]
  float* thr = gmin + BS * 64;                                    // [B, S]
  int* nsub = reinterpret_cast<int*>(thr + BS);                   // [B, S, chunks]
  int* flag = nsub + BS * chunks;
  const dim3 g(sblocks, B * chunks);
  hipLaunchKernelGGL(k_seed_scan<1>, g, dim3(256), 0, s, featn_img, seeds, (float*)nullptr, gmin, (const float*)nullptr, nsub, cand, flag, N, tiles, S, chunks, ptab);
  hipLaunchKernelGGL(k_seed_thresh, dim3((S + 3) / 4, B), dim3(256), 0, s, (const float*)gmin, thr, flag, S, k, ptab);
  hipLaunchKernelGGL(k_seed_scan<2>, g, dim3(256), 0, s, featn_img, seeds, (float*)nullptr, gmin, (const float*)thr, nsub, cand, flag, N, tiles, S, chunks, ptab);
  hipLaunchKernelGGL(k_knn_rank, dim3((S + 3) / 4, B), dim3(256), 0, s, (const int*)nsub, (const float2*)cand, knn_idx, N, S, k, chunks, ptab);
  // fall-back behind the overflow flag (normally both return at once): the two-kernel form over the whole batch
  hipLaunchKernelGGL(k_seed_scan<0>, g, dim3(256), 0, s, featn_img, seeds, dist, gmin, (const float*)nullptr, nsub, cand, flag, N, tiles, S, chunks, ptab);
  const int ld = tiles * 32;
  if (N <= 256 * 8) hipLaunchKernelGGL(k_knn_select_fast<8>, dim3(S, B), dim3(256), 0, s, (const float*)dist, knn_idx, N, S, k, ptab, ld, (const int*)flag);
  else if (N <= 256 * 20) hipLaunchKernelGGL(k_knn_select_fast<20>, dim3(S, B), dim3(256), 0, s, (const float*)dist, knn_idx, N, S, k, ptab, ld, (const int*)flag);
  else if (N <= 256 * 32) hipLaunchKernelGGL(k_knn_select_fast<32>, dim3(S, B), dim3(256), 0, s, (const float*)dist, knn_idx, N, S, k, ptab, ld, (const int*)flag);
  else hipLaunchKernelGGL(k_knn_select_fast<64>, dim3(S, B), dim3(256), 0, s, (const float*)dist, knn_idx, N, S, k, ptab, ld, (const int*)flag);
  return hipGetLastError();
}

