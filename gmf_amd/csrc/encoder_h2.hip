// Linear stages of the encoder on the f16 MFMA with split-fp16 operands ("fp16x2": two planes, three partial
// products, fp32 accumulate; arithmetic notes in mfma_core.hpp).  Same math, same fusion and same blob/stage
// sizes as the fp32-MFMA kernels in encoder_kernels.hip - a 32 x K weight block is 4 B per element either way -
// but 8 x 3 MFMAs of 32 cycles per 32 x 128 block instead of 64 MFMAs of 64 cycles.
//   k_front_h2        PointDSC.py:88,104-109,56-58     [layer0] + PointCN + Q'/K/V projections -> fp16x2 images
//   k_fusion_attn_h2  fusion_layer.py:119-121,44,84-94,190
//   k_fusion_ff_h2p   fusion_layer.py:54-69,191
//   k_ctx_prep_h2     fusion_layer.py:124-126,46-49,86-87
#include <algorithm>
#include <cstdlib>
#include "enc_common.hpp"
#include "enc_ff.hpp"
#include "launchers.hpp"

namespace gmf {

#ifndef GMF_H2_RING
#define GMF_H2_RING 2
#endif
constexpr int kRing = GMF_H2_RING;   // LDS ring depth of the weight / context stage streams (16 KiB each)

// =========================================================================================
// k_front_h2: stages (16 x 16 KiB, fp16x2 images): Wp[4] | Wq'[4] | Wk[4] | Wv[4]
//   vecs (fp32): bp | bq' | bk | bv | b0 | W0 image (fp32, K=8: layer0 stays on the f32 MFMA, 4 MFMAs per block)
//   outputs: f as fp32 P32 image; Q', K, V as fp16x2 plane images (16 KiB per tile).
// =========================================================================================
// (bx, pair): the workgroup's row block and pair; zsel = -1 (all outputs) or the ONE output (0 = Q' (+ f), 1 = K, 2 = V) it
// produces.  `lds`: kRing * 16 KiB owned by the workgroup.
template <int MODE>
GMF_DEVINL void front_h2_body(float* lds, const int bx, const int pair, const int zsel_in, const float* __restrict__ in,
                              const float* __restrict__ wst, const float* __restrict__ vecs, float* __restrict__ f_out,
                              float* __restrict__ q_out, float* __restrict__ k_out, float* __restrict__ v_out, int N, int tiles,
                              const PairTab* __restrict__ ptab = nullptr, unsigned* __restrict__ v_scale = nullptr,
                              const PvGuard guard = {}) {
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t row0 = pair_row0(ptab, pair, N);            // ragged batch (MODE 3 only): corr_pos is packed [sum n, 6]
  N = pair_rows(ptab, pair, N);
  const int tiles_p = ptab ? (N + 31) >> 5 : tiles;
  if (bx * kWavesPerWG >= tiles_p) return;                 // (uniform per workgroup, before any barrier)
  const int tile_raw = bx * kWavesPerWG + wave;
  const bool active = tile_raw < tiles_p;
  const int tile = active ? tile_raw : tiles_p - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);

  // gridDim.z == 3 (small grids): workgroup z recomputes PointCN (MODE 0 / 1) and produces ONE of Q' (z = 0, it also stores
  // f), K, V - 8 (MODE 2: 4) weight stages instead of 16 (12) per workgroup, three times the workgroups.
  // MODE 3: corr_pos -> layer0 -> PointCN -> f only (the first layer's f when the later PointCNs run in the attention epilogue).
  const bool zsplit = (MODE != 3) && zsel_in >= 0;
  const int zsel = zsplit ? zsel_in : -1;
  StageRing<kRing> ss;
  if (MODE == 3) ss.init(lds, wave, lane, wst, 4);
  else if (MODE == 2 && zsplit) ss.init(lds, wave, lane, wst + (4 + 4 * zsel) * kStageFloats, 4);
  else if (MODE == 2) ss.init(lds, wave, lane, wst + 4 * kStageFloats, 12);
  else if (zsplit) ss.init(lds, wave, lane, wst, 4, wst + (4 + 4 * zsel) * kStageFloats, 4);
  else ss.init(lds, wave, lane, wst, 16);
  ss.prime();

  float f[CF];
  FragH2<8> fx;
  {
    float x[CF];
    if (MODE == 1 || MODE == 3) {
      const int row = tile * 32 + i;
      float pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 4 * h + e;
        pk[e] = (row < N && k < 6) ? in[(row0 + row) * 6 + k] : 0.f;
      }
      const float4* w0 = reinterpret_cast<const float4*>(vecs + 5 * C) + lane;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f32x16 acc = zero16();
        const float4 w = w0[mb * 64];
        acc = mfma32(w.x, pk[0], acc); acc = mfma32(w.y, pk[1], acc);
        acc = mfma32(w.z, pk[2], acc); acc = mfma32(w.w, pk[3], acc);
        float b[16];
        load_vec_block(b, vecs + 4 * C, mb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[16 * mb + r] = acc[r] + b[r];
      }
    } else {
      load_frag_p32<CF>(x, in + toff, lane);
    }
    if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < CF; ++e) f[e] = x[e];
    } else {
      fx.set(x);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const f16x8* lw = as_h2(ss.acquire());
        f32x16 acc = zero16();
        mma_wx_h2<8>(acc, lw, fx);
        float b[16];
        load_vec_block(b, vecs, mb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) f[16 * mb + r] = relu_nan(fmaf(acc[r], kH2Inv, b[r]));
      }
    }
  }
  if (active && zsel <= 0 && MODE != 2) store_frag_p32<CF>(f_out + toff, f, lane);
  if (MODE == 3) {
    // [r5] the first layer's "pv_fp8" statistic: max row |f_0|^2 of the pair (PvGuard; later layers: the attention epilogue)
    if (guard.stat_next && active) {
      float ssq = 0.f;
#pragma unroll
      for (int e = 0; e < CF; ++e) ssq = fmaf(f[e], f[e], ssq);
      pv_stat_raise(guard.stat_next, pair, ssq, tile * 32 + i < N, lane);
    }
    return;
  }
  fx.set(f);

  // e4m3 cross planes (V: per (feature, key tile); [r5] K: per (key, 32-channel block)) or the low fp16 planes - uniform per pair and layer
  const bool v8 = pv_planes_on(v_scale, guard, pair);
  unsigned* const sw_tile = v_scale ? v_scale + ((size_t)pair * tiles + tile) * 128 + lane : nullptr;     // [V: 64 lanes | K: 64 lanes]
  unsigned ksw = 0;
#pragma unroll
  for (int which = 0; which < 2; ++which) {   // Q', K
    if (zsplit && zsel != which) continue;
    float* dst = (which == 0 ? q_out : k_out) + toff;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const f16x8* lw = as_h2(ss.acquire());
      f32x16 acc = zero16();
      mma_wx_h2<8>(acc, lw, fx);
      float b[16], t[16];
      load_vec_block(b, vecs + (1 + which) * C, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, b[r]);
      if (active) {
        if (which == 1 && v8) store_block_v8(dst, mb, t, lane, ksw);
        else store_block_h2(dst, mb, t, lane);
      }
    }
    if (which == 1 && active && v8) sw_tile[64] = ksw;
  }
  if (zsplit && zsel != 2) return;
  unsigned vsw = 0;
#pragma unroll
  for (int db = 0; db < 4; ++db) {             // V (feature on lane)
    const f16x8* lw = as_h2(ss.acquire());
    f32x16 acc = zero16();
    mma_xw_h2<8>(acc, lw, fx);
    const float bv = vecs[3 * C + 32 * db + i];
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bv);
    if (active) {
      if (v8) store_block_v8(v_out + toff, db, t, lane, vsw);     // the attention's pv_fp8 form (CompatCache::v_scale)
      else store_block_h2(v_out + toff, db, t, lane);
    }
  }
  if (active && v8) sw_tile[0] = vsw;
}

template <int MODE>
__global__ void __launch_bounds__(256, 2)
k_front_h2(const float* __restrict__ in, const float* __restrict__ wst, const float* __restrict__ vecs,
           float* __restrict__ f_out, float* __restrict__ q_out, float* __restrict__ k_out,
           float* __restrict__ v_out, int N, int tiles, const PairTab* __restrict__ ptab, unsigned* __restrict__ v_scale,
           const PvGuard guard) {
  __shared__ __attribute__((aligned(16))) float lds[kRing * kStageFloats];
  front_h2_body<MODE>(lds, blockIdx.x, blockIdx.y, gridDim.z == 3 ? (int)blockIdx.z : -1, in, wst, vecs, f_out, q_out, k_out, v_out,
                      N, tiles, ptab, v_scale, guard);
}

// =========================================================================================
// k_ctx_prep_h2: output per token tile (4096 floats): Kc | Vc as fp16x2 images for d_head = 64:
//   Kc: unit ((plane*4 + s)*64 + lane), 2 planes x 4 k-steps = 2048 floats ; Vc: unit ((plane*4 + slot)*64 + lane),
//   slot = 2*db + s2, 2048 floats.   stages (4): Wk[2] | Wv[2] as fp16x2 images.
// =========================================================================================
// a row of a ROW-MAJOR [n_rows, 128] tensor as a fragment (zero beyond n_rows): the kernels that read the caller's tokens
// directly instead of a packed image (one launch less where launches are what a pass costs)
GMF_DEVINL void load_frag_rowmajor(float (&x)[CF], const float* __restrict__ rows_base, int row, int n_rows, int h) {
  const float4* p = reinterpret_cast<const float4*>(rows_base + (size_t)min(row, n_rows - 1) * C) + h;
  const bool ok = row < n_rows;
#pragma unroll
  for (int g = 0; g < CF / 4; ++g) {                 // fragment group g = features 32 (g >> 2) + 8 (g & 3) + 4 h .. + 3
    const float4 t = p[8 * (g >> 2) + 2 * (g & 3)];
    x[4 * g + 0] = ok ? t.x : 0.f; x[4 * g + 1] = ok ? t.y : 0.f; x[4 * g + 2] = ok ? t.z : 0.f; x[4 * g + 3] = ok ? t.w : 0.f;
  }
}

// (bx, pair, set): the workgroup's block of four token tiles, its pair (of nb) and its weight set; `lds`: 2 x 16 KiB.
template <bool PE, bool ROWMAJOR = false>
GMF_DEVINL void ctx_prep_h2_body(float* lds, const int bx, const int pair, const int set, const int nb, const float* __restrict__ ctx,
                                 const float* __restrict__ wst, const float* __restrict__ vecs, float* __restrict__ out, int T,
                                 int ttiles, int wst_stride, int vec_stride) {
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_raw = bx * kWavesPerWG + wave;
  const bool active = tile_raw < ttiles;
  const int tile = active ? tile_raw : ttiles - 1;
  wst += (size_t)set * wst_stride;
  vecs += (size_t)set * vec_stride;
  const float* pair_base = ROWMAJOR ? ctx + (size_t)pair * T * C : ctx + (size_t)pair * ttiles * (32 * C);
  float* dst = out + (((size_t)set * nb + pair) * ttiles + tile) * kStageFloats;

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst, 4);
  ss.prime();
  FragH2<8> cx;
  {
    float x[CF], cn[CF];
    if (PE) lcpe_frag(x, pair_base, tile * 32 + i, T, vecs, h);
    else if (ROWMAJOR) load_frag_rowmajor(x, pair_base, tile * 32 + i, T, h);
    else load_frag_p32<CF>(x, pair_base + (size_t)tile * (32 * C), lane);
    layernorm_frag<CF>(cn, x, vecs + 4 * C, vecs + 5 * C, h);
    cx.set(cn);
  }
  f16x8* kdst = reinterpret_cast<f16x8*>(dst);
  f16x8* vdst = reinterpret_cast<f16x8*>(dst + 32 * DH);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const f16x8* lw = as_h2(ss.acquire());
    f32x16 acc = zero16();
    mma_wx_h2<8>(acc, lw, cx);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
    if (active) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f16x8 hi, lo;
        split8h(&t[8 * half], hi, lo);
        const int s = 2 * mb + half;
        kdst[(0 * 4 + s) * 64 + lane] = hi;
        kdst[(1 * 4 + s) * 64 + lane] = lo;
      }
    }
  }
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const f16x8* lw = as_h2(ss.acquire());
    f32x16 acc = zero16();
    mma_xw_h2<8>(acc, lw, cx);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
    if (active) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 hi, lo;
        split8h(&t[8 * s2], hi, lo);
        const int slot = 2 * db + s2;
        vdst[(0 * 4 + slot) * 64 + lane] = hi;
        vdst[(1 * 4 + slot) * 64 + lane] = lo;
      }
    }
  }
}

template <bool PE, bool ROWMAJOR = false>
__global__ void __launch_bounds__(256, 2)
k_ctx_prep_h2(const float* __restrict__ ctx, const float* __restrict__ wst, const float* __restrict__ vecs,
              float* __restrict__ out, int T, int ttiles, int wst_stride, int vec_stride) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  ctx_prep_h2_body<PE, ROWMAJOR>(lds, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, ctx, wst, vecs, out, T, ttiles, wst_stride, vec_stride);
}

// =========================================================================================
// k_fusion_attn_h2: stages: Wq''[2] | ctx tiles [ttiles] (Kc | Vc fp16x2 images from k_ctx_prep_h2) | Wo[2]
//   vecs: query taps | gamma | beta | bo (fp32).  P is scaled by 2^10 as in k_scattn_h2.
// =========================================================================================
constexpr int kFattnLdsFloats = kRing * kStageFloats + 7 * C + kWavesPerWG * 2 * C;   // ring | vectors | halo rows

template <bool PE, bool ROWMAJOR = false>
GMF_DEVINL void fusion_attn_h2_body(float* lds, const int bx, const int pair, const float* __restrict__ xin,
                                    const float* __restrict__ ctx_img, const float* __restrict__ wst,
                                    const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles,
                                    const int tiles_stride = 0) {    // [r5] ragged batch: N, tiles are the PAIR's own, its slot in the images keeps this stride
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_raw = bx * kWavesPerWG + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const int ts = tiles_stride ? tiles_stride : tiles;
  const float* pair_base = ROWMAJOR ? xin + (size_t)pair * N * C : xin + (size_t)pair * ts * (32 * C);
  const size_t toff = ((size_t)pair * ts + tile) * (32 * C);

  // behind the ring (kFattnLdsFloats): the kernel's per-feature vectors and, per wave, the two halo rows of the LCPE - requested
  // before the ring's stages (k_linear_h2 has the reasons: a tap / gamma / bias fetched from global memory where it is used is a
  // memory round trip in the middle of the chain, and a lone wave per SIMD has nothing to hide it behind)
  float* const lvec = lds + kRing * kStageFloats;
  float* const halo = lvec + 7 * C + wave * (2 * C);
  dma_vec(vecs, lvec, 7 * C, wave, kWavesPerWG, lane);
  if (PE) LcpeHalo<CF>::issue(pair_base, tile, tiles, halo, lane);
  StageRing<kRing> ss;
  ss.init(lds, wave, lane, wst, 2,
          ctx_img + (size_t)pair * ttiles * kStageFloats, ttiles, wst + 2 * kStageFloats, 2);
  ss.prime();

  float xp[CF];
  static_assert(!(PE && ROWMAJOR), "the row-major form is the Fusion-1 instance (no LCPE)");
  if (ROWMAJOR) load_frag_rowmajor(xp, pair_base, tile * 32 + i, N, h);
  else load_frag_p32<CF>(xp, pair_base + (size_t)tile * (32 * C), lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (PE) LcpeHalo<CF>::apply(xp, halo, lvec, tile * 32 + i, N, lane);

  FragH2<4> qx;
  {
    FragH2<8> nx;
    {
      float xn[CF];
      layernorm_frag<CF>(xn, xp, lvec + 4 * C, lvec + 5 * C, h);
      nx.set(xn);
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const f16x8* lw = as_h2(ss.acquire());
      f32x16 acc = zero16();
      mma_wx_h2<8>(acc, lw, nx);
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
      qx.set_block(mb, t);
    }
  }

  f32x16 oacc[2];
  oacc[0] = zero16(); oacc[1] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  for (int t = 0; t < ttiles; ++t) {
    const f16x8* lk = as_h2(ss.acquire());
    const f16x8* lv = lk + 2 * 4 * 64;            // Vc image follows the Kc image (2 planes x 4 steps)
    f32x16 s = zero16();
    mma_wx_h2<4>(s, lk, qx);
    float x[16];
    float mx = -INFINITY;
    const int jbase = t * 32 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = 8 * (r >> 2) + (r & 3);
      const float v = (jbase + jl < T) ? s[r] : -INFINITY;
      x[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const float m_off = m_new - 10.0f;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_off); ls += x[r]; }
    l_half = fmaf(l_half, alpha, ls);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int slot = 2 * db + s2;
        mma3(oacc[db], lv[(0 * 4 + slot) * 64], lv[(1 * 4 + slot) * 64], ph, pl);
      }
    }
  }
  FragH2<4> ox;
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = oacc[db][r] * inv;
      ox.set_block(db, t);
    }
  }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const f16x8* lw = as_h2(ss.acquire());
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int mb = 2 * st + hb;
      f32x16 acc = zero16();
      mma_wx_h2<4>(acc, lw + hb * (2 * 4 * 64), ox);     // a 32 x 64 block = 2 planes x 4 steps x 64 units
      float b[16], t[16];
      load_vec_block(b, lvec + 6 * C, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, b[r]) + xp[16 * mb + r];
      if (active) store_block_p32(x1_out + toff, mb, t, lane);
    }
  }
}

template <bool PE, bool ROWMAJOR = false>
__global__ void __launch_bounds__(256, 2)
k_fusion_attn_h2(const float* __restrict__ xin, const float* __restrict__ ctx_img, const float* __restrict__ wst,
                 const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles) {
  __shared__ __attribute__((aligned(16))) float lds[kFattnLdsFloats];
  fusion_attn_h2_body<PE, ROWMAJOR>(lds, blockIdx.x, blockIdx.y, xin, ctx_img, wst, vecs, x1_out, N, tiles, T, ttiles);
}

// [r5] fusion_attn_tile_h2_body: the cross-attention role of the small-grid layer with ONE WORKGROUP PER QUERY TILE, its four waves
// dealing the CONTEXT TILES among themselves (wave w: tiles w, w + 4, ...) - fusion_layer.py:119-121,44,84-94,190 as
// fusion_attn_h2_body, where a wave walks all T / 32 context tiles of its own query tile one after the other: at B = 1 that walk
// (10 tiles at T = 300) is the longest link of the layer's first launch (19.8 us of which ~10 are the walk; the launch is a third of a
// forward at N = 1000).  Here every wave forms LCPE + LayerNorm + q for the tile (the same values, redundantly - they are a chain,
// not work), streams its own context tiles STRAIGHT INTO REGISTERS (their fragments are 16-byte units in the image: no LDS ring, no
// stage barriers, all of a wave's tiles requested while it still computes q), keeps its own running (m, l, O), and the four partial
// results meet in the LDS: common maximum, waves added in index order.  The out-projection's four 32-feature blocks then run one
// per wave.  LDS: Wq'' | Wo (4 stages, fetched once, behind ONE barrier) | vectors | halo rows | the partials (over Wq'').
// Results: the same arithmetic per product; the softmax's running maximum is per wave and the partial sums are added in another order
// than the sequential walk's, so x1 agrees with fusion_attn_h2_body to fp32 rounding, not bit for bit.
constexpr int kFattnTileLdsFloats = 4 * kStageFloats + 7 * C + kWavesPerWG * 2 * C + kWavesPerWG * 128;

GMF_DEVINL void fusion_attn_tile_h2_body(float* lds, const int tile, const int pair, const float* __restrict__ xin,
                                         const float* __restrict__ ctx_img, const float* __restrict__ wst,
                                         const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles,
                                         const int tiles_stride = 0) {    // [r5] ragged batch: as fusion_attn_h2_body
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ts = tiles_stride ? tiles_stride : tiles;
  const float* pair_base = xin + (size_t)pair * ts * (32 * C);
  const size_t toff = ((size_t)pair * ts + tile) * (32 * C);
  float* const lvec = lds + 4 * kStageFloats;
  float* const halo = lvec + 7 * C + wave * (2 * C);
  float* const lml = lvec + 7 * C + kWavesPerWG * (2 * C);         // [wave][lane][2]: running maximum, row sum (this lane half's)
  float* const lpart = lds;                                        // [wave][8][64] float4: the partial O (over Wq'', dead by then)
  // this wave's context tiles: fragments of tile t are the 16 units ((plane * 4 + s) * 64 + lane) of Kc and of Vc (k_ctx_prep_h2)
  const f16x8* const gctx = reinterpret_cast<const f16x8*>(ctx_img + (size_t)pair * ttiles * kStageFloats) + lane;
  // (Kc of the next tile is requested at the top of a tile into a second set of registers, Vc of the next tile into the SAME registers
  // right after the last product that reads them: 96 fragment registers instead of 128, which spilled)
  f16x8 kf[8], vf[8], kn[8];
  auto fetch_k = [&](int t, f16x8 (&k8)[8]) {
    const f16x8* g = gctx + (size_t)min(t, ttiles - 1) * (kStageFloats / 4);      // (a tile of 4096 floats = 1024 units)
#pragma unroll
    for (int u = 0; u < 8; ++u) k8[u] = g[u * 64];
  };
  auto fetch_v = [&](int t, f16x8 (&v8)[8]) {
    const f16x8* g = gctx + (size_t)min(t, ttiles - 1) * (kStageFloats / 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = g[(8 + u) * 64];
  };
  dma_vec(vecs, lvec, 7 * C, wave, kWavesPerWG, lane);
  LcpeHalo<CF>::issue(pair_base, tile, tiles, halo, lane);
#pragma unroll
  for (int st = 0; st < 4; ++st) dma_4k_s(wst + (size_t)st * kStageFloats + wave * 1024, lds + st * kStageFloats + wave * 1024, (unsigned)lane * 16u);
  float xp[CF];
  load_frag_p32<CF>(xp, pair_base + (size_t)tile * (32 * C), lane);
  fetch_k(wave, kf);
  fetch_v(wave, vf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LcpeHalo<CF>::apply(xp, halo, lvec, tile * 32 + i, N, lane);

  FragH2<4> qx;
  {
    FragH2<8> nx;
    {
      float xn[CF];
      layernorm_frag<CF>(xn, xp, lvec + 4 * C, lvec + 5 * C, h);
      nx.set(xn);
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const f16x8* lw = reinterpret_cast<const f16x8*>(lds + mb * kStageFloats) + lane;
      f32x16 acc = zero16();
      mma_wx_h2<8>(acc, lw, nx);
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
      qx.set_block(mb, t);
    }
  }
  float xres[16];                                  // the residual of this wave's output block (x' of features 32 wave ..)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = xp[r];
#pragma unroll
    for (int mb = 1; mb < 4; ++mb) v = (wave == mb) ? xp[16 * mb + r] : v;
    xres[r] = v;
  }

  f32x16 oacc[2];
  oacc[0] = zero16(); oacc[1] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  for (int t = wave; t < ttiles; t += kWavesPerWG) {
    const bool more = t + kWavesPerWG < ttiles;
    if (more) fetch_k(t + kWavesPerWG, kn);
    f32x16 s = zero16();
#pragma unroll
    for (int ss4 = 0; ss4 < 4; ++ss4) mma3(s, kf[ss4], kf[4 + ss4], qx.h[ss4], qx.l[ss4]);
    float x[16];
    float mx = -INFINITY;
    const int jbase = t * 32 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = 8 * (r >> 2) + (r & 3);
      const float v = (jbase + jl < T) ? s[r] : -INFINITY;
      x[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const float m_off = m_new - 10.0f;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_off); ls += x[r]; }
    l_half = fmaf(l_half, alpha, ls);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int slot = 2 * db + s2;
        mma3(oacc[db], vf[slot], vf[4 + slot], ph, pl);
      }
    }
    if (more) {
      fetch_v(t + kWavesPerWG, vf);
#pragma unroll
      for (int u = 0; u < 8; ++u) kf[u] = kn[u];
    }
  }
  // ---- the four partial results meet: O_w, m_w, l_w -> LDS (every wave is past its use of Wq'': the barrier orders it) ----
  __syncthreads();
  {
    float4* pw = reinterpret_cast<float4*>(lpart) + (size_t)wave * 8 * 64 + lane;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        pw[(4 * db + q) * 64] = make_float4(oacc[db][4 * q], oacc[db][4 * q + 1], oacc[db][4 * q + 2], oacc[db][4 * q + 3]);
    reinterpret_cast<float2*>(lml)[wave * 64 + lane] = make_float2(m_run, l_half);
  }
  __syncthreads();
  FragH2<4> ox;
  {
    float mw[kWavesPerWG], lw_[kWavesPerWG], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < kWavesPerWG; ++w) {
      const float2 ml = reinterpret_cast<const float2*>(lml)[w * 64 + lane];
      mw[w] = ml.x; lw_[w] = ml.y;
      M = fmaxf(M, ml.x);
    }
    float l = 0.f, wk[kWavesPerWG];
#pragma unroll
    for (int w = 0; w < kWavesPerWG; ++w) { wk[w] = __builtin_amdgcn_exp2f(mw[w] - M); l = fmaf(lw_[w], wk[w], l); }
    const float inv = 1.0f / xhalf_sum(l);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
      for (int w = 0; w < kWavesPerWG; ++w) {
        const float4* pr = reinterpret_cast<const float4*>(lpart) + (size_t)w * 8 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = pr[(4 * db + q) * 64];
          t[4 * q] = fmaf(v.x, wk[w], t[4 * q]); t[4 * q + 1] = fmaf(v.y, wk[w], t[4 * q + 1]);
          t[4 * q + 2] = fmaf(v.z, wk[w], t[4 * q + 2]); t[4 * q + 3] = fmaf(v.w, wk[w], t[4 * q + 3]);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] *= inv;
      ox.set_block(db, t);
    }
  }
  // ---- out-projection: output block mb = wave (stage 2 + (wave >> 1), half wave & 1), + bo + x' ----
  {
    const int mb = wave;
    const f16x8* lw = reinterpret_cast<const f16x8*>(lds + (2 + (mb >> 1)) * kStageFloats) + lane;
    f32x16 acc = zero16();
    mma_wx_h2<4>(acc, lw + (mb & 1) * (2 * 4 * 64), ox);
    float b[16], t[16];
    load_vec_block(b, lvec + 6 * C, mb, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, b[r]) + xres[r];
    store_block_p32(x1_out + toff, mb, t, lane);
  }
}

// k_small_front_fattn: small grids (e.g. B = 1, the reference's evaluation mode) - the first of the three launches of a layer:
// workgroups z = 0, 1, 2 project Q', K, V from f (front_h2_body<2>, one output each), workgroups z = 3 run LCPE + LayerNorm +
// cross-attention -> x1 (fusion_attn_h2_body).  The two are independent given f; as separate launches on these latency-
// bound grids they cost their sum (8.7 + 18.6 us per layer at B = 1, N = 5000), in one launch the longer of the two.
__global__ void __launch_bounds__(256, 2)
k_small_front_fattn(const float* __restrict__ f_in, const float* __restrict__ front_wst, const float* __restrict__ front_vec,
                    const float* __restrict__ ctx_img, const float* __restrict__ attn_wst, const float* __restrict__ attn_vec,
                    float* __restrict__ q_out, float* __restrict__ k_out, float* __restrict__ v_out, float* __restrict__ x1_out,
                    int N, int tiles, int T, int ttiles, unsigned* __restrict__ v_scale, const PvGuard guard,
                    const PairTab* __restrict__ ptab) {                  // [r5] ragged batches: the pairs' own rows
  // gridDim.z == 4: z = 3 is the cross-attention of the block's four query tiles, one per wave (fusion_attn_h2_body);
  // gridDim.z == 7 [r5]: z = 3 .. 6 are ONE query tile each, its context tiles dealt to the four waves (fusion_attn_tile_h2_body)
  __shared__ __attribute__((aligned(16))) float lds[kFattnTileLdsFloats];
  static_assert(kFattnTileLdsFloats >= kFattnLdsFloats, "one buffer serves both forms of the role");
  if (blockIdx.z < 3) {
    front_h2_body<2>(lds, blockIdx.x, blockIdx.y, blockIdx.z, f_in, front_wst, front_vec, nullptr, q_out, k_out, v_out, N, tiles,
                     ptab, v_scale, guard);
    return;
  }
  // ragged batch: the LCPE's zero padding sits at the pair's OWN last row, query tiles beyond its own have nothing to do (uniform
  // per workgroup, before any barrier); the pair's slot in the images keeps the stride `tiles`
  const int Np = pair_rows(ptab, blockIdx.y, N), tiles_p = ptab ? (Np + 31) >> 5 : tiles;
  if (gridDim.z == 4) {
    if ((int)blockIdx.x * kWavesPerWG >= tiles_p) return;
    fusion_attn_h2_body<true>(lds, blockIdx.x, blockIdx.y, f_in, ctx_img, attn_wst, attn_vec, x1_out, Np, tiles_p, T, ttiles, tiles);
  } else {
    const int tile = (int)blockIdx.x * kWavesPerWG + ((int)blockIdx.z - 3);
    if (tile < tiles_p) fusion_attn_tile_h2_body(lds, tile, blockIdx.y, f_in, ctx_img, attn_wst, attn_vec, x1_out, Np, tiles_p, T, ttiles, tiles);
  }
}

// =========================================================================================
// k_fusion_ff_h2p: LayerNorm + Linear(128 -> 1024) + GEGLU + Linear(512 -> 128) + residual (fusion_layer.py:54-69,191) with
//   the chunk loop software-pipelined inside each wave.
//   stages (48 x 16 KiB): for c in 0..15: W1a_c | W1g_c | W2_c (4 blocks of 32 x 32: 2 planes x 2 steps)
//   A GEGLU chunk is 48 MFMAs of W1 (value | gate, K = 128), 16 GELUs (~24 vector instructions each) and 24 MFMAs of
//   W2; in program order they run back to back and the kernel costs matrix time PLUS vector time.  Here the W1 MFMAs
//   of chunk c+1 are issued three at a time with one GELU of chunk c in their issue gaps (16 units), the fp16 split
//   of the gated values rides in the later units, and only the 24 W2 MFMAs run bare.  The biases start the
//   accumulators (no separate add), the stage ring is 4 deep and its acquire is branch-free (stage order
//   A0 G0 | A1 G1 W2_0 | ... | A15 G15 W2_14 | W2_15 addressed into the unchanged blob; past the end the last stage is
//   re-fetched into a free slot instead of branching), so a whole chunk is one basic block for the scheduler.
// =========================================================================================
template <bool SPLIT>
__global__ void __launch_bounds__(256, 2)
k_fusion_ff_h2p(const float* __restrict__ x1, const float* __restrict__ wst, const float* __restrict__ vecs,
                float* __restrict__ x2_out, int tiles, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStageFloats];
  fusion_ff_h2p_body<SPLIT>(lds, blockIdx.x, blockIdx.y, gridDim.y, blockIdx.z, gridDim.z, x1, wst, vecs, x2_out, tiles, part);
}

// =========================================================================================
// [r5] Small grids: the forward's prologue as three launches of two roles each.  Before the first layer the forward runs two
// INDEPENDENT chains of small kernels on one stream, one after the other:
//   image side:  Fusion-1 context (k_ctx_prep_h2) -> Fusion-1 cross-attention -> Fusion-1 feed-forward (-> reduce -> the L context sets)
//   point side:  key points (k_pack_pts8) -> compat cache (k_compat_build) -> layer 0 + first PointCN (k_front_h2<3>)
// - each a handful of workgroups on 256 CUs (B = 1, T = 300: 3 workgroups per image-side kernel), so a launch costs its
// latency chain, not its work.  Here one launch carries ONE LINK OF EACH chain: the point side (28 us at N = 1000, 45 at 5000)
// runs under the image side's three links.  The roles call the kernels' own bodies: same arithmetic, bit-identical results.
// Workgroups [0, n_a) take the image-side role (first: the longer chain), the rest the point-side role.
// =========================================================================================
__global__ void __launch_bounds__(256, 2)
k_pro_ctx_pts(const int n_a, const int gx_a, const int nb, const float* __restrict__ ctx, const float* __restrict__ wst,
              const float* __restrict__ vecs, float* __restrict__ out, int T, int ttiles,
              const float* __restrict__ src, const float* __restrict__ tgt, float* __restrict__ pts8, int N, int Npad, long total,
              const PairTab* __restrict__ ptab, unsigned* __restrict__ zero_words, int n_zero) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int id = blockIdx.x;
  if (id < n_a) {
    ctx_prep_h2_body<false, true>(lds, id % gx_a, id / gx_a, 0, nb, ctx, wst, vecs, out, T, ttiles, 0, 0);
    return;
  }
  pack_pts8_body((long)(id - n_a) * 256 + threadIdx.x, src, tgt, pts8, N, Npad, total, ptab, zero_words, n_zero);
}

constexpr int kProBLdsFloats = kFattnLdsFloats > kCompatLdsFloats ? kFattnLdsFloats : kCompatLdsFloats;
__global__ void __launch_bounds__(256, 2)
k_pro_fattn_compat(const int n_a, const int gx_a, const float* __restrict__ xin, const float* __restrict__ ctx_img,
                   const float* __restrict__ wst, const float* __restrict__ vecs, float* __restrict__ x1_out, int T, int ttiles,
                   const float* __restrict__ pts8, float* __restrict__ c_dense, int N, int tiles, int gy, float inv_sig2,
                   const PairTab* __restrict__ ptab) {
  __shared__ __attribute__((aligned(16))) float lds[kProBLdsFloats];
  const int id = blockIdx.x;
  if (id < n_a) {
    fusion_attn_h2_body<false, true>(lds, id % gx_a, id / gx_a, xin, ctx_img, wst, vecs, x1_out, T, ttiles, T, ttiles);
    return;
  }
  const int j = id - n_a, I = j % tiles, rest = j / tiles;
  compat_build_body<0>(lds, I, rest % gy, rest / gy, pts8, c_dense, N, tiles, inv_sig2, ptab);
}

constexpr int kProCLdsFloats = (kRing > 4 ? kRing : 4) * kStageFloats;
// MINB = 1: grids of at most one workgroup per CU (the two roles' union wants a few registers over 256: with 512 there is no scratch);
// MINB = 2: larger ones, two workgroups per CU as the kernels of both roles run on their own (SPLIT: 12 bytes of scratch per lane)
template <bool SPLIT, int MINB>
__global__ void __launch_bounds__(256, MINB)
k_pro_ff_front(const int n_a, const int gx_a, const int nb, const int hs, const float* __restrict__ x1, const float* __restrict__ wst,
               const float* __restrict__ vecs, float* __restrict__ x2_out, int ttiles, float* __restrict__ part,
               const int gx_f, const float* __restrict__ in, const float* __restrict__ fwst, const float* __restrict__ fvecs,
               float* __restrict__ f_out, float* __restrict__ q_out, float* __restrict__ k_out, float* __restrict__ v_out, int N,
               int tiles, const PairTab* __restrict__ ptab, const PvGuard guard) {
  __shared__ __attribute__((aligned(16))) float lds[kProCLdsFloats];
  const int id = blockIdx.x;
  if (id < n_a) {                                   // (bx, pair, z) with bx fastest
    const int bx = id % gx_a, r = id / gx_a;
    fusion_ff_h2p_body<SPLIT>(lds, bx, r % nb, nb, r / nb, hs, x1, wst, vecs, x2_out, ttiles, part);
    return;
  }
  const int j = id - n_a;
  front_h2_body<3>(lds, j % gx_f, j / gx_f, -1, in, fwst, fvecs, f_out, q_out, k_out, v_out, N, tiles, ptab, nullptr, guard);
}

// =========================================================================================
// k_linear_h2: every linear stage of one encoder layer in ONE pass over f = ReLU(PointCN(feat)) [B, N, 128]
// (PointDSC.py:56-58, fusion_layer.py:172-201):
//     Q' = Wq' f + bq', K = Wk f + bk, V = Wv f + bv                          -> split-fp16 images for the attention kernel
//     x' = LCPE(f) ; x1 = x' + Wo softmax(LN(x') Wq'' Kc^T) Vc + bo           (cross-attention over the T context tokens)
//     x2 = x1 + W2 GEGLU(W1 LN(x1) + b1) + b2                                 -> fp32 image (the Fusion-2 branch of the block sum)
// i.e. k_front_h2<2> + k_fusion_attn_h2<true> + k_fusion_ff_h2p with x' and x1 never leaving the registers: the residuals
// START the accumulators of the following projection (256 (x' + bo), 256 (x1 + b2) - the weight images are 256 W), so they
// cost no registers across the stage that follows, and the only HBM traffic of the layer's linear part is f (+ two LCPE
// neighbour rows per tile) in and Q', K, V, x2 out (2.5 KB per row instead of 5 KB in the three-kernel form).
//   stages: Wq' Wk Wv [12] | Wq'' [2] | context tiles [ttiles] | Wo [2] through a 2-slot ring, then the 48 feed-forward
//   stages through the 4-slot ring of ff_chunks (the same 64 KiB of LDS; one workgroup barrier between the two).
// =========================================================================================
// PART 0: the whole kernel.  PART 1 / PART 2: its two independent halves as workgroup ROLES of one launch (k_linear_roles, grids
// of 256 .. ~300 workgroups: one workgroup per CU cannot hide the 71-stage chain of a wave, two roles per row block shorten
// it to its longer half) - 1 = the Q'/K/V projections, 2 = Fusion-2 (LCPE + cross-attention + feed-forward).
// LDS of the fused linear kernel: 4 stages | attention vectors | feed-forward vectors | 4 waves x 2 halo rows | the three projection
// biases = 78.5 KiB (two workgroups per CU)
constexpr int kLinLdsFloats = 4 * kStageFloats + 7 * C + (3 * C + 2 * FFH) + kWavesPerWG * 2 * C + 3 * C;

// QSKIP [r4]: the Q' projection is left to the attention kernel's prologue (scattn_h2p_body, qf_img): 8 instead of 12 projection
// stages, 32 instead of 48 KiB of stores per wave in the kernel's write-bound phase
template <int PART, int NP = 3, bool QSKIP = false>
GMF_DEVINL void linear_h2_body(float* lds, const float* __restrict__ f_in, const float* __restrict__ front_wst,
                               const float* __restrict__ front_vec, const float* __restrict__ ctx_img,
                               const float* __restrict__ attn_wst, const float* __restrict__ attn_vec,
                               const float* __restrict__ ff_wst, const float* __restrict__ ff_vec, float* __restrict__ q_out,
                               float* __restrict__ k_out, float* __restrict__ v_out, float* __restrict__ x2_out, int N, int tiles,
                               int T, int ttiles, const PairTab* __restrict__ ptab = nullptr,
                               unsigned* __restrict__ v_scale = nullptr,     // non-null: V with e4m3 cross planes (store_block_v8) ...
                               const PvGuard guard = {}) {                   // ... for the pairs the guard lets through (pv_planes_on)
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  // ragged batch: the pair's own rows and tiles - the LCPE's zero padding sits at ITS last row, row blocks beyond its tiles have
  // nothing to do (uniform per workgroup, before any barrier); the slot in every image keeps the stride `tiles`
  if (ptab) N = ptab[pair].n;
  const int tiles_p = ptab ? (N + 31) >> 5 : tiles;
  if ((int)blockIdx.x * kWavesPerWG >= tiles_p) return;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < tiles_p;
  const int tile = active ? tile_raw : tiles_p - 1;
  const float* pair_base = f_in + (size_t)pair * tiles * (32 * C);
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);

  // 4-slot ring, three stages in flight.  The waits are COUNTED (vmcnt counts loads, LDS-DMA pieces and stores alike, in
  // issue order): during the Q'/K/V stages a wave has, behind the pieces of the stage it acquires, the pieces of the next two
  // stages and the 4 stores of each of the last three stages - vmcnt(20) in steady state - so neither the younger stages nor
  // the stores (whose acknowledgement takes a memory round trip) are waited for at a stage barrier.  For the counts to hold
  // on every wave, the padding waves of a pair's last workgroup store too: they recompute the pair's last tile and write the
  // same bits to the same place as the wave that owns it.
  // Behind the ring (kLinLdsFloats): the per-feature vectors of the attention (7 x 128 floats) and of the feed-forward
  // (1408 floats) and, per wave, the two halo rows of the LCPE - requested first, so every later counted wait covers them.
  // (A bias / gamma / tap fetched from global memory where it is used is a memory round trip in the middle of a stage chain.)
  float* const lvec_a = lds + 4 * kStageFloats;
  float* const lvec_f = lvec_a + 7 * C;
  float* const halo = lvec_f + (3 * C + 2 * FFH) + wave * (2 * C);
  float* const lvec_p = lvec_f + (3 * C + 2 * FFH) + kWavesPerWG * (2 * C);      // bq' | bk | bv
  if (PART != 2) dma_vec(front_vec + C, lvec_p, 3 * C, wave, kWavesPerWG, lane);
  if (PART != 1) {
    dma_vec(attn_vec, lvec_a, 7 * C, wave, kWavesPerWG, lane);
    dma_vec(ff_vec, lvec_f, 3 * C + 2 * FFH, wave, kWavesPerWG, lane);
    LcpeHalo<CF>::issue(pair_base, tile, tiles_p, halo, lane);
  }
  StageRing<4> ss;
  if (PART == 0) ss.init(lds, wave, lane, front_wst + (QSKIP ? 8 : 4) * kStageFloats, QSKIP ? 8 : 12, attn_wst, 2,
                         ctx_img + (size_t)pair * ttiles * kStageFloats, ttiles, attn_wst + 2 * kStageFloats, 2);
  else if (PART == 1) ss.init(lds, wave, lane, front_wst + 4 * kStageFloats, 12);
  else ss.init(lds, wave, lane, attn_wst, 2, ctx_img + (size_t)pair * ttiles * kStageFloats, ttiles, attn_wst + 2 * kStageFloats, 2);

  // ---- Q', K, V from f ---------------------------------------------------------------------------------------------
  // [r4] f is loaded ONCE and stays in registers for the Fusion-2 half (PART 0): a reload behind the 48 stores of this phase
  // retires in order behind them, i.e. only when the last store has been acknowledged (stamped: 1 500 - 27 000 cycles).
  float f[CF];
  if (PART != 2) {
    FragH2<8> fx;
    load_frag_p32<CF>(f, f_in + toff, lane);
    fx.set(f);
    // [r4] the three bias vectors come from the LDS copy (lvec_p): held in registers they were 128 VGPRs, and the compiler - which
    // counts its own loads but not the ring's DMA pieces - put `s_waitcnt vmcnt(35 .. 32)` in front of every stage's bias add,
    // i.e. a wait for the store acknowledgements of four stages back in the middle of every epilogue from stage 5 on.
    ss.prime();                                  // AFTER the load above: its wait then leaves the three primed stages in flight
    const bool v8 = pv_planes_on(v_scale, guard, pair);          // [r5] decides the K image's cross planes as well as V's
    unsigned* const sw_tile = v_scale ? v_scale + ((size_t)pair * tiles + tile) * 128 + lane : nullptr;     // [V: 64 lanes | K: 64 lanes]
    unsigned ksw = 0;
#pragma unroll
    for (int which = (QSKIP ? 1 : 0); which < 2; ++which) {   // Q', K
      float* dst = (which == 0 ? q_out : k_out) + toff;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        constexpr int kYounger[4] = {8, 12, 16, 20};
        const int sidx = 4 * (which - (QSKIP ? 1 : 0)) + mb;   // stage index (compile-time: the loops are unrolled)
        // (role 1 has no stages behind its last ones, so the counts of the steady state do not hold there: plain acquire)
        const f16x8* lw = as_h2(PART == 1 ? ss.acquire()
                                : sidx == 0 ? ss.acquire_counted<kYounger[0]>() : sidx == 1 ? ss.acquire_counted<kYounger[1]>()
                                : sidx == 2 ? ss.acquire_counted<kYounger[2]>() : ss.acquire_counted<kYounger[3]>());
        f32x16 acc = zero16();
        mma_wx_h2n<8, NP>(acc, lw, fx);
        float t[16], bb[16];
        load_vec_block(bb, lvec_p + which * C, mb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bb[r]);
        if (which == 1 && v8) store_block_v8(dst, mb, t, lane, ksw);     // (4 stores of 16 bytes either way)
        else store_block_h2(dst, mb, t, lane);
      }
    }
    unsigned vsw = 0;
#pragma unroll
    for (int db = 0; db < 4; ++db) {             // V (feature on lane)
      const f16x8* lw = as_h2(PART == 1 ? ss.acquire() : ss.acquire_counted<20>());
      f32x16 acc = zero16();
      mma_xw_h2n<8, NP>(acc, lw, fx);
      float t[16];
      const float bvd = lvec_p[2 * C + 32 * db + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bvd);
      if (v8) store_block_v8(v_out + toff, db, t, lane, vsw);     // (4 stores of 16 bytes either way: the counted waits hold)
      else store_block_h2(v_out + toff, db, t, lane);
    }
    // one more store than the counted waits of the stages that follow assume: their counts are lower bounds, it only makes them
    // wait for one operation more
    if (v_scale) { sw_tile[0] = vsw; sw_tile[64] = ksw; }     // (written either way: one store count for both forms)
  }

  if (PART == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (PART == 2) ss.prime();
  // ---- cross-attention: x1 = x' + Wo softmax(q Kc^T) Vc + bo --------------------------------------------------------
  f32x16 x1a[4];                                 // 256 x1, accumulator layout (block mb = features 32 mb .. 32 mb + 31)
  {
    FragH2<4> qx;
    {
      float xp[CF];
      if (PART == 2) load_frag_p32<CF>(xp, f_in + toff, lane);
      else {
#pragma unroll
        for (int k = 0; k < CF; ++k) xp[k] = f[k];
      }
      if (PART == 2) {                           // (PART 0: the barriers of the Q'/K/V stages have made the vectors visible)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      LcpeHalo<CF>::apply(xp, halo, lvec_a, tile * 32 + i, N, lane);
      {
        FragH2<8> nx;
        {
          float xn[CF];
          layernorm_frag<CF>(xn, xp, lvec_a + 4 * C, lvec_a + 5 * C, h);
          nx.set(xn);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const f16x8* lw = as_h2(ss.acquire());
          f32x16 acc = zero16();
          mma_wx_h2n<8, NP>(acc, lw, nx);
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
          qx.set_block(mb, t);
        }
      }
      // the residual and the output bias start the out-projection accumulators (Wo images are 256 W)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        float b[16];
        load_vec_block(b, lvec_a + 6 * C, mb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) x1a[mb][r] = (xp[16 * mb + r] + b[r]) * 256.0f;
      }
    }
    f32x16 oacc[2];
    oacc[0] = zero16(); oacc[1] = zero16();
    float m_run = -INFINITY, l_half = 0.f;
    for (int t = 0; t < ttiles; ++t) {
      const f16x8* lk = as_h2(ss.acquire());
      const f16x8* lv = lk + 2 * 4 * 64;            // Vc image follows the Kc image (2 planes x 4 steps)
      f32x16 sc = zero16();
      mma_wx_h2n<4, NP>(sc, lk, qx);
      float x[16];
      float mx = -INFINITY;
      if (t + 1 < ttiles) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[r] = sc[r]; mx = fmaxf(mx, sc[r]); }
      } else {                                      // only the last context tile can hold tokens >= T
        const int jbase = t * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jl = 8 * (r >> 2) + (r & 3);
          x[r] = (jbase + jl < T) ? sc[r] : -INFINITY;
          mx = fmaxf(mx, x[r]);
        }
      }
      mx = xhalf_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const bool moved = m_new > m_run;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      const float m_off = m_new - 10.0f;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_off); ls += x[r]; }
      l_half = fmaf(l_half, alpha, ls);
      if (__any(moved)) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 ph, pl;
        split8h(&x[8 * s2], ph, pl);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int slot = 2 * db + s2;
          mma_np<NP>(oacc[db], lv[(0 * 4 + slot) * 64], lv[(1 * 4 + slot) * 64], ph, pl);
        }
      }
    }
    FragH2<4> ox;
    {
      const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = oacc[db][r] * inv;
        ox.set_block(db, t);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const f16x8* lw = as_h2(ss.acquire());
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) mma_wx_h2n<4, NP>(x1a[2 * st + hb], lw + hb * (2 * 4 * 64), ox);
    }
  }

  // ---- feed-forward: x2 = x1 + W2 GEGLU(W1 LN(x1) + b1) + b2 --------------------------------------------------------
  FragH2<8> nx;
  f32x16 y[4];
  {
    float x1[CF], xn[CF];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) x1[16 * mb + r] = x1a[mb][r] * kH2Inv;
    layernorm_frag<CF>(xn, x1, lvec_f, lvec_f + C, h);
    nx.set(xn);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float b[16];
      load_vec_block(b, lvec_f + 2 * C + 2 * FFH, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) y[mb][r] = (x1[16 * mb + r] + b[r]) * 256.0f;
    }
  }
  __syncthreads();                               // every wave is done with the 2-slot ring: the 4-slot ring may overwrite it
  ff_chunks<NP>(nx, y, lds, ff_wst, lvec_f + 2 * C, lvec_f + 2 * C + FFH, wave, lane, h, 0, FFH / 32);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = y[mb][r] * kH2Inv;
    if (active) store_block_p32(x2_out + toff, mb, t, lane);
  }
}

// NP = 3: the parity kernel.  NP = 1: the throughput numerics mode (gmf_set_tuning "precision" = 1) - every product of the layer's
// linear part on the high fp16 planes only (a third of the MFMAs, half the LDS reads); LayerNorms, softmax, GELU, biases and
// residuals stay fp32.  NOT within the parity gate.
template <int NP, bool QSKIP = false>
__global__ void __launch_bounds__(256, 2)
k_linear_h2(const float* __restrict__ f_in, const float* __restrict__ front_wst, const float* __restrict__ front_vec,
            const float* __restrict__ ctx_img, const float* __restrict__ attn_wst, const float* __restrict__ attn_vec,
            const float* __restrict__ ff_wst, const float* __restrict__ ff_vec, float* __restrict__ q_out,
            float* __restrict__ k_out, float* __restrict__ v_out, float* __restrict__ x2_out, int N, int tiles, int T,
            int ttiles, const PairTab* __restrict__ ptab, unsigned* __restrict__ v_scale, const PvGuard guard) {
  __shared__ __attribute__((aligned(16))) float lds[kLinLdsFloats];
  tail_priority(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  linear_h2_body<0, NP, QSKIP>(lds, f_in, front_wst, front_vec, ctx_img, attn_wst, attn_vec, ff_wst, ff_vec, q_out, k_out, v_out, x2_out, N,
                               tiles, T, ttiles, ptab, v_scale, guard);
}

// grid (ceil(tiles / 4), B, 2): blockIdx.z = 0 the Q'/K/V role, 1 the Fusion-2 role of the same 128 rows
__global__ void __launch_bounds__(256, 2)
k_linear_roles(const float* __restrict__ f_in, const float* __restrict__ front_wst, const float* __restrict__ front_vec,
               const float* __restrict__ ctx_img, const float* __restrict__ attn_wst, const float* __restrict__ attn_vec,
               const float* __restrict__ ff_wst, const float* __restrict__ ff_vec, float* __restrict__ q_out,
               float* __restrict__ k_out, float* __restrict__ v_out, float* __restrict__ x2_out, int N, int tiles, int T,
               int ttiles, unsigned* __restrict__ v_scale, const PvGuard guard, const PairTab* __restrict__ ptab) {   // [r5] ragged batches too
  __shared__ __attribute__((aligned(16))) float lds[kLinLdsFloats];
  if (blockIdx.z == 0)
    linear_h2_body<1>(lds, f_in, front_wst, front_vec, ctx_img, attn_wst, attn_vec, ff_wst, ff_vec, q_out, k_out, v_out, x2_out, N,
                      tiles, T, ttiles, ptab, v_scale, guard);
  else
    linear_h2_body<2>(lds, f_in, front_wst, front_vec, ctx_img, attn_wst, attn_vec, ff_wst, ff_vec, q_out, k_out, v_out, x2_out, N,
                      tiles, T, ttiles, ptab);
}

// k_ff_reduce: x2 = sum_z part[z] + b2 + x1 (z in index order) for the hidden-split form of k_fusion_ff_h2p.
// grid (ceil(tiles/4), B, 4): a wave adds one 32 x 32 block; all hs partial blocks are requested before the first add.
__global__ void __launch_bounds__(256)
k_ff_reduce(const float* __restrict__ part, const float* __restrict__ x1, const float* __restrict__ vecs,
            float* __restrict__ x2_out, int tiles, int hs) {
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int tile = blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
  if (tile >= tiles) return;
  const int mb = blockIdx.z;
  const size_t n_all = (size_t)gridDim.y * tiles;
  const size_t toff = ((size_t)blockIdx.y * tiles + tile) * (32 * C);
  float b[16], xr[16], p[8][16];
  load_vec_block(b, vecs + 2 * C + 2 * FFH, mb, h);
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < hs) load_block_p32(p[z], part + (size_t)z * n_all * (32 * C) + toff, mb, lane);
  load_block_p32(xr, x1 + toff, mb, lane);
  float t[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = p[0][r];
#pragma unroll
  for (int z = 1; z < 8; ++z)
    if (z < hs) {
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] += p[z][r];
    }
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = fmaf(t[r], kH2Inv, b[r]) + xr[r];
  store_block_p32(x2_out + toff, mb, t, lane);
}

// -----------------------------------------------------------------------------------------
static inline dim3 tgrid(int tiles, int B, int sets = 1) { return dim3((tiles + kWavesPerWG - 1) / kWavesPerWG, B, sets); }

hipError_t launch_front_h2(const Tuning& tune, int mode, const float* in, const float* wst, const float* vecs, float* f,
                           float* q, float* k, float* v, int B, int N, int tiles, hipStream_t s, const PairTab* ptab, unsigned* v_scale,
                           PvGuard guard) {
  dim3 g = tgrid(tiles, B);
  if (g.x * B < 128 && tune.front_split) g.z = 3;      // small grids: one workgroup per output (Q' + f | K | V)
  if (mode == 3) { g.z = 1; hipLaunchKernelGGL(k_front_h2<3>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles, ptab, v_scale, guard); }
  else if (mode == 1) hipLaunchKernelGGL(k_front_h2<1>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles, ptab, v_scale, guard);
  else if (mode == 2) hipLaunchKernelGGL(k_front_h2<2>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles, ptab, v_scale, guard);
  else hipLaunchKernelGGL(k_front_h2<0>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles, ptab, v_scale, guard);
  return hipGetLastError();
}

hipError_t launch_linear_h2(const Tuning& tune, const float* f, const float* front_wst, const float* front_vec, const float* ctx_img,
                            const float* attn_wst, const float* attn_vec, const float* ff_wst, const float* ff_vec, float* q,
                            float* k, float* v, float* x2, int B, int N, int tiles, int T, int ttiles, hipStream_t s,
                            bool one_product, const PairTab* ptab, unsigned* v_scale, PvGuard guard) {
  // grids that give a CU about one workgroup: two roles per row block (the Q'/K/V projections | Fusion-2) in one launch
  const int W = ((tiles + 3) / 4) * B;
  if (tune.mid_grid_roles > 0 && W < tune.mid_grid_roles && q)
    hipLaunchKernelGGL(k_linear_roles, tgrid(tiles, B, 2), dim3(256), 0, s, f, front_wst, front_vec, ctx_img, attn_wst, attn_vec,
                       ff_wst, ff_vec, q, k, v, x2, N, tiles, T, ttiles, v_scale, guard, ptab);
  else if (one_product)                            // throughput numerics mode: high planes only
    hipLaunchKernelGGL(k_linear_h2<1>, tgrid(tiles, B), dim3(256), 0, s, f, front_wst, front_vec, ctx_img, attn_wst, attn_vec, ff_wst,
                       ff_vec, q, k, v, x2, N, tiles, T, ttiles, ptab, v_scale, guard);
  else if (!q)                                     // [r4] no Q' image: the attention kernel projects its own (CompatCache::qf_img)
    hipLaunchKernelGGL((k_linear_h2<3, true>), tgrid(tiles, B), dim3(256), 0, s, f, front_wst, front_vec, ctx_img, attn_wst, attn_vec, ff_wst,
                       ff_vec, q, k, v, x2, N, tiles, T, ttiles, ptab, v_scale, guard);
  else
    hipLaunchKernelGGL(k_linear_h2<3>, tgrid(tiles, B), dim3(256), 0, s, f, front_wst, front_vec, ctx_img, attn_wst, attn_vec, ff_wst,
                       ff_vec, q, k, v, x2, N, tiles, T, ttiles, ptab, v_scale, guard);
  return hipGetLastError();
}

hipError_t launch_ctx_prep_h2(bool pe, const float* ctx, const float* wst, const float* vecs, float* out, int B, int T,
                              int ttiles, int sets, int wst_stride, int vec_stride, hipStream_t s, bool rowmajor) {
  if (pe) hipLaunchKernelGGL(k_ctx_prep_h2<true>, tgrid(ttiles, B, sets), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles, wst_stride, vec_stride);
  else if (rowmajor) hipLaunchKernelGGL((k_ctx_prep_h2<false, true>), tgrid(ttiles, B, sets), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles, wst_stride, vec_stride);
  else hipLaunchKernelGGL(k_ctx_prep_h2<false>, tgrid(ttiles, B, sets), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles, wst_stride, vec_stride);
  return hipGetLastError();
}

hipError_t launch_fusion_attn_h2(bool pe, const float* x, const float* ctx_img, const float* wst, const float* vecs,
                                 float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s, bool rowmajor) {
  if (pe) hipLaunchKernelGGL(k_fusion_attn_h2<true>, tgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  else if (rowmajor) hipLaunchKernelGGL((k_fusion_attn_h2<false, true>), tgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  else hipLaunchKernelGGL(k_fusion_attn_h2<false>, tgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  return hipGetLastError();
}

// [r5] the three prologue launches of small grids (k_pro_*): image-side role | point-side role
hipError_t launch_pro_ctx_pts(const float* p_tokens, const float* wst, const float* vecs, float* f1ctx, int B, int T, int ttiles,
                              const float* src, const float* tgt, float* pts8, int N, hipStream_t s, const PairTab* ptab,
                              unsigned* zero_words, int n_zero) {
  const dim3 ga = tgrid(ttiles, B);
  const int Npad = ((N + 31) / 32) * 32;
  const long total = (long)B * Npad;
  const int n_a = (int)(ga.x * ga.y), n_b = (int)((std::max(total, (long)n_zero) + 255) / 256);
  hipLaunchKernelGGL(k_pro_ctx_pts, dim3(n_a + n_b), dim3(256), 0, s, n_a, (int)ga.x, B, p_tokens, wst, vecs, f1ctx, T, ttiles, src, tgt,
                     pts8, N, Npad, total, ptab, zero_words, n_zero);
  return hipGetLastError();
}

hipError_t launch_pro_fattn_compat(const float* q_tokens, const float* f1ctx, const float* wst, const float* vecs, float* x1t, int B,
                                   int T, int ttiles, const float* pts8, float* c_dense, int N, int tiles, float sigma_d,
                                   hipStream_t s, const PairTab* ptab) {
  const dim3 ga = tgrid(ttiles, B);
  const int gy = (tiles + 4 * kJPerWave - 1) / (4 * kJPerWave);
  const int n_a = (int)(ga.x * ga.y);
  const long n_b = (long)tiles * gy * B;
  hipLaunchKernelGGL(k_pro_fattn_compat, dim3((unsigned)(n_a + n_b)), dim3(256), 0, s, n_a, (int)ga.x, q_tokens, f1ctx, wst, vecs, x1t, T,
                     ttiles, pts8, c_dense, N, tiles, gy, 1.0f / (sigma_d * sigma_d), ptab);
  return hipGetLastError();
}

// returns through *hs_out the hidden splits of the feed-forward role (> 1: the caller runs launch_ff_reduce_h2 next)
hipError_t launch_pro_ff_front(const Tuning& tune, const float* x1t, const float* wst, const float* vecs, float* imgfeat, int B,
                               int ttiles, float* part, int max_parts, const float* corr_pos, const float* fwst, const float* fvecs,
                               float* f, float* q, float* k, float* v, int N, int tiles, hipStream_t s, const PairTab* ptab,
                               PvGuard guard, int* hs_out) {
  const dim3 ga = tgrid(ttiles, B), gf = tgrid(tiles, B);
  const int hs = part ? plan_ff_split(tune, ga.x * B, max_parts) : 1;
  const int n_a = (int)(ga.x * B) * hs, n_b = (int)(gf.x * B);
#define GMF_PRO_C(SPLIT, MINB, HS, PART)                                                                                              \
  hipLaunchKernelGGL((k_pro_ff_front<SPLIT, MINB>), dim3(n_a + n_b), dim3(256), 0, s, n_a, (int)ga.x, B, HS, x1t, wst, vecs, imgfeat, ttiles, \
                     PART, (int)gf.x, corr_pos, fwst, fvecs, f, q, k, v, N, tiles, ptab, guard)
  const bool one_per_cu = n_a + n_b <= 256;
  if (hs > 1) { if (one_per_cu) GMF_PRO_C(true, 1, hs, part); else GMF_PRO_C(true, 2, hs, part); }
  else { if (one_per_cu) GMF_PRO_C(false, 1, 1, (float*)nullptr); else GMF_PRO_C(false, 2, 1, (float*)nullptr); }
#undef GMF_PRO_C
  *hs_out = hs;
  return hipGetLastError();
}

hipError_t launch_ff_reduce_h2(const float* part, const float* x1, const float* vecs, float* x2, int B, int tiles, int hs, hipStream_t s) {
  const dim3 g = tgrid(tiles, B);
  hipLaunchKernelGGL(k_ff_reduce, dim3(g.x, g.y, 4), dim3(256), 0, s, part, x1, vecs, x2, tiles, hs);
  return hipGetLastError();
}

// hidden splits of the feed-forward on a grid of `base` workgroups (tune.ff_split: 0 = automatic, 1 = off, 2 / 4 / 8 = forced)
int plan_ff_split(const Tuning& tune, int base, int max_parts) {
  int hs = 1;
  if (max_parts >= 2) {
    if (tune.ff_split > 0) hs = tune.ff_split;
    else if (base < 256) hs = base <= 32 ? 8 : base <= 64 ? 4 : 2;   // (measured at B = 1 .. 4: the merge reads every partial)
    hs = std::min(hs, max_parts);
    if (hs != 2 && hs != 4 && hs != 8) hs = 1;
  }
  return hs;
}

hipError_t launch_small_front_fattn(const float* f, const float* front_wst, const float* front_vec, const float* ctx_img,
                                    const float* attn_wst, const float* attn_vec, float* q, float* k, float* v, float* x1, int B,
                                    int N, int tiles, int T, int ttiles, hipStream_t s, unsigned* v_scale, PvGuard guard, bool tile_role,
                                    const PairTab* ptab) {
  // [r5] the cross-attention role per query tile (its waves split the context tiles) wherever that leaves the chip room: the role's
  // workgroups quadruple, so up to 256 query tiles; beyond, one workgroup per four tiles as before
  const bool per_tile = tile_role && ttiles >= 2 && (long)tiles * B <= 256;
  hipLaunchKernelGGL(k_small_front_fattn, tgrid(tiles, B, per_tile ? 7 : 4), dim3(256), 0, s, f, front_wst, front_vec, ctx_img, attn_wst, attn_vec,
                     q, k, v, x1, N, tiles, T, ttiles, v_scale, guard, ptab);
  return hipGetLastError();
}

hipError_t launch_fusion_ff_h2(const Tuning& tune, const float* x1, const float* wst, const float* vecs, float* x2, int B,
                               int tiles, hipStream_t s, float* part, int max_parts) {
  const dim3 g = tgrid(tiles, B);
  // small grids: divide the 16 hidden chunks over 2 / 4 / 8 workgroups (deterministic two-pass sum)
  const int hs = part ? plan_ff_split(tune, g.x * B, max_parts) : 1;
  if (hs > 1) {
    hipLaunchKernelGGL(k_fusion_ff_h2p<true>, dim3(g.x, g.y, hs), dim3(256), 0, s, x1, wst, vecs, x2, tiles, part);
    hipLaunchKernelGGL(k_ff_reduce, dim3(g.x, g.y, 4), dim3(256), 0, s, part, x1, vecs, x2, tiles, hs);
  }
  else hipLaunchKernelGGL(k_fusion_ff_h2p<false>, g, dim3(256), 0, s, x1, wst, vecs, x2, tiles, (float*)nullptr);
  return hipGetLastError();
}

}  // namespace gmf
