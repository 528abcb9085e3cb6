// GMF correspondence encoder on gfx950: the fp32-MFMA kernels of every stage and the spatial-consistency attention
// kernels (the split-fp16 linear stages live in encoder_h2.hip).
//
// Replaces (file:line relative to /root/reference/GMF_PointDSC/):
//   k_front      models/PointDSC.py:88,104-109,56-58   layer0 / PointCN conv1x1+BN+ReLU, Q/K/V conv1x1
//   k_scattn*    models/PointDSC.py:216-221,60-65,73   compat matrix + softmax(compat*QK^T/sqrt(C)) V + fc_message + block sum
//   k_compat_build models/PointDSC.py:216-221          the compat matrix once per batch, in the attention kernel's register order
//   k_ctx_prep   models/fusion_layer.py:124-126,46-49,86-87   LCPE(content) + LayerNorm + to_kv
//   k_fusion_attn models/fusion_layer.py:119-121,44,84-94,190 LCPE(q) + LayerNorm + to_q + softmax(QK^T) V + to_out + residual
//   k_fusion_ff  models/fusion_layer.py:54-69,191      LayerNorm + Linear + GEGLU + Linear + residual
//   k_head       models/PointDSC.py:175-181,229,241    classifier head + F.normalize
//
// Forms of the attention kernel (launch_scattn, `scattn_variant`):
//   k_scattn_h2p  (18, DEFAULT) split-fp16 MFMA operands, c streamed from the compat cache, tile loop software-pipelined
//                 inside each wave, split-fp16 fc_message epilogue; KSPLIT form + k_scattn_merge for small grids
//   k_scattn_h2   (9)  split-fp16, c recomputed in-kernel or cached, not pipelined: fallback when the cache does not fit
//   k_scattn      (0)  fp32 MFMA on fp32 images: the single-stage entry point gmf_scattn_forward and, as k_scattn<true>,
//                 the drop-in NonLocalBlock with a caller-supplied dense `attention` [B,N,N]
// The measured-and-rejected forms of round 1 (split-bf16, 8-wave, rational compat, 16x16x32 MFMA, timing ablations) are
// kept as source only in tools/ubench/archive/ and are not part of the library.
//
// Every kernel runs 4 waves per workgroup, each wave owning one 32-row tile ("rows on lanes",
// see mfma_core.hpp).  Weights, K/V tiles and context tiles stream L2 -> LDS in 16 KiB stages by
// LDS-DMA; activations chain from one MFMA's accumulators into the next MFMA's operand registers.
#include <algorithm>
#include <type_traits>
#include "enc_common.hpp"
#include "enc_ff.hpp"

namespace gmf {

// =========================================================================================
// k_front:  [layer0] -> PointCN (BN folded) -> f ; Q' = (Wq f + bq) * log2(e)/sqrt(C) ; K ; V
//   stages (16): Wp[4] | Wq'[4] | Wk[4] | Wv[4]      vecs: bp | bq' | bk | bv | b0 | W0 image (K=8)
//   outputs: f, Q', K as P32 images; V as T image.
// =========================================================================================
// MODE 0: in = feat image, PointCN applied.  MODE 1: in = corr_pos, layer0 then PointCN.
// MODE 2: in = feat image used as-is (stand-alone NonLocalBlock whose caller already applied PointCN).
template <int MODE>
__global__ void __launch_bounds__(256, 2)
k_front(const float* __restrict__ in, const float* __restrict__ wst, const float* __restrict__ vecs,
        float* __restrict__ f_out, float* __restrict__ q_out, float* __restrict__ k_out,
        float* __restrict__ v_out, int N, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);

  constexpr bool FIRST = (MODE == 1);
  StageStream ss;
  if (MODE == 2) ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst + 4 * kStageFloats, 12);
  else ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst, 16);
  ss.prime();

  float x[CF];
  if (FIRST) {
    const int row = tile * 32 + i;
    float pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * h + e;
      pk[e] = (row < N && k < 6) ? in[((size_t)pair * N + row) * 6 + k] : 0.f;
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

  float f[CF];
  if (MODE == 2) {
#pragma unroll
    for (int e = 0; e < CF; ++e) f[e] = x[e];
  } else {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const float4* lw = ss.acquire();
      f32x16 acc = zero16();
      mma_wx<CF>(acc, lw, x);
      float b[16];
      load_vec_block(b, vecs, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) f[16 * mb + r] = relu_nan(acc[r] + b[r]);
    }
  }
  if (active) store_frag_p32<CF>(f_out + toff, f, lane);

#pragma unroll
  for (int which = 0; which < 2; ++which) {   // Q', K
    float* dst = (which == 0 ? q_out : k_out) + toff;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const float4* lw = ss.acquire();
      f32x16 acc = zero16();
      mma_wx<CF>(acc, lw, f);
      float b[16], t[16];
      load_vec_block(b, vecs + (1 + which) * C, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] + b[r];
      if (active) store_block_p32(dst, mb, t, lane);
    }
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {             // V (feature on lane)
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_xw<CF>(acc, lw, f);
    const float bv = vecs[3 * C + 32 * db + i];
    if (active) store_block_timg(v_out + toff, db, acc, bv, lane);
  }
}

// =========================================================================================
// k_scattn: spatial-consistency self-attention, flash style, compat recomputed per (i,j).
//   S^T = K Q'^T (keys on registers, queries on lanes) ; x = c_ij * s ; online softmax in base 2 ;
//   O^T += V^T P^T.  Epilogue: fc_message (BN folded) and  out = msg + fusion2_out.
//   stages (5): Wa[2] (64x128) | Wb (64x64, 2 blocks) | Wc[2] (128x64, 2 blocks each)
//   vecs: ba[64] | bb[64] | bc[128]
//   pts8: [B, Npad, 8] = (sx,sy,sz,0,tx,ty,tz,0)
// =========================================================================================
constexpr int kAttnBufFloats = 2 * kStageFloats + 256;   // K tile | V tile | 32 x pts8

// DENSE = true: c_ij is read from the caller's [B,N,N] matrix `pts8` points to (drop-in NonLocalBlock).
template <bool DENSE>
__global__ void __launch_bounds__(256, 2)
k_scattn(const float* __restrict__ q_img, const float* __restrict__ k_img, const float* __restrict__ v_img,
         const float* __restrict__ pts8, const float* __restrict__ fus, const float* __restrict__ wst,
         const float* __restrict__ vecs, float* __restrict__ out, int N, int tiles, float inv_sig2) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kAttnBufFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t pbase = (size_t)pair * tiles;
  const size_t toff = (pbase + tile) * (32 * C);

  float qf[CF];
  load_frag_p32<CF>(qf, q_img + toff, lane);
  float si[3] = {0.f, 0.f, 0.f}, ti[3] = {0.f, 0.f, 0.f};
  const float* crow = nullptr;   // DENSE: row i of the pair's compat matrix
  if (DENSE) {
    const int row = min(tile * 32 + i, N - 1);
    crow = pts8 + ((size_t)pair * N + row) * (size_t)N;
  } else {
    const float4* pp = reinterpret_cast<const float4*>(pts8 + (pbase * 32 + (size_t)tile * 32 + i) * 8);
    const float4 a = pp[0], b = pp[1];
    si[0] = a.x; si[1] = a.y; si[2] = a.z; ti[0] = b.x; ti[1] = b.y; ti[2] = b.z;
  }

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float m_run = -INFINITY, l_half = 0.f;

  const float* gk = k_img + pbase * (32 * C);
  const float* gv = v_img + pbase * (32 * C);
  const float* gp = pts8 + pbase * 32 * 8;

  auto issue_tile = [&](int t) {
    float* buf = lds + (t & 1) * kAttnBufFloats;
    dma_issue(gk + (size_t)t * kStageFloats, buf, 16, wave, kWavesPerWG, lane);
    dma_issue(gv + (size_t)t * kStageFloats, buf + kStageFloats, 16, wave, kWavesPerWG, lane);
    if (!DENSE && wave == (t & 3)) dma_piece_1k(gp + (size_t)t * 256, buf + 2 * kStageFloats, lane);
  };

  issue_tile(0);
  for (int t = 0; t < tiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < tiles) issue_tile(t + 1);
    const float* buf = lds + (t & 1) * kAttnBufFloats;
    const float4* lk = reinterpret_cast<const float4*>(buf) + lane;
    const float4* lv = reinterpret_cast<const float4*>(buf + kStageFloats) + lane;
    const float4* lp = reinterpret_cast<const float4*>(buf + 2 * kStageFloats) + 8 * h;

    f32x16 s = zero16();
    mma_wx<CF>(s, lk, qf);

    float x[16];
    const int jbase = t * 32 + 4 * h;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = 8 * (r >> 2) + (r & 3);           // + 4h folded into lp / jbase
      float c;
      if (DENSE) {
        c = crow[min(jbase + jl, N - 1)];
      } else {
        const float4 a = lp[2 * jl], b = lp[2 * jl + 1];
        const float ax = si[0] - a.x, ay = si[1] - a.y, az = si[2] - a.z;
        const float bx = ti[0] - b.x, by = ti[1] - b.y, bz = ti[2] - b.z;
        const float ds = sqrtf(fmaf(az, az, fmaf(ay, ay, ax * ax)));
        const float dt = sqrtf(fmaf(bz, bz, fmaf(by, by, bx * bx)));
        const float d = ds - dt;
        c = fmaxf(1.0f - d * d * inv_sig2, 0.f);
      }
      float v = c * s[r];
      v = (jbase + jl < N) ? v : -INFINITY;
      x[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_new); ls += x[r]; }
    l_half = fmaf(l_half, alpha, ls);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 vv = lv[(db * 4 + q) * 64];
        oacc[db] = mfma32(vv.x, x[4 * q + 0], oacc[db]);
        oacc[db] = mfma32(vv.y, x[4 * q + 1], oacc[db]);
        oacc[db] = mfma32(vv.z, x[4 * q + 2], oacc[db]);
        oacc[db] = mfma32(vv.w, x[4 * q + 3], oacc[db]);
      }
    }
  }

  // ---- epilogue: normalise, fc_message, add the Fusion-2 branch -------------------------
  float o[CF];
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[16 * db + r] = oacc[db][r] * inv;
  }
  __syncthreads();   // all waves are done with the K/V buffers before weights stream into them
  StageStream ss;
  ss.init(lds, lds + kAttnBufFloats, wave, kWavesPerWG, lane, wst, 5);
  ss.prime();
  float m1[DHF], m2[DHF];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<CF>(acc, lw, o);
    float b[16];
    load_vec_block(b, vecs, mb, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) m1[16 * mb + r] = relu_nan(acc[r] + b[r]);
  }
  {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x16 acc = zero16();
      mma_wx<DHF>(acc, lw + mb * (32 * DH / 4), m1);
      float b[16];
      load_vec_block(b, vecs + 64, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) m2[16 * mb + r] = relu_nan(acc[r] + b[r]);
    }
  }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int mb = 2 * st + hb;
      f32x16 acc = zero16();
      mma_wx<DHF>(acc, lw + hb * (32 * DH / 4), m2);
      float b[16], fz[16], t[16];
      load_vec_block(b, vecs + 128, mb, h);
      load_block_p32(fz, fus + toff, mb, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] + b[r] + fz[r];
      if (active) store_block_p32(out + toff, mb, t, lane);
    }
  }
}

// (len_diff / compat_times: enc_common.hpp - shared with the prologue role kernels of encoder_h2.hip)

// =========================================================================================
// k_scattn_h2: the attention with split-fp16 operands (two planes, three partial products, 16 KiB tiles), c_ij
// recomputed per (i, j) from the key points (cache-less fallback, `scattn_variant` 9) or streamed from the compat cache.
// P is scaled by 2^10 (folded into the exponent; the row sum carries the same factor, so O/l is unchanged) to keep
// its low plane out of the fp16 subnormal range.
// K | V | pts8 double buffered (66 KB per workgroup, two workgroups per CU: the two waves that share a SIMD belong to
// different workgroups, are not barrier-locked to each other, and one's MFMA phase runs under the other's VALU phase),
// ONE barrier per tile.
// CACHED: c_ij does not depend on the layer, so k_compat_build evaluates it ONCE per batch and stores it in the element
// order of this kernel (4 KiB per 32x32 tile: [q][lane][4 floats], registers r = 4q .. 4q+3 of each lane).  The 12
// attention launches then stream it back with four coalesced, non-temporal 16-byte loads per lane and tile (prefetched
// one tile ahead) instead of re-evaluating 2 square roots and ~20 more VALU instructions per element: 100 MB per pair
// and layer at N = 5000 - HBM bandwidth this otherwise compute-bound kernel was not using.
template <bool CACHED>
__global__ void __launch_bounds__(256, 2)
k_scattn_h2(const float* __restrict__ q_img, const float* __restrict__ k_img, const float* __restrict__ v_img,
            const float* __restrict__ pts8, const float* __restrict__ fus, const float* __restrict__ wst,
            const float* __restrict__ vecs, float* __restrict__ out, int N, int tiles, float inv_sig2, int wgs_per_pair,
            const float* __restrict__ c_dense) {
  constexpr int WAVES = 4;
  constexpr int kBuf = 2 * kStageFloats + 256;
  __shared__ __attribute__((aligned(16))) float lds[2 * kBuf];
  float* const ldsK = lds;
  float* const ldsV = lds + kStageFloats;
  float* const ldsP = lds + 2 * kStageFloats;
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware work mapping: hardware deals workgroup L to XCD L % 8, so give XCD x one contiguous run of
  // (pair, query-block) items.  All workgroups of a pair then stream the SAME K/V tiles through ONE L2 at
  // about the same time (each tile is fetched from HBM/MALL once per pair instead of once per workgroup).
  // Pure speed choice: any placement computes the same result.
  int pair, qblock;
  {
    const int total = gridDim.x, L = blockIdx.x;
    const int chunk = total >> 3, rem = total & 7, xcd = L & 7, kth = L >> 3;
    const int start = (xcd < rem) ? xcd * (chunk + 1) : rem * (chunk + 1) + (xcd - rem) * chunk;
    const int logical = start + kth;
    pair = logical / wgs_per_pair;
    qblock = logical - pair * wgs_per_pair;
  }
  const int tile_raw = qblock * WAVES + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t pbase = (size_t)pair * tiles;
  const size_t toff = (pbase + tile) * (32 * C);

  f16x8 qh[8], ql[8];
  {
    const f16x8* qp = reinterpret_cast<const f16x8*>(q_img + (pbase + tile) * (size_t)kStageFloats) + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) { qh[s] = qp[(0 * 8 + s) * 64]; ql[s] = qp[(1 * 8 + s) * 64]; }
  }
  float si[3], ti[3];
  {
    const float4* pp = reinterpret_cast<const float4*>(pts8 + (pbase * 32 + (size_t)tile * 32 + i) * 8);
    const float4 a = pp[0], b = pp[1];
    si[0] = a.x; si[1] = a.y; si[2] = a.z; ti[0] = b.x; ti[1] = b.y; ti[2] = b.z;
  }
  const f32x4* crow = nullptr;
  float c_a[16], c_b[16];
  if (CACHED) crow = reinterpret_cast<const f32x4*>(c_dense) + ((pbase + tile) * (size_t)tiles) * 256 + lane;
  auto fetch_c = [&](int t, float (&c_nxt)[16]) {   // cached c of key tile t for this wave's 32 queries
    const f32x4* ct = crow + (size_t)t * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = __builtin_nontemporal_load(ct + q * 64);
      c_nxt[4 * q + 0] = v[0]; c_nxt[4 * q + 1] = v[1]; c_nxt[4 * q + 2] = v[2]; c_nxt[4 * q + 3] = v[3];
    }
  };
  const float* gk = k_img + pbase * (size_t)kStageFloats;
  const float* gv = v_img + pbase * (size_t)kStageFloats;
  const float* gp = pts8 + pbase * 32 * 8;
  auto issueK = [&](int t) { dma_issue(gk + (size_t)t * kStageFloats, ldsK + (t & 1) * kBuf, 16, wave, WAVES, lane); };
  auto issueV = [&](int t) {
    dma_issue(gv + (size_t)t * kStageFloats, ldsV + (t & 1) * kBuf, 16, wave, WAVES, lane);
    if (!CACHED && wave == (t & (WAVES - 1))) dma_piece_1k(gp + (size_t)t * 256, ldsP + (t & 1) * kBuf, lane);
  };

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  const f16x8* lk0 = reinterpret_cast<const f16x8*>(ldsK) + lane;
  const f16x8* lv0 = reinterpret_cast<const f16x8*>(ldsV) + lane;
  const float4* lp0 = reinterpret_cast<const float4*>(ldsP) + 8 * h;

  issueK(0);
  issueV(0);
  if (CACHED) fetch_c(0, c_a);
  auto tile_step = [&](const int t, float (&c_cur)[16], float (&c_nxt)[16]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile t (issued one whole tile ago) has landed
    __syncthreads();                                     // ... for every wave; buffer (t+1)&1 is free again
    if (t + 1 < tiles) { issueK(t + 1); issueV(t + 1); }
    if (CACHED && t + 1 < tiles) fetch_c(t + 1, c_nxt);
    const int boff = (t & 1) * (kBuf / 4);               // in 16-byte units
    const f16x8* lk = lk0 + boff;
    const f16x8* lv = lv0 + boff;
    const float4* lp = lp0 + boff;
    // ---- S^T = K_t Q'^T : 8 k-steps x 3 partial products ----
    f32x16 sacc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f16x8 kh = lk[(0 * 8 + s) * 64], kl = lk[(1 * 8 + s) * 64];
      mma3(sacc, kh, kl, qh[s], ql[s]);
    }
    // ---- compat, scores, online softmax ----
    float x[16];
    float mx = -INFINITY;
    if (t + 1 < tiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = 8 * (r >> 2) + (r & 3);
        x[r] = CACHED ? c_cur[r] * sacc[r] : compat_times(lp, jl, si, ti, inv_sig2, sacc[r]);
        mx = fmaxf(mx, x[r]);
      }
    } else {
      const int jbase = t * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = 8 * (r >> 2) + (r & 3);
        const float v = CACHED ? c_cur[r] * sacc[r] : compat_times(lp, jl, si, ti, inv_sig2, sacc[r]);
        x[r] = (jbase + jl < N) ? v : -INFINITY;
        mx = fmaxf(mx, x[r]);
      }
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const bool moved = m_new > m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
    const float m_off = m_new - 10.0f;   // P' = 2^10 P
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_off); ls += x[r]; }
    l_half = fmaf(l_half, alpha, ls);
    if (__any(moved)) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    // ---- O^T += V_t^T P^T : P planes straight from the accumulator registers ----
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int slot = 2 * db + s2;
        const f16x8 vh = lv[(0 * 8 + slot) * 64], vl = lv[(1 * 8 + slot) * 64];
        mma3(oacc[db], vh, vl, ph, pl);
      }
    }
  };
  // two register sets for the cached c tile: tile t multiplies out of one while tile t+1 streams into the other
  for (int t = 0; t < tiles; t += 2) {
    tile_step(t, c_a, c_b);
    if (t + 1 < tiles) tile_step(t + 1, c_b, c_a);
  }

  // ---- epilogue: normalise, fc_message (fp32 MFMA), add the Fusion-2 branch ----
  float o[CF];
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[16 * db + r] = oacc[db][r] * inv;
  }
  __syncthreads();
  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, WAVES, lane, wst, 5);
  ss.prime();
  float m1[DHF], m2[DHF];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<CF>(acc, lw, o);
    float b[16];
    load_vec_block(b, vecs, mb, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) m1[16 * mb + r] = relu_nan(acc[r] + b[r]);
  }
  {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x16 acc = zero16();
      mma_wx<DHF>(acc, lw + mb * (32 * DH / 4), m1);
      float b[16];
      load_vec_block(b, vecs + 64, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) m2[16 * mb + r] = relu_nan(acc[r] + b[r]);
    }
  }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int mb = 2 * st + hb;
      f32x16 acc = zero16();
      mma_wx<DHF>(acc, lw + hb * (32 * DH / 4), m2);
      float b[16], fz[16], t[16];
      load_vec_block(b, vecs + 128, mb, h);
      load_block_p32(fz, fus + toff, mb, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] + b[r] + fz[r];
      if (active) store_block_p32(out + toff, mb, t, lane);
    }
  }
}

// =========================================================================================
// k_compat_build: the spatial-consistency matrix c_ij = max(0, 1 - (||s_i-s_j|| - ||t_i-t_j||)^2 / sigma_d^2)
// (PointDSC.py:216-221), evaluated once per batch with the reference's roundings and stored in the element order of the
// attention kernels: for query tile I and key tile J, lane (h, i) register r is the pair
// (32I + i, 32J + 8(r>>2) + 4h + (r&3)); tile (I, J) is the 4 KiB block [q][lane][4] with r = 4q + e.
// grid (tiles, ceil(tiles / (4 * kJPerWave)), B), block 256
// =========================================================================================
// (compat_build_body, kJPerWave: enc_common.hpp)
template <int FMT>
__global__ void __launch_bounds__(256)
k_compat_build(const float* __restrict__ pts8, float* __restrict__ c_dense, int N, int tiles, float inv_sig2,
               const PairTab* __restrict__ ptab) {
  __shared__ float tr[kCompatLdsFloats];
  compat_build_body<FMT>(tr, blockIdx.x, blockIdx.y, blockIdx.z, pts8, c_dense, N, tiles, inv_sig2, ptab);
}

// =========================================================================================
// k_scattn_h2p: the cached-c split-fp16 attention with the tile loop software-pipelined inside each wave.
//   With c_ij streamed from the cache a tile costs 48 MFMAs and ~130 vector instructions per wave - few enough to sit
//   in the MFMA issue gaps (an MFMA holds the vector issue port for 8 of its 32 cycles).  K tiles run one tile ahead of
//   V in the LDS rings: phase 1 issues the 24 MFMAs of S_{t+1} = K_{t+1} Q'^T, one per sched_barrier(0) unit, each
//   unit carrying its share of tile t's scores (c * s, running max), the exponentials of the first 8 keys and their
//   fp16 split; phase 2 issues the 24 MFMAs of O^T += V_t^T P^T with the other 8 exponentials underneath.  The row
//   max crosses the two K-halves with v_permlane32_swap (no LDS round trip).  Same accumulation order as
//   k_scattn_h2<.., CACHED>, so the results are bit-identical to it.
//   LDS: K ring [2][16 KiB] | V ring [2][16 KiB]; one barrier per tile; two workgroups per CU.
// =========================================================================================
GMF_DEVINL f32x16 mma3_part(int u, f32x16 acc, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
  switch (u) {
    case 0: return mfma_h16(al, bh, acc);
    case 1: return mfma_h16(ah, bl, acc);
    default: return mfma_h16(ah, bh, acc);
  }
}


// A tile's 8 LDS-DMA pieces are issued one per unit in the bare MFMA gaps of phase 2 (at the tile top / in phase 1
// measured 1.19 / 1.17 ms against 1.16 ms per launch in round 1).
// fc_message on the f16 MFMA (split-fp16 weights, tail_wst_h2) + bias + Fusion-2 branch for the 32 rows of one wave, given
// the normalised attention output o (fragment order).  `ss` is primed on the 5 weight stages; every wave of the workgroup
// calls this (padding waves with active = false keep their seat at the stage barriers).
// NEXT_PCN: the block output feat = msg + fusion2_out is not stored; the NEXT layer's PointCN (conv1x1 + folded BatchNorm +
// ReLU, PointDSC.py:104-109,141) runs on it here - 4 more weight stages (front_wst_h2 of layer l+1, blocks Wp) - and
// f_{l+1} = ReLU(Wp' feat + bp') is what goes to memory: the next layer's linear kernel needs f with its two LCPE neighbour
// rows, and the raw feat of a middle layer has no other reader.
// The Fusion-2 branch x2 of the block sum, per 32-feature block of this wave's tile: read from the x2 image (FusTile), or -
// small grids - formed here from the hidden-split feed-forward partials, x2 = (sum_z part[z]) / 256 + b2 + x1 with the
// partials added in index order (k_scattn_merge, FusRegs: bit-identical to k_ff_reduce, which this replaces).
struct FusTile {
  const float* tile;
  GMF_DEVINL void load(float (&fz)[16], int mb, int lane, int h) const { load_block_p32(fz, tile, mb, lane); }
};

// the Fusion-2 branch already summed into registers (k_scattn_merge: every partial is requested before the first stage is used)
struct FusRegs {
  const float (&f)[CF];
  GMF_DEVINL void load(float (&fz)[16], int mb, int lane, int h) const {
#pragma unroll
    for (int r = 0; r < 16; ++r) fz[r] = f[16 * mb + r];
  }
};

// The epilogue's per-feature vectors in the LDS: ba[64] | bb[64] | bc[128] (the fc_message biases) | the next layer's PointCN
// bias [128].  Issued before the epilogue's weight stages; their first acquire (wait + barrier) makes them visible.
constexpr int kTailVecFloats = 3 * C;
GMF_DEVINL void stage_tail_vecs(float* lvec, const float* __restrict__ vecs, const float* __restrict__ next_bias, int wave, int lane) {
  if (wave == 0) dma_piece_1k(vecs, lvec, lane);                                   // 256 floats
  if (wave == 1 && next_bias && lane < 32) dma_piece_1k(next_bias, lvec + 2 * C, lane);   // 128 floats
}

template <bool NEXT_PCN, class Stages, class Fus>
GMF_DEVINL void scattn_epilogue_h2(const float (&o)[CF], const bool active, Stages& ss, const float* __restrict__ vecs,
                                   const Fus& fus_tile, float* __restrict__ out_tile, const int lane, const int h,
                                   const float* __restrict__ next_bias = nullptr,
                                   // [r5] NEXT_PCN: non-null = raise the next layer's "pv_fp8" statistic of `pair` to these rows' |f_{l+1}|^2
                                   unsigned* __restrict__ stat_next = nullptr, const int pair = 0, const bool row_valid = false) {
    // stages (fp16x2 images, same sizes as the fp32 ones): Wa block 0 | Wa block 1 | Wb (2 blocks) | Wc 0,1 | Wc 2,3
    FragH2<8> ox, featx;
    FragH2<4> m1x, m2x;
    if (active) ox.set(o);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const f16x8* lw = as_h2(ss.acquire());
      if (!active) continue;
      f32x16 acc = zero16();
      mma_wx_h2<8>(acc, lw, ox);
      float b[16], t1[16];
      load_vec_block(b, vecs, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) t1[r] = relu_nan(fmaf(acc[r], kH2Inv, b[r]));
      m1x.set_block(mb, t1);
    }
    {
      const f16x8* lw = as_h2(ss.acquire());
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if (!active) continue;
        f32x16 acc = zero16();
        mma_wx_h2<4>(acc, lw + mb * (2 * 4 * 64), m1x);
        float b[16], t2[16];
        load_vec_block(b, vecs + 64, mb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[r] = relu_nan(fmaf(acc[r], kH2Inv, b[r]));
        m2x.set_block(mb, t2);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const f16x8* lw = as_h2(ss.acquire());
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        if (!active) continue;
        const int mb = 2 * st + hb;
        f32x16 acc = zero16();
        mma_wx_h2<4>(acc, lw + hb * (2 * 4 * 64), m2x);
        float b[16], fz[16], tt[16];
        load_vec_block(b, vecs + 128, mb, h);
        fus_tile.load(fz, mb, lane, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) tt[r] = fmaf(acc[r], kH2Inv, b[r]) + fz[r];
        if (NEXT_PCN) featx.set_block(mb, tt);
        else if (active) store_block_p32(out_tile, mb, tt, lane);
      }
    }
    if (NEXT_PCN) {
      float ssq = 0.f;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const f16x8* lw = as_h2(ss.acquire());
        if (!active) continue;
        f32x16 acc = zero16();
        mma_wx_h2<8>(acc, lw, featx);
        float b[16], tt[16];
        load_vec_block(b, next_bias, mb, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) { tt[r] = relu_nan(fmaf(acc[r], kH2Inv, b[r])); ssq = fmaf(tt[r], tt[r], ssq); }
        store_block_p32(out_tile, mb, tt, lane);
      }
      if (stat_next && active) pv_stat_raise(stat_next, pair, ssq, row_valid, lane);     // (uniform per wave)
    }
}

// Work mapping of the attention grid (k_scattn_h2p and k_scattn_merge).  The n_items = B * wgs_per_pair work items
// (pair, block of 4 query tiles) are dealt to the 8 XCDs in contiguous runs (hardware sends workgroup L to XCD L % 8), so
// that all workgroups of a pair stream the SAME K/V tiles through ONE L2 at about the same time.  Within an XCD's run the
// first `n_full` items are WHOLE workgroups (all key tiles, epilogue in-kernel); each remaining item is SPLIT by key range
// over `ksplits` workgroups that write un-normalised partial results for k_scattn_merge.  Two uses:
//   * small grids (fewer items than the chip has workgroup slots, e.g. B = 1 - the reference's evaluation mode): n_full = 0;
//   * the tail of a large grid: 32 pairs x 40 query blocks = 1280 items on 512 slots are 2.5 rounds, i.e. 3 rounds of
//     wall time with half the chip idle in the last; with n_full = 2 rounds' worth and the last 256 items split in two,
//     the tail is 512 half-length workgroups = one half round.
struct AttnItem { int item, ks; bool split, valid; };
GMF_DEVINL AttnItem attn_item(int L, int n_items, int n_full, int ksplits) {
  const int chunk = n_items >> 3, rem = n_items & 7, xcd = L & 7, kth = L >> 3;
  const int start = (xcd < rem) ? xcd * (chunk + 1) : rem * (chunk + 1) + (xcd - rem) * chunk;
  const int items = chunk + (xcd < rem ? 1 : 0);
  const int nf = min(n_full, items);
  if (kth < nf) return AttnItem{start + kth, 0, false, true};
  const int j = kth - nf, r = j / ksplits;
  return AttnItem{start + nf + r, j - r * ksplits, true, nf + r < items};
}

// `bid`: the workgroup's index in the attention grid; `lds`: 64 KiB owned by the workgroup.
// NPROD = 3, CFMT = 0: the parity form (split-fp16 operands, three partial products, c as fp32); CFMT = 2: the same with c
// streamed as 16-bit fixed point (k_compat_build<2>; gmf_set_tuning "compat_format" = 2, opt-in).
// NPROD = 1, CFMT = 1: the THROUGHPUT numerics mode (gmf_set_tuning "precision" = 1): only the high
// fp16 planes of Q', K, V and of the probabilities are multiplied (one product, fp32 accumulation; the softmax statistics,
// the compat product and the epilogue stay as they are) and c is streamed as fp16 - NOT within the 1e-4 parity gate.
constexpr float kInvU16 = 1.0f / 65535.0f;
// (WAVES = 8 - one 256-query workgroup per CU, every K / V tile fetched once per 256 queries, half the LDS-DMA pieces per wave - was
// built and measured in round 3: bit-identical results, 19.2 instead of 18.4 ms per step; eight waves behind each tile barrier
// cost more than the halved stream saves.  The parameter stays, the instantiation is gone.)
// PVF8 (with NPROD = 3): the two CROSS products of O += P V (P_hi V_lo + P_lo V_hi) on the block-scaled fp8 matrix pipe - ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per feature block and tile instead of four f16 MFMAs; P_hi V_hi and all of S = Q' K^T stay
// three-product split-fp16.  The V image then carries e4m3 planes with one scale per (feature, tile) in place of its low fp16
// plane (store_block_v8, enc_common.hpp; scales in `v_scale`), the probabilities are p ~ ph + pl8 with ph = fp16(p) and
// pl8 = e4m3 of the residual, and the row sum is taken over exactly those values - numerator and denominator of the softmax see the
// same probabilities, so what is left of the e4m3 rounding is (v_j - o) weighted: 1.5e-6 ... 2.5e-6 on the logits in the fp64
// emulation of the whole encoder (tests/tools/mx_cross_emulation.py; 2e-5 with the row sum over the unrounded values).
template <int NPROD = 3, int CFMT = 0, int WAVES = 4, bool PVF8 = false>
GMF_DEVINL void scattn_h2p_body(float* lds, const int bid, const float* __restrict__ q_img, const float* __restrict__ k_img,
                                const float* __restrict__ v_img, const float* __restrict__ fus, const float* __restrict__ wst,
                                const float* __restrict__ vecs, float* out, int N, int tiles, int wgs_per_pair,
                                const float* __restrict__ c_dense, int n_items, int n_full, int ksplits,
                                float* __restrict__ part_o, float* __restrict__ part_ml, const float* __restrict__ next_wst,
                                const float* __restrict__ next_bias, const PairTab* __restrict__ ptab = nullptr,
                                const unsigned* __restrict__ v_scale = nullptr,
                                // [r4] non-null: q_img is not read - the workgroup projects its own Q' from the layer's f tile (below)
                                // (qf_img is NOT restrict: for every layer but the last it is the buffer `out` - f_{l+1} overwrites f_l in
                                // place.  The invariant that makes this safe: a tile of f is read only by the workgroup(s) of that tile - in
                                // the prologue, before the tile loop - and written only by that workgroup's epilogue (whole items) or by the
                                // LATER merge launch (split items): no tile is written before its last reader is done.  ADVICE r4)
                                const float* qf_img = nullptr, const float* __restrict__ qw_wst = nullptr,
                                const float* __restrict__ qw_bias = nullptr,
                                unsigned* __restrict__ stat_next = nullptr) {     // [r5] PvGuard::stat_next (whole items with the next PointCN)
  static_assert(!PVF8 || NPROD == 3, "the fp8 cross products belong to the three-product form");
  float* const ldsK = lds;
  float* const ldsV = lds + 2 * kStageFloats;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane_off16 = lane * 16;
  const AttnItem it = attn_item(bid, n_items, n_full, ksplits);
  if (!it.valid) return;                       // grid padding of the split tail (uniform per workgroup)
  const bool split = it.split;
  const int ks = it.ks;
  const int slot = it.item / wgs_per_pair, qblock = it.item - slot * wgs_per_pair;
  const int pair = ptab ? ptab[slot].ord : slot;       // ragged batch: the slots hold the pairs in a work-balanced order (PairTab::ord)
  // ragged batch: the pair's own row and tile counts (its slot in every image keeps the stride `tiles`); query blocks beyond
  // them have nothing to do (uniform per workgroup, before any barrier)
  if (ptab) N = ptab[pair].n;
  const int tiles_p = ptab ? (N + 31) >> 5 : tiles;
  if (qblock * WAVES >= tiles_p) return;
  // key tiles [t_begin, t_end) of this workgroup
  const int t_begin = split ? (tiles_p * ks) / ksplits : 0;
  const int t_end = split ? (tiles_p * (ks + 1)) / ksplits : tiles_p;
  const int tile_raw = qblock * WAVES + wave;
  const bool active = tile_raw < tiles_p;
  const int tile = active ? tile_raw : tiles_p - 1;
  const size_t pbase = (size_t)pair * tiles;
  const size_t toff = (pbase + tile) * (32 * C);

  f16x8 qh[8], ql[8];
  if (NPROD == 3 && WAVES == 4 && qf_img) {
    // [r4] Q' = Wq' f + bq' for this workgroup's 128 query rows HERE instead of in k_linear_h2 (PointDSC.py:56: projection_q).  Q'
    // is the one projection only its own workgroup consumes: computing it in the linear kernel meant 82 MB written and read
    // back per layer and a third of the stores of that kernel's write-bound Q'/K/V phase (k_linear_h2 without it: 307 -> 272 us).
    // The same arithmetic, bit for bit (k_linear_h2: mma_wx_h2n<8, 3> on the same weight stages, fmaf(acc, 2^-8, bias), fp16
    // split), so every other path - which still reads Q' images - agrees with this one.  The four weight stages (64 KiB) fill
    // the K / V rings' LDS before the tile loop needs it.
    FragH2<8> fx;
    {
      float fq[CF];
      load_frag_p32<CF>(fq, qf_img + toff, lane);
      fx.set(fq);
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) dma_4k_s(qw_wst + st * kStageFloats + wave * 1024, lds + st * kStageFloats + wave * 1024, lane_off16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const f16x8* lw = reinterpret_cast<const f16x8*>(lds + mb * kStageFloats) + lane;
      f32x16 acc = zero16();
      mma_wx_h2n<8, 3>(acc, lw, fx);
      float t[16], bq[16];
      load_vec_block(bq, qw_bias, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bq[r]);
      split8h(&t[0], qh[2 * mb], ql[2 * mb]);
      split8h(&t[8], qh[2 * mb + 1], ql[2 * mb + 1]);
    }
    __syncthreads();                               // every wave is done with the stages: the tile rings may overwrite them
  } else {
    const f16x8* qp = reinterpret_cast<const f16x8*>(q_img + (pbase + tile) * (size_t)kStageFloats) + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      qh[s] = qp[(0 * 8 + s) * 64];
      if (NPROD == 3) ql[s] = qp[(1 * 8 + s) * 64];
    }
  }
  // [r5] PVF8: the two CROSS products of S = K Q'^T run on the block-scaled fp8 pipe as well - per key tile 8 f16 MFMAs (K_hi Q'_hi) and 4
  // v_mfma_scale_f32_32x32x64_f8f6f4 (one per 32-channel block: [e4m3(k - k_hi) | e4m3(k)] x [e4m3(q) | e4m3(q_lo)]) instead of 24 f16
  // MFMAs: 512 instead of 768 matrix-pipe cycles.  Allowed by the same device-side guard as the P V form (a layer whose scores can
  // reach the bound multiplies everything on the f16 pipe): fp64 emulation of the whole encoder under the guard, max |logit - fp64|:
  // 3.00e-4 / 3.28e-4 on the ill-conditioned KITTI weight set (P V alone: 3.01e-4 / 3.29e-4; the reference's own fp32: 3.08e-4 / 2.68e-4),
  // 1.9e-6 / 4.5e-6 on 3DMatch-shape inputs (reference 1.8e-5 / 1.3e-5) - tests/tools/mx_cross_emulation.py.  The Q' operand is built
  // here, once per workgroup, from the planes (q_planes8); the K operand arrives in the K image (store_block_v8 on K blocks).
  i32x8 qb[4];
  unsigned qsw = 0;
  if (PVF8) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) q_planes8(qh[2 * cb], ql[2 * cb], qh[2 * cb + 1], ql[2 * cb + 1], cb, lane, qb[cb], qsw);
  }
  constexpr int kCTile16 = CFMT ? 128 : 256;           // 16-byte pieces per c tile
  const f32x4* const crow = reinterpret_cast<const f32x4*>(c_dense) + ((pbase + tile) * (size_t)tiles) * kCTile16 + lane;
  float c[16];                                         // CFMT 0 / 1: the compat values of the tile
  unsigned cw[8];                                      // CFMT 2: the tile's 16 x 16 bit as loaded (decoded where the scores are formed)
  // PVF8: the E8M0 scale bytes of V tile t for this lane (byte db), fetched with c one tile ahead and handed over at the tile top
  // ... [r5] and of K tile t + 1 (byte cb: the key's 32-channel block cb): the scale words of a tile lie as [V: 64 lanes | K: 64 lanes]
  const unsigned* const vsrow = PVF8 ? v_scale + pbase * 128 + lane : nullptr;
  unsigned vsw_next = 0, vsw = 0, ksw_next = 0, ksw = 0;
  const int psc = (lane & 32) ? 117 : 129;             // scale bytes of the probability operand: block 0 = e4m3(p / 4), block 1 = e4m3(p_lo 2^10)
  auto fetch_c = [&](int t) {
    const f32x4* ct = crow + (size_t)t * kCTile16;
    if (PVF8) {
      vsw_next = __builtin_nontemporal_load(vsrow + (size_t)t * 128);
      ksw_next = __builtin_nontemporal_load(vsrow + (size_t)min(t + 1, tiles_p - 1) * 128 + 64);      // (consumed by S_{t+1} during tile t)
    }
    if (CFMT == 1) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const f16x8 hv = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(ct + q2 * 64));
#pragma unroll
        for (int e = 0; e < 8; ++e) c[8 * q2 + e] = (float)hv[e];
      }
    } else if (CFMT >= 2) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const f32x4 v = __builtin_nontemporal_load(ct + q2 * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) cw[4 * q2 + e] = __float_as_uint(v[e]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = __builtin_nontemporal_load(ct + q * 64);
        c[4 * q + 0] = v[0]; c[4 * q + 1] = v[1]; c[4 * q + 2] = v[2]; c[4 * q + 3] = v[3];
      }
    }
  };
  // score of register r: c * s.  CFMT 2: u * s with u = 65535 c (v_cvt_f32_u32 with a 16-bit source select) - the factor
  // 1 / 65535 is applied where the scores are used (row maximum: once per tile; exponent: an fma instead of a subtraction).
  auto score = [&](int r, float sv) -> float {
    if (CFMT == 2) return (float)((r & 1) ? (cw[r >> 1] >> 16) : (cw[r >> 1] & 0xffffu)) * sv;
    return c[r] * sv;
  };
  auto expo = [&](float xv, float m_off) -> float {
    return __builtin_amdgcn_exp2f(CFMT == 2 ? fmaf(xv, kInvU16, -m_off) : xv - m_off);
  };
  // one product (hi x hi) or the three of the split-fp16 scheme
  auto mma_n = [&](f32x16& acc, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
    if (NPROD == 3) mma3(acc, ah, al, bh, bl);
    else acc = mfma_h16(ah, bh, acc);
  };
  auto to_planes2 = [&](float x0, float x1, f16x8& hi, f16x8& lo, int j) {
    if (NPROD == 3) split2h(x0, x1, hi, lo, j);
    else { const f32x2 xx = {x0, x1}; const f16x2 hh = __builtin_convertvector(xx, f16x2); hi[j] = hh[0]; hi[j + 1] = hh[1]; }
  };
  // PVF8: pair kp (keys 2 kp, 2 kp + 1 of the lane's 16) of the fp8 operand pb = [e4m3(ph / 4) x 16 | e4m3(pl 2^10) x 16], and the
  // row sums over the values the matrix pipe will multiply: ph (exact in fp16) and the DECODED pl8
  auto planes_f8 = [&](const f16x8& hi, const f16x8& lo, const int j, const int kp, i32x8& pb, float& ls_h, float& ls_l) {
    const f16x2 hh = {hi[j], hi[j + 1]}, ll = {lo[j], lo[j + 1]};
    const f16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
    const int w = kp >> 1;
    if (kp & 1) {
      pb[w] = cvt2_fp8_f16<true>(pb[w], hh, 4.0f);
      pb[4 + w] = cvt2_fp8_f16<true>(pb[4 + w], ll, 0x1p-10f);
      ls_l = __builtin_amdgcn_fdot2(dec2_fp8_f16<true>(pb[4 + w], 0x1p-10f), ones, ls_l, false);
    } else {
      int d0, d1;                                  // (the other half of each word is written by the odd pair: no zero-fill)
      asm("" : "=v"(d0), "=v"(d1));
      pb[w] = cvt2_fp8_f16<false>(d0, hh, 4.0f);
      pb[4 + w] = cvt2_fp8_f16<false>(d1, ll, 0x1p-10f);
      ls_l = __builtin_amdgcn_fdot2(dec2_fp8_f16<false>(pb[4 + w], 0x1p-10f), ones, ls_l, false);
    }
    ls_h = __builtin_amdgcn_fdot2(hh, ones, ls_h, false);
    asm volatile("" : "+v"(ls_h), "+v"(ls_l));     // the sums stay in this unit's issue gap (left alone they sink to the next tile top)
  };
  const float* gk = k_img + pbase * (size_t)kStageFloats;
  const float* gv = v_img + pbase * (size_t)kStageFloats;
  constexpr int kPiecesPerWave = ((NPROD == 3) ? 16 : 8) / WAVES;   // one product: only the high plane (the first 8 KiB) of a tile is used
  auto issue16k = [&](const float* g, float* l) {      // this wave's 4 of the 16 KiB-pieces of one tile
#pragma unroll
    for (int q = 0; q < kPiecesPerWave; ++q) dma_piece_1k_s(g + (wave + WAVES * q) * 256, l + (wave + WAVES * q) * 256, lane_off16);
  };
  auto issueK = [&](int t) { issue16k(gk + (size_t)t * kStageFloats, ldsK + (t & 1) * kStageFloats); };
  auto issueV = [&](int t) { issue16k(gv + (size_t)t * kStageFloats, ldsV + (t & 1) * kStageFloats); };

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float m_run = -INFINITY, l_half = 0.f;

  issueK(t_begin);
  if (t_begin + 1 < t_end) issueK(t_begin + 1);
  issueV(t_begin);
  fetch_c(t_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 s_a = zero16(), s_b;
  // S of one key tile in the fp8 form: 8 f16 MFMAs on the high planes + 4 scaled fp8 MFMAs on the cross planes
  auto s_cross8 = [&](f32x16 acc, const f16x8* lkt, const int cb, const unsigned kscale) -> f32x16 {
    const i32x4* l8 = reinterpret_cast<const i32x4*>(lkt);
    const i32x4 a_lo = l8[(1 * 8 + 2 * cb) * 64], a_hi = l8[(1 * 8 + 2 * cb + 1) * 64];
    const i32x8 ka = {a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
    return mfma_f8s2(ka, qb[cb], acc, cb, (int)kscale, (int)qsw);
  };
  if (active) {
    const f16x8* lk = reinterpret_cast<const f16x8*>(ldsK + (t_begin & 1) * kStageFloats) + lane;
    if (PVF8) {
      const unsigned ks0 = vsrow[(size_t)t_begin * 128 + 64];
#pragma unroll
      for (int s = 0; s < 8; ++s) s_a = mfma_h16(lk[(0 * 8 + s) * 64], qh[s], s_a);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) s_a = s_cross8(s_a, lk, cb, ks0);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) mma_n(s_a, lk[(0 * 8 + s) * 64], lk[(1 * 8 + s) * 64], qh[s], ql[s]);
    }
  }

  // top of tile t: K_{t+1}, V_t, c_t have landed and every wave is done with K_t and V_{t-1}.  The scores x = c_t * s_t
  // are formed BEFORE the refills are issued: the compiler guards every use of a loaded register with its own
  // s_waitcnt vmcnt, and behind younger LDS-DMA pieces that wait would cover them too.
  auto tile_top = [&](const int t, const f32x16& s_cur, float (&x)[16], float& mx) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (PVF8) vsw = vsw_next;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      x[r] = score(r, s_cur[r]);
      x[r + 1] = score(r + 1, s_cur[r + 1]);
      mx = __builtin_fmaxf(mx, __builtin_fmaxf(x[r], x[r + 1]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < t_end) fetch_c(t + 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  // piece q (0..3 of K_{t+2}, 4..7 of V_{t+1}) of this wave's share of the refills
  auto issue_piece = [&](const int t, const int q) {
    if ((q & 3) >= kPiecesPerWave) return;
    if (q < 4) {
      if (t + 2 < t_end) dma_piece_1k_s(gk + (size_t)(t + 2) * kStageFloats + (wave + WAVES * q) * 256,
                                        ldsK + (t & 1) * kStageFloats + (wave + WAVES * q) * 256, lane_off16);
    } else {
      dma_piece_1k_s(gv + (size_t)(t + 1) * kStageFloats + (wave + WAVES * (q - 4)) * 256,
                     ldsV + ((t + 1) & 1) * kStageFloats + (wave + WAVES * (q - 4)) * 256, lane_off16);
    }
  };
  // the same refills as ONE statement per tile and wave (dma_4k_s: the wave's four consecutive KiB under one M0 / address setup)
  auto issue_k4 = [&](const int t) {
    if (WAVES == 4 && t + 2 < t_end && (NPROD == 3 || wave < 2))
      dma_4k_s(gk + (size_t)(t + 2) * kStageFloats + wave * 1024, ldsK + (t & 1) * kStageFloats + wave * 1024, lane_off16);
  };
  auto issue_v4 = [&](const int t) {
    if (WAVES == 4 && (NPROD == 3 || wave < 2))
      dma_4k_s(gv + (size_t)(t + 1) * kStageFloats + wave * 1024, ldsV + ((t + 1) & 1) * kStageFloats + wave * 1024, lane_off16);
  };
  auto rescale = [&](const bool moved, const float alpha) {
    if (__any(moved)) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
  };
  // one pipelined tile (t + 1 < tiles): scores of tile t are in s_cur, S_{t+1} accumulates into s_next
  auto tile_step = [&](const int t, const f32x16& s_cur, f32x16& s_next) {
    float x[16];
    float mx = -INFINITY, m_off = 0.f, alpha = 1.f, ls = 0.f;
    tile_top(t, s_cur, x, mx);
    const f16x8* lv = reinterpret_cast<const f16x8*>(ldsV + (t & 1) * kStageFloats) + lane;
    const f16x8* lk = reinterpret_cast<const f16x8*>(ldsK + ((t + 1) & 1) * kStageFloats) + lane;
    bool moved = false;
    f16x8 ph0, pl0, ph1, pl1;
    // running maximum first, and the (rare) accumulator rescale with it: no branch may sit between the two phases, or
    // the compiler sinks the phase-1 vector work below it, out of the MFMA issue gaps
    {
      mx = xhalf_max_swap(mx);
      if (CFMT == 2) mx *= kInvU16;
      const float m_new = __builtin_fmaxf(m_run, mx);
      moved = m_new > m_run;
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      m_off = m_new - 10.0f;                 // P' = 2^10 P
      rescale(moved, alpha);
    }
    s_next = zero16();
    // ---- phase 1: S_{t+1} on the matrix pipe, tile t's scores / max / first exponentials in its issue gaps ----
    {
      f16x8 kh = lk[0], kl = kh;
      if (NPROD == 3) kl = lk[8 * 64];
      f16x8 kh_n = kh, kl_n = kl;
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const int s = u / 3, pr = u % 3;
        if (pr == 0 && s < 7) { kh_n = lk[(0 * 8 + s + 1) * 64]; if (NPROD == 3) kl_n = lk[(1 * 8 + s + 1) * 64]; }
        if (NPROD == 3 || pr == 2) s_next = mma3_part(pr, s_next, kh, kl, qh[s], ql[s]);
        if (pr == 2) { kh = kh_n; kl = kl_n; }
        if (u < 2) {
          // (the row maximum and the accumulator rescale are settled before the phase: see below)
        } else if (u <= 11) {
          const int r = u - 2;               // exponentials 0..9
          x[r] = expo(x[r], m_off);
          ls += x[r];
        } else if (u <= 15) {
          const int j = 2 * (u - 12);        // split pairs 0..3: the first 8 keys
          to_planes2(x[j], x[j + 1], ph0, pl0, j);
        } else if (u <= 21) {
          const int r = u - 6;               // exponentials 10..15
          x[r] = expo(x[r], m_off);
          ls += x[r];
        } else {
          const int j = 2 * (u - 22);        // split pairs 4, 5
          to_planes2(x[8 + j], x[8 + j + 1], ph1, pl1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- phase 2: O^T += V_t^T P^T, the remaining exponentials and the second split in its issue gaps ----
    {
      f16x8 vh = lv[0], vl = vh;
      if (NPROD == 3) vl = lv[8 * 64];
      f16x8 vh_n = vh, vl_n = vl;
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const int s2 = u / 12, db = (u % 12) / 3, pr = u % 3;
        if (pr == 0 && u < 21) {
          const int un = u + 3, slot = 2 * ((un % 12) / 3) + un / 12;
          vh_n = lv[(0 * 8 + slot) * 64]; if (NPROD == 3) vl_n = lv[(1 * 8 + slot) * 64];
        }
        if (NPROD == 3 || pr == 2) oacc[db] = mma3_part(pr, oacc[db], vh, vl, s2 ? ph1 : ph0, s2 ? pl1 : pl0);
        if (pr == 2) { vh = vh_n; vl = vl_n; }
        if (u < 2) {
          const int j = 4 + 2 * u;           // split pairs 6, 7 (needed from u = 12 on)
          to_planes2(x[8 + j], x[8 + j + 1], ph1, pl1, j);
        }
        if (u >= 3 && u < 11) issue_piece(t, u - 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l_half = fmaf(l_half, alpha, ls);
  };
  // PVF8: the tile step with NO serial vector work at the tile top.  The compat product, the row maximum and the (rare) rescale
  // of tile t are interleaved with the first S_{t+1} MFMAs (units 0..3), the exponentials and splits follow in units 4..20;
  // behind the tile barrier a wave goes straight back to the matrix pipe (1080-1087 us per launch against 1091-1097 with those ~45
  // dependent instructions between the barrier and the first MFMA, on all four waves of the workgroup at once).
  // What the interleaving is for: a wave's own vector instructions do NOT overlap its MFMA (tools/ubench/mfma_valu_overlap.hip:
  // 34.5 + 3.2 K cycles for one MFMA + K v_fma) - it is the OTHER wave of the SIMD that runs its vector work under this wave's
  // MFMAs, and only a fine alternation of the two kinds gives it the chance.
  constexpr float kLazy = 5.0f;
  auto tile_step_f8 = [&](const int t, const f32x16& s_cur, f32x16& s_next) {
    float x[16];
    float mx = -INFINITY, m_off = 0.f, alpha = 1.f, ls = 0.f, ls_l = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    vsw = vsw_next;
    ksw = ksw_next;
    const f16x8* lv = reinterpret_cast<const f16x8*>(ldsV + (t & 1) * kStageFloats) + lane;
    const f16x8* lk = reinterpret_cast<const f16x8*>(ldsK + ((t + 1) & 1) * kStageFloats) + lane;
    f16x8 ph0, pl0, ph1, pl1;
    i32x8 pb;
    f16x8 vr[3];
    auto hslot = [](int u) { return (0 * 8 + 2 * (u & 3) + (u >> 2)) * 64; };
    s_next = zero16();
    {
      // [r5] 12 matrix instructions in the phase's 24 units: K_hi Q'_hi of k-step s at unit 3 s + 2, the scaled fp8 MFMA of channel block
      // cb (k-steps 2 cb, 2 cb + 1) at unit 6 cb + 3; its two 16-byte K fragments are requested one k-step earlier (unit 6 cb)
      const i32x4* lk8 = reinterpret_cast<const i32x4*>(lk);
      f16x8 kh = lk[0];
      f16x8 kh_n = kh;
      i32x4 k8a = lk8[(1 * 8) * 64], k8b = lk8[(1 * 8 + 1) * 64];
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const int s = u / 3, pr = u % 3;
        if (pr == 0 && s < 7) kh_n = lk[(0 * 8 + s + 1) * 64];
        if (pr == 2) s_next = mfma_h16(kh, qh[s], s_next);
        if (pr == 0 && (s & 1)) {
          const int cb = s >> 1;
          const i32x8 ka = {k8a[0], k8a[1], k8a[2], k8a[3], k8b[0], k8b[1], k8b[2], k8b[3]};
          s_next = mfma_f8s2(ka, qb[cb], s_next, cb, (int)ksw, (int)qsw);
          if (cb < 3) { k8a = lk8[(1 * 8 + 2 * cb + 2) * 64]; k8b = lk8[(1 * 8 + 2 * cb + 3) * 64]; }
        }
        if (pr == 2) kh = kh_n;
        if (u < 4) {
#pragma unroll
          for (int r = 4 * u; r < 4 * u + 4; r += 2) {
            x[r] = score(r, s_cur[r]);
            x[r + 1] = score(r + 1, s_cur[r + 1]);
            mx = __builtin_fmaxf(mx, __builtin_fmaxf(x[r], x[r + 1]));
          }
        }
        if (u == 3) {
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < t_end) fetch_c(t + 1);            // (c_t is consumed)
          __builtin_amdgcn_sched_barrier(0);
          mx = xhalf_max_swap(mx);
          if (CFMT == 2) mx *= kInvU16;
          // LAZY reference: m_run only moves when a score exceeds it by more than kLazy (2^kLazy in probability), and the
          // probabilities are formed as 2^(10 - kLazy) P, so that p <= 2^10 either way (the e4m3 planes' range).  A wave's 32 rows
          // set new maxima in half of its 157 tiles (in nearly all of a key-split workgroup's 26), each costing the 64
          // multiplications of the accumulator rescale on the critical path; beyond the margin they are rare.  The scale of
          // numerator and denominator is the same power of two: O / l does not change.
          const bool moved = mx > m_run + kLazy;
          const float m_new = moved ? mx : m_run;
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
          m_off = m_new - (10.0f - kLazy);
          rescale(moved, alpha);
        }
        if (u >= 4 && u < 20) x[u - 4] = expo(x[u - 4], m_off);
        if (u >= 6 && u <= 18 && (u & 1) == 0) {         // split pair k once its two exponentials are there (units 5 + 2 k)
          const int k = (u - 6) >> 1;
          if (k < 4) split2h(x[2 * k], x[2 * k + 1], ph0, pl0, 2 * k);
          else split2h(x[2 * k], x[2 * k + 1], ph1, pl1, 2 * k - 8);
        }
        if (u == 20) split2h(x[14], x[15], ph1, pl1, 6);
        // the tile ring's refills ride in the lighter units of this phase, each as ONE statement (dma_4k_s): K_{t+2} (its slot held K_t, read
        // during tile t - 1) at unit 12, V_{t+1} (slot of V_{t-1}) at unit 20 [round 4; round 5: units 16 / 23, below] - 1028 us per launch against 1043 for eight separately set-up
        // pieces at units 12..15 / 20..23, which in turn beat V in the first units of phase 2 by 1 % (tools/ubench/ablate_h2p.py p8_*)
        // [r5] with the cross products of S on the fp8 pipe the phase is 256 matrix-pipe cycles shorter; re-measured: (16, 23) 11.60-11.62 ms per
        // pass of 12 launches, (12, 20) 11.67-11.69, (8, 16) 12.05, (0, 8) 12.54, V in phase 2 11.74-11.90 (profiles/r05_attention_notes.txt)
        if (u == 16) issue_k4(t);
        if (u == 23) issue_v4(t);
        if (u >= 21) vr[u - 21] = lv[hslot(u - 21)];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      const i32x4* lv8 = reinterpret_cast<const i32x4*>(lv);
      i32x4 fa[2][2];
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        if (u < 8) {
          const int s2 = u >> 2, db = u & 3;
          oacc[db] = mfma_h16(vr[u % 3], s2 ? ph1 : ph0, oacc[db]);
          if (u + 3 < 8) vr[u % 3] = lv[hslot(u + 3)];
          if (u == 4 || u == 6) { const int fd = (u - 4) >> 1; fa[fd][0] = lv8[(1 * 8 + 2 * fd) * 64]; fa[fd][1] = lv8[(1 * 8 + 2 * fd + 1) * 64]; }
          planes_f8(u < 4 ? ph0 : ph1, u < 4 ? pl0 : pl1, 2 * (u & 3), u, pb, ls, ls_l);
        } else {
          const int db = u - 8;
          const i32x4 a_lo = fa[db & 1][0], a_hi = fa[db & 1][1];
          const i32x8 va = {a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
          oacc[db] = mfma_f8s(va, pb, oacc[db], db, (int)vsw, psc);
          if (db < 2) { fa[db & 1][0] = lv8[(1 * 8 + 2 * db + 4) * 64]; fa[db & 1][1] = lv8[(1 * 8 + 2 * db + 5) * 64]; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l_half = fmaf(l_half, alpha, ls + ls_l);
  };
  // the last tile (keys >= N masked out), no S_{t+1} to overlap with
  auto tile_last = [&](const int t, const f32x16& s_cur) {
    float x[16];
    float mx = -INFINITY, ls = 0.f;
    tile_top(t, s_cur, x, mx);
    const f16x8* lv = reinterpret_cast<const f16x8*>(ldsV + (t & 1) * kStageFloats) + lane;
    const int jbase = t * 32 + 4 * h;
    mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = 8 * (r >> 2) + (r & 3);
      x[r] = (jbase + jl < N) ? x[r] : -INFINITY;
      mx = __builtin_fmaxf(mx, x[r]);
    }
    mx = xhalf_max_swap(mx);
    if (CFMT == 2) mx *= kInvU16;
    const float m_new = __builtin_fmaxf(m_run, mx);
    const bool moved = m_new > m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const float m_off = m_new - (PVF8 ? 10.0f - kLazy : 10.0f);      // (the scale of the tiles before it: tile_step_f8)
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = expo(x[r], m_off); if (!PVF8) ls += x[r]; }
    rescale(moved, alpha);
    if (PVF8) {
      i32x8 pb;
      float ls_l = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 ph, pl;
        split8h(&x[8 * s2], ph, pl);
#pragma unroll
        for (int j = 0; j < 8; j += 2) planes_f8(ph, pl, j, 4 * s2 + (j >> 1), pb, ls, ls_l);
#pragma unroll
        for (int db = 0; db < 4; ++db) oacc[db] = mfma_h16(lv[(0 * 8 + 2 * db + s2) * 64], ph, oacc[db]);
      }
      ls += ls_l;
      const i32x4* lv8 = reinterpret_cast<const i32x4*>(lv);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const i32x4 a_lo = lv8[(1 * 8 + 2 * db) * 64], a_hi = lv8[(1 * 8 + 2 * db + 1) * 64];
        const i32x8 va = {a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
        oacc[db] = mfma_f8s(va, pb, oacc[db], db, (int)vsw, psc);
      }
    } else {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int slot = 2 * db + s2;
        mma_n(oacc[db], lv[(0 * 8 + slot) * 64], lv[(1 * 8 + slot) * 64], ph, pl);
      }
    }
    }
    l_half = fmaf(l_half, alpha, ls);
  };
  if (!active) {
    // padding wave of a pair's last workgroup (tiles is rarely a multiple of 4): it owns no queries, but a quarter of
    // every K/V stage and a seat at every barrier - one barrier per tile, the refills of tile t issued during tile t
    for (int t = t_begin; t < t_end; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 1 < t_end) {
        if (PVF8) { issue_k4(t); issue_v4(t); }      // (the same KiB of every stage as this wave would fetch with queries: tile_step_f8)
        else {
#pragma unroll
          for (int q = 0; q < 8; ++q) issue_piece(t, q);
        }
      }
    }
  } else {
    int t = t_begin;
    if (PVF8) {
      for (; t + 2 < t_end; t += 2) {
        tile_step_f8(t, s_a, s_b);
        tile_step_f8(t + 1, s_b, s_a);
      }
      if (t + 1 < t_end) { tile_step_f8(t, s_a, s_b); tile_last(t + 1, s_b); }
      else tile_last(t, s_a);
    } else {
    for (; t + 2 < t_end; t += 2) {          // explicit ping-pong: no accumulator copies at the loop back-edge
      tile_step(t, s_a, s_b);
      tile_step(t + 1, s_b, s_a);
    }
    if (t + 1 < t_end) { tile_step(t, s_a, s_b); tile_last(t + 1, s_b); }
    else tile_last(t, s_a);
    }
  }
  if (split) {
    // partial result of this key range: un-normalised O as a P32 tile image, row maximum and row sum
    if (active) {
      float of[CF];
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) of[16 * db + r] = oacc[db][r];
      const size_t n_tiles_all = (size_t)(n_items / wgs_per_pair) * tiles;     // B * tiles
      const size_t pt = (size_t)ks * n_tiles_all + pbase + tile;
      store_frag_p32<CF>(part_o + pt * (32 * C), of, lane);
      const float lsum = xhalf_sum(l_half);
      if (h == 0) { part_ml[(pt * 32 + (lane & 31)) * 2] = m_run; part_ml[(pt * 32 + (lane & 31)) * 2 + 1] = lsum; }
    }
    return;
  }

  // ---- epilogue: normalise, fc_message (split-fp16 MFMA), add the Fusion-2 branch ----
  float o[CF];
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[16 * db + r] = oacc[db][r] * inv;
  }
  __syncthreads();
  float* const lvec = lds + 2 * kStageFloats;    // (the epilogue streams its stages through the first two of the four slots)
  stage_tail_vecs(lvec, vecs, next_wst ? next_bias : nullptr, wave, lane);
  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, WAVES, lane, wst, 5, next_wst, next_wst ? 4 : 0);
  ss.prime();
  const FusTile ft{fus + toff};
  if (next_wst) scattn_epilogue_h2<true>(o, active, ss, lvec, ft, out + toff, lane, h, lvec + 2 * C, stat_next, pair,
                                         tile_raw * 32 + (lane & 31) < N);
  else scattn_epilogue_h2<false>(o, active, ss, lvec, ft, out + toff, lane, h);
}

// [r5] PVF8 kernels under the guard (PvGuard): the pair of this workgroup's item decides between the two forms of the body -
// one uniform branch at the top, both bodies in the kernel.  (The V image of the pair was written in the matching format.)
GMF_DEVINL bool attn_pv_on(const unsigned* __restrict__ v_scale, const PvGuard& guard, int bid, int n_items, int n_full, int ksplits,
                           int wgs_per_pair, const PairTab* __restrict__ ptab = nullptr) {
  if (!guard.stat) return v_scale != nullptr;
  const AttnItem it = attn_item(bid, n_items, n_full, ksplits);
  const int slot = it.valid ? it.item / wgs_per_pair : 0;
  return pv_planes_on(v_scale, guard, ptab ? ptab[slot].ord : slot);
}

template <int NPROD, int CFMT, bool PVF8 = false>
__global__ void __launch_bounds__(256, 2)
k_scattn_h2p(const float* __restrict__ q_img, const float* __restrict__ k_img, const float* __restrict__ v_img,
             const float* __restrict__ fus, const float* __restrict__ wst, const float* __restrict__ vecs,
             float* out, int N, int tiles, int wgs_per_pair, const float* __restrict__ c_dense,
             int n_items, int n_full, int ksplits, float* __restrict__ part_o, float* __restrict__ part_ml,
             const float* __restrict__ next_wst, const float* __restrict__ next_bias, const PairTab* __restrict__ ptab,
             const unsigned* __restrict__ v_scale, const float* qf_img, const float* __restrict__ qw_wst,
             const float* __restrict__ qw_bias, const PvGuard guard) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStageFloats];
  tail_priority(blockIdx.x, gridDim.x);
  if (!PVF8 || attn_pv_on(v_scale, guard, blockIdx.x, n_items, n_full, ksplits, wgs_per_pair, ptab))
    scattn_h2p_body<NPROD, CFMT, 4, PVF8>(lds, blockIdx.x, q_img, k_img, v_img, fus, wst, vecs, out, N, tiles, wgs_per_pair, c_dense, n_items,
                                          n_full, ksplits, part_o, part_ml, next_wst, next_bias, ptab, v_scale, qf_img, qw_wst, qw_bias,
                                          guard.stat_next);
  else
    scattn_h2p_body<NPROD, CFMT, 4, false>(lds, blockIdx.x, q_img, k_img, v_img, fus, wst, vecs, out, N, tiles, wgs_per_pair, c_dense, n_items,
                                           n_full, ksplits, part_o, part_ml, next_wst, next_bias, ptab, nullptr, qf_img, qw_wst, qw_bias,
                                           guard.stat_next);
}

// =========================================================================================
// k_scattn_fast: the attention of the THROUGHPUT numerics mode (gmf_set_tuning "precision" = 1) on large grids.  With one
// fp16 product instead of three a key tile is 16 MFMAs (512 matrix-pipe cycles): shorter than one memory round trip, so
// the one-tile-ahead streams of k_scattn_h2p leave its waves waiting (measured: 58 % of their cycles).  This form keeps
// every stream TWO tiles ahead, and lands all of them in the LDS (nothing is in flight towards a register, so the
// compiler's register allocation cannot interfere with data that has not arrived yet):
//   LDS (72 KiB, two workgroups per CU): K ring [3][8 KiB] | V ring [3][8 KiB] (high planes only) | c ring [wave][3][2 KiB]
//   (fp16 compat tiles, each wave its own query tile's);
//   per tile and wave 6 LDS-DMA pieces in a fixed order (2 of K_{t+2}, 2 of V_{t+2}, 2 of c_{t+2}), all issued
//   unconditionally (past the end: the last tile again, into slots nobody reads) and all outside the compiler's s_waitcnt
//   bookkeeping - so ONE counted wait, s_waitcnt vmcnt(6) at the tile top, guarantees everything issued two tiles ago or
//   earlier while the six pieces of the previous tile stay in flight.
// No intra-wave MFMA / VALU interleaving: two workgroups per CU overlap each other.  Same work mapping (whole items only),
// same epilogue (fc_message + Fusion-2 branch + next PointCN, parity arithmetic) as k_scattn_h2p.
// =========================================================================================
__global__ void __launch_bounds__(256, 2)
k_scattn_fast(const float* __restrict__ q_img, const float* __restrict__ k_img, const float* __restrict__ v_img,
              const float* __restrict__ fus, const float* __restrict__ wst, const float* __restrict__ vecs,
              float* __restrict__ out, int N, int tiles, int wgs_per_pair, const float* __restrict__ c_half, int n_items,
              const float* __restrict__ next_wst, const float* __restrict__ next_bias) {
  constexpr int WAVES = 4;
  constexpr int kHiFloats = kStageFloats / 2;          // the high plane of a tile image: 8 KiB
  constexpr int kCFloats = 512;                        // one fp16 compat tile: 2 KiB
  __shared__ __attribute__((aligned(16))) float lds[6 * kHiFloats + WAVES * 3 * kCFloats];
  float* const ldsK = lds;
  float* const ldsV = lds + 3 * kHiFloats;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const ldsC = lds + 6 * kHiFloats + wave * 3 * kCFloats;
  const unsigned lane_off16 = lane * 16;
  const AttnItem it = attn_item(blockIdx.x, n_items, 0x7fffffff, 1);
  if (!it.valid) return;
  const int pair = it.item / wgs_per_pair, qblock = it.item - pair * wgs_per_pair;
  const int tile_raw = qblock * WAVES + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t pbase = (size_t)pair * tiles;
  const size_t toff = (pbase + tile) * (32 * C);

  f16x8 qh[8];
  {
    const f16x8* qp = reinterpret_cast<const f16x8*>(q_img + (pbase + tile) * (size_t)kStageFloats) + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) qh[s] = qp[s * 64];
  }
  const float* gc = c_half + ((pbase + tile) * (size_t)tiles) * kCFloats;       // this wave's row of compat tiles
  const float* gk = k_img + pbase * (size_t)kStageFloats;
  const float* gv = v_img + pbase * (size_t)kStageFloats;
  // the 6 pieces of tile t (clamped to the last one) into ring slot `slot`: this wave's 2 of the 8 KiB-pieces of the K and
  // of the V high plane, and the 2 KiB of its own compat tile
  auto issue_tile = [&](int t, const int slot) {
    t = min(t, tiles - 1);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      dma_piece_1k_s(gk + (size_t)t * kStageFloats + (wave + WAVES * q) * 256, ldsK + slot * kHiFloats + (wave + WAVES * q) * 256, lane_off16);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      dma_piece_1k_s(gv + (size_t)t * kStageFloats + (wave + WAVES * q) * 256, ldsV + slot * kHiFloats + (wave + WAVES * q) * 256, lane_off16);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      dma_piece_1k_s(gc + (size_t)t * kCFloats + q * 256, ldsC + slot * kCFloats + q * 256, lane_off16);
  };
  // x' = c * s - m_off in ONE instruction, c one fp16 half of a 32-bit LDS word: the compiler selects v_fma_mix_f32 (fp16 x fp32
  // + fp32 -> fp32) for this expression.  (Written as an asm statement the same instruction read the scores before the MFMA
  // had delivered them: hipcc's hazard recogniser does not put the MFMA -> VALU wait states in front of inline assembly.)
  auto cfma = [&](const float cdw, const float sv, const float neg_m_off, const bool hi) {
    const f16x2 hv = __builtin_bit_cast(f16x2, cdw);
    return __builtin_fmaf((float)(hi ? hv[1] : hv[0]), sv, neg_m_off);
  };

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float l_half = 0.f;

  // The running maximum is taken LAZILY: the exponent offset m_off stays what it was as long as no score of the tile exceeds it
  // by more than kSlack (the probabilities P' = 2^(x - m_off) then stay below 2^kSlack <= fp16's range, and a softmax does not
  // care which offset its numerator and denominator share), so the steady state of a tile is: 16 fused c * s - m_off, their
  // maximum (8 x v_max3), 16 exponentials, 8 packed conversions and 8 dot-products for the row sum - no exp2 of a maximum
  // difference, no accumulator rescale.  Only when a score climbs above the slack (the first tile, and rarely after) the
  // offset moves and accumulators and row sum are rescaled.
  constexpr float kSlack = 15.0f;
  float m_off = 0.f;                                   // defined by the first tile (l_half = 0, oacc = 0: nothing to rescale)
  auto tile_step = [&](auto last_tag, auto first_tag, const int t, const int slot) {
    constexpr bool LAST = decltype(last_tag)::value;       // only the last key tile can hold keys >= N
    constexpr bool FIRST = decltype(first_tag)::value;
    // everything issued two tiles ago or earlier has landed (tile t's K, V, c among it); the previous tile's 6 stay in flight
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();                                   // ... for every wave; and every wave is done with tile t - 1's slots
    issue_tile(t + 2, slot == 0 ? 2 : slot - 1);       // -> slot (t + 2) % 3 = tile t - 1's
    if (!active) return;
    const f16x8* lk = reinterpret_cast<const f16x8*>(ldsK + slot * kHiFloats) + lane;
    const f16x8* lv = reinterpret_cast<const f16x8*>(ldsV + slot * kHiFloats) + lane;
    const f32x4* lc = reinterpret_cast<const f32x4*>(ldsC + slot * kCFloats) + lane;
    const f32x4 c0a = lc[0], c0b = lc[64];
    f32x16 sc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) sc = mfma_h16(lk[s * 64], qh[s], sc);
    float x[16];
    const float nm = FIRST ? 0.f : -m_off;
    const int jbase = t * 32 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float cdw = (r < 8) ? c0a[r >> 1] : c0b[(r - 8) >> 1];
      x[r] = cfma(cdw, sc[r], nm, (r & 1) != 0);
      if (LAST) {
        const int jl = 8 * (r >> 2) + (r & 3);
        x[r] = (jbase + jl < N) ? x[r] : -INFINITY;
      }
    }
    float mx = __builtin_fmaxf(x[0], x[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = __builtin_fmaxf(mx, __builtin_fmaxf(x[r], x[r + 1]));
    mx = xhalf_max_swap(mx);                            // the row's maximum over this tile, relative to the offset
    if (FIRST || __any(mx > kSlack)) {
      // new offset: this tile's maximum lands at 2^10 (as in the parity kernel); rows that stay within the slack keep theirs
      const float shift = (FIRST || mx > kSlack) ? mx - 10.0f : 0.f;
      const float alpha = FIRST ? 1.f : __builtin_amdgcn_exp2f(-shift);
      m_off = FIRST ? shift : m_off + shift;
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] -= shift;
      if (!FIRST) {
        l_half *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2 xx = {x[8 * s2 + j], x[8 * s2 + j + 1]};
        const f16x2 hh = __builtin_convertvector(xx, f16x2);
        ph[j] = hh[0]; ph[j + 1] = hh[1];
        // the row sum adds the ROUNDED probabilities, the ones the matrix pipe multiplies (v_dot2_f32_f16: two per instruction)
        l_half = __builtin_amdgcn_fdot2(hh, f16x2{(_Float16)1, (_Float16)1}, l_half, false);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[db] = mfma_h16(lv[(2 * db + s2) * 64], ph, oacc[db]);
    }
  };

  // prologue: tiles 0 and 1 (past-the-end tiles clamp to the last one)
  issue_tile(0, 0);
  issue_tile(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const std::false_type no;
    const std::true_type yes;
    if (tiles == 1) {
      tile_step(yes, yes, 0, 0);
    } else {
      tile_step(no, yes, 0, 0);
      int slot = 1;
      for (int t = 1; t + 1 < tiles; ++t) {
        tile_step(no, no, t, slot);
        slot = (slot == 2) ? 0 : slot + 1;
      }
      tile_step(yes, no, tiles - 1, slot);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped refills of the last tiles: nothing may land after the epilogue starts
  // ---- epilogue: normalise, fc_message, Fusion-2 branch, next PointCN (parity arithmetic) ----
  float o[CF];
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[16 * db + r] = oacc[db][r] * inv;
  }
  __syncthreads();
  float* const lvec = lds + 2 * kStageFloats;
  stage_tail_vecs(lvec, vecs, next_wst ? next_bias : nullptr, wave, lane);
  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, WAVES, lane, wst, 5, next_wst, next_wst ? 4 : 0);
  ss.prime();
  const FusTile ft{fus + toff};
  if (next_wst) scattn_epilogue_h2<true>(o, active, ss, lvec, ft, out + toff, lane, h, lvec + 2 * C);
  else scattn_epilogue_h2<false>(o, active, ss, lvec, ft, out + toff, lane, h);
}

// k_small_attn_ff: small grids, the second of the three launches of a layer - the first n_attn workgroups are the key-split
// attention workgroups (every item split: n_full = 0; they need only Q', K, V and c), the remaining ones the hidden-split
// feed-forward workgroups (they need only x1).  Both write partial results; k_scattn_merge adds them up.  As separate
// launches the two cost their sum on these latency-bound grids (B = 1, N = 5000: 44 + 18 us per layer); the attention
// workgroups are one per CU (attn_item), so the feed-forward workgroups find a second slot on every CU.
template <int CFMT, bool PVF8 = false>
__global__ void __launch_bounds__(256, 2)
k_small_attn_ff(const int n_attn, const float* __restrict__ q_img, const float* __restrict__ k_img,
                const float* __restrict__ v_img, const float* __restrict__ tail_wst, const float* __restrict__ tail_vecs, int N,
                int tiles, int wgs_per_pair, const float* __restrict__ c_dense, int n_items, int ksplits,
                float* __restrict__ part_o, float* __restrict__ part_ml, const float* __restrict__ x1,
                const float* __restrict__ ff_wst, const float* __restrict__ ff_vecs, float* __restrict__ ff_part, int n_pairs,
                int ff_hs, const unsigned* __restrict__ v_scale, const PvGuard guard,
                const PairTab* __restrict__ ptab) {                      // [r5] ragged batches: the pairs' own rows
  __shared__ __attribute__((aligned(16))) float lds[4 * kStageFloats];
  if ((int)blockIdx.x < n_attn) {
    // (n_full = 0: every item is split and leaves through the partial-result branch; `fus` / `out` of the whole-item epilogue
    // are never touched - they get valid pointers all the same, a literal null there crashes this compiler's optimiser)
    if (!PVF8 || attn_pv_on(v_scale, guard, blockIdx.x, n_items, 0, ksplits, wgs_per_pair, ptab))
      scattn_h2p_body<3, CFMT, 4, PVF8>(lds, blockIdx.x, q_img, k_img, v_img, x1, tail_wst, tail_vecs, ff_part, N, tiles, wgs_per_pair, c_dense,
                                        n_items, 0, ksplits, part_o, part_ml, nullptr, nullptr, ptab, v_scale);
    else
      scattn_h2p_body<3, CFMT, 4, false>(lds, blockIdx.x, q_img, k_img, v_img, x1, tail_wst, tail_vecs, ff_part, N, tiles, wgs_per_pair, c_dense,
                                         n_items, 0, ksplits, part_o, part_ml, nullptr, nullptr, ptab, nullptr);
  } else {
    const int id = (int)blockIdx.x - n_attn;           // (bx, pair, z) with z fastest: the splits of a row block start together
    const int z = id % ff_hs, r = id / ff_hs;
    const int pair = r / wgs_per_pair, bx = r - pair * wgs_per_pair;
    if (ptab && bx * 4 >= ((ptab[pair].n + 31) >> 5)) return;            // ragged batch: a row block beyond the pair's own tiles
    fusion_ff_h2p_body<true>(lds, bx, pair, n_pairs, z, ff_hs, x1, ff_wst, ff_vecs, ff_part, tiles, ff_part);   // (x2_out unused: hs > 1)
  }
}

// All five fc_message weight stages in LDS at once (80 KiB): for grids that do not fill the chip anyway (the merge step
// of the key-split form) the chain "wait for a stage, use it for ~0.3 us, wait for the next" is what the kernel costs, so
// every stage is requested up front and there is a single wait.
struct StagesPreloaded {
  const float* base;
  int used, lane;
  // n_stages of `g`, then (if given) 4 stages of `g2`
  GMF_DEVINL void init(float* lds, int wave, int lane_, const float* g, int n_stages, const float* g2 = nullptr) {
    base = lds; used = 0; lane = lane_;
    for (int st = 0; st < n_stages; ++st)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dma_piece_1k(g + (size_t)st * kStageFloats + (wave + 4 * q) * 256, lds + st * kStageFloats + (wave + 4 * q) * 256, lane_);
    if (g2)
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dma_piece_1k(g2 + (size_t)st * kStageFloats + (wave + 4 * q) * 256, lds + (n_stages + st) * kStageFloats + (wave + 4 * q) * 256, lane_);
  }
  GMF_DEVINL const float4* acquire() {
    if (used == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const float* cur = base + used * kStageFloats;
    ++used;
    return reinterpret_cast<const float4*>(cur) + lane;
  }
};

// k_scattn_merge: combines the `ksplits` partial results of k_scattn_h2p<.., KSPLIT> per query tile - common maximum M,
// O = sum_k O_k 2^(m_k - M), l = sum_k l_k 2^(m_k - M), splits added in index order - and runs the epilogue.
// One workgroup per SPLIT item of attn_item's mapping: grid 8 * (largest number of split items of an XCD), block 256.  The
// partials are fetched four splits at a time (all loads of a group in flight).
__global__ void __launch_bounds__(256, 1)
k_scattn_merge(const float* __restrict__ part_o, const float* __restrict__ part_ml, const float* __restrict__ fus,
               const float* __restrict__ wst, const float* __restrict__ vecs, float* __restrict__ out, int tiles,
               int wgs_per_pair, int n_items, int n_full, int ksplits, const float* __restrict__ next_wst,
               const float* __restrict__ next_bias, const float* __restrict__ ff_part, int ff_hs, const float* __restrict__ x1,
               const float* __restrict__ ff_b2, unsigned* __restrict__ stat_next, int n_rows,     // [r5] PvGuard::stat_next, the pairs' row count
               const PairTab* __restrict__ ptab = nullptr) {                                     // [r5] ragged batch: key-split items too
  // (behind the nine stages: the biases of the epilogue - a bias fetched from global memory after each stage's MFMAs is a
  // memory round trip per stage on a grid where nothing else runs on the CU)
  __shared__ __attribute__((aligned(16))) float lds[9 * kStageFloats + kTailVecFloats];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // split item r of XCD x: the workgroup slot (x, n_full + r * 1) of a ksplits = 1 mapping
  const AttnItem it = attn_item(blockIdx.x + 8 * min(n_full, (n_items >> 3) + ((int)(blockIdx.x & 7) < (n_items & 7) ? 1 : 0)),
                                n_items, n_full, 1);
  if (!it.valid) return;
  const int slot = it.item / wgs_per_pair;
  const int pair = ptab ? ptab[slot].ord : slot;           // (ragged batch: the slots hold the pairs in a work-balanced order, PairTab::ord)
  if (ptab) n_rows = ptab[pair].n;
  const int tiles_p = ptab ? (n_rows + 31) >> 5 : tiles;   // ragged batch: the pair's own tiles; its slot in every image keeps the stride `tiles`
  if ((it.item - slot * wgs_per_pair) * 4 >= tiles_p) return;            // (uniform per workgroup, before any barrier)
  const int tile_raw = (it.item - slot * wgs_per_pair) * 4 + wave;
  const bool active = tile_raw < tiles_p;
  const int tile = active ? tile_raw : tiles_p - 1;
  const size_t n_tiles_all = (size_t)(n_items / wgs_per_pair) * tiles;
  const size_t pt0 = (size_t)pair * tiles + tile;
  const size_t toff = pt0 * (32 * C);
  float* const lvec = lds + 9 * kStageFloats;
  stage_tail_vecs(lvec, vecs, next_wst ? next_bias : nullptr, wave, lane);
  vecs = lvec;
  next_bias = lvec + 2 * C;
  StagesPreloaded ss;
  ss.init(lds, wave, lane, wst, 5, next_wst);
  // small grids: the Fusion-2 branch arrives as hidden-split partials (k_small_attn_ff): x2 = sum_z part[z] 2^-8 + b2 + x1, summed
  // in index order HERE - all (hs + 1) x 16 loads of the tile in flight together with the weight stages - instead of block by
  // block between the epilogue's last stages
  float x2[CF];
  if (ff_part) {
    const float* pt = ff_part + toff;
    const size_t zs = n_tiles_all * (32 * C);
    float acc[CF], xr[CF];
    load_frag_p32<CF>(acc, pt, lane);
    load_frag_p32<CF>(xr, x1 + toff, lane);
    for (int z = 1; z < ff_hs; ++z) {
      float pz[CF];
      load_frag_p32<CF>(pz, pt + (size_t)z * zs, lane);
#pragma unroll
      for (int e = 0; e < CF; ++e) acc[e] += pz[e];
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float b[16];
      load_vec_block(b, ff_b2, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) x2[16 * mb + r] = fmaf(acc[16 * mb + r], kH2Inv, b[r]) + xr[16 * mb + r];
    }
  }
  float mk[8], lk[8];
  float M = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mk[k] = -INFINITY; lk[k] = 0.f;
    if (k < ksplits) {
      const float2 ml = *reinterpret_cast<const float2*>(part_ml + ((k * n_tiles_all + pt0) * 32 + i) * 2);
      mk[k] = ml.x; lk[k] = ml.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) M = fmaxf(M, mk[k]);
  float o[CF];
#pragma unroll
  for (int e = 0; e < CF; ++e) o[e] = 0.f;
  float l = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < 8; k0 += 4) {
    if (k0 >= ksplits) break;
    float ok[4][CF];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < ksplits) load_frag_p32<CF>(ok[j], part_o + ((k0 + j) * n_tiles_all + pt0) * (32 * C), lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (k0 + j < ksplits) {
        const float wk = __builtin_amdgcn_exp2f(mk[k0 + j] - M);
        l = fmaf(lk[k0 + j], wk, l);
#pragma unroll
        for (int e = 0; e < CF; ++e) o[e] = fmaf(ok[j][e], wk, o[e]);
      }
    }
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int e = 0; e < CF; ++e) o[e] *= inv;
  if (ff_part) {
    const FusRegs fp{x2};
    if (next_wst) scattn_epilogue_h2<true>(o, active, ss, vecs, fp, out + toff, lane, h, next_bias, stat_next, pair, tile_raw * 32 + i < n_rows);
    else scattn_epilogue_h2<false>(o, active, ss, vecs, fp, out + toff, lane, h);
    return;
  }
  const FusTile ft{fus + toff};
  if (next_wst) scattn_epilogue_h2<true>(o, active, ss, vecs, ft, out + toff, lane, h, next_bias, stat_next, pair, tile_raw * 32 + i < n_rows);
  else scattn_epilogue_h2<false>(o, active, ss, vecs, ft, out + toff, lane, h);
}

// k_scattn_merge_tile: the merge step of the small-grid layer (every item split, Fusion-2 branch as hidden-split partials) with
// ONE workgroup per query TILE instead of one per four: the grids it serves leave most CUs idle (B = 1, N = 5000: 40 row
// blocks), and its cost is a dependent chain - partial loads, then fc_message's 5 + PointCN's 4 weight blocks one after the
// other in each wave.  Here the four waves of a workgroup share ONE tile and split its 32-feature blocks: every wave merges
// the partials of its own block (a quarter of the loads), and the blocks of a level are multiplied in PARALLEL -
//   fc_message 128 -> 64 (2 blocks, waves 0 1) -> 64 (2 blocks) -> 128 (4 blocks, + the Fusion-2 branch) -> PointCN 128 (4 blocks)
// - four levels instead of twelve block steps, the level outputs exchanged as split-fp16 fragments through 16 KiB of LDS.
// Same arithmetic per element, same order of additions as k_scattn_merge: bit-identical results.  grid (tiles, B), block 256.
__global__ void __launch_bounds__(256, 1)
k_scattn_merge_tile(const float* __restrict__ part_o, const float* __restrict__ part_ml, const float* __restrict__ wst,
                    const float* __restrict__ vecs, float* __restrict__ out, int tiles, int ksplits,
                    const float* __restrict__ next_wst, const float* __restrict__ next_bias, const float* __restrict__ ff_part,
                    int ff_hs, const float* __restrict__ x1, const float* __restrict__ ff_b2, unsigned* __restrict__ stat_next,
                    int n_rows,                                                 // [r5] PvGuard::stat_next, the pairs' row count
                    const PairTab* __restrict__ ptab) {                         // [r5] ragged batches
  __shared__ __attribute__((aligned(16))) float lds[10 * kStageFloats];       // 9 weight stages | exchange
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, pair = blockIdx.y;
  if (ptab) {                                      // ragged batch: the pair's own rows; tiles beyond them have nothing to merge
    n_rows = ptab[pair].n;
    if (tile >= ((n_rows + 31) >> 5)) return;
  }
  const size_t n_tiles_all = (size_t)gridDim.y * tiles;
  const size_t pt0 = (size_t)pair * tiles + tile;
  const size_t toff = pt0 * (32 * C);
  float* const xch = lds + 9 * kStageFloats;
  f16x8* const xh = reinterpret_cast<f16x8*>(xch);
  StagesPreloaded ss;
  ss.init(lds, wave, lane, wst, 5, next_wst);
  auto stage = [&](int st) { return reinterpret_cast<const f16x8*>(lds + st * kStageFloats) + lane; };

  // everything this wave will need from global memory is requested now
  float b1[16], b2[16], b3[16], b4[16];
  if (wave < 2) { load_vec_block(b1, vecs, wave, h); load_vec_block(b2, vecs + 64, wave, h); }
  load_vec_block(b3, vecs + 128, wave, h);
  if (next_wst) load_vec_block(b4, next_bias, wave, h);
  float x2b[16];                                  // the Fusion-2 branch, block `wave`: x2 = sum_z part[z] 2^-8 + b2 + x1
  {
    float bb[16], xr[16], acc[16];
    load_vec_block(bb, ff_b2, wave, h);
    load_block_p32(acc, ff_part + toff, wave, lane);
    load_block_p32(xr, x1 + toff, wave, lane);
    for (int z = 1; z < ff_hs; ++z) {
      float pz[16];
      load_block_p32(pz, ff_part + toff + (size_t)z * n_tiles_all * (32 * C), wave, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += pz[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) x2b[r] = fmaf(acc[r], kH2Inv, bb[r]) + xr[r];
  }
  // merge of the key-split partials, feature block `wave`
  float o[16];
  {
    float mk[8], lk[8], ok[8][16];
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mk[k] = -INFINITY; lk[k] = 0.f;
      if (k < ksplits) {
        const float2 ml = *reinterpret_cast<const float2*>(part_ml + ((k * n_tiles_all + pt0) * 32 + i) * 2);
        mk[k] = ml.x; lk[k] = ml.y;
        load_block_p32(ok[k], part_o + (k * n_tiles_all + pt0) * (32 * C), wave, lane);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) M = fmaxf(M, mk[k]);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < ksplits) {
        const float wk = __builtin_amdgcn_exp2f(mk[k] - M);
        l = fmaf(lk[k], wk, l);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = fmaf(ok[k][r], wk, o[r]);
      }
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= inv;
  }
  store_block_h2(xch, wave, o, lane);             // the attention output as a 128-wide fragment image (16 KiB)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the nine weight stages
  __syncthreads();
  // ---- level 1: m1 = ReLU(Wa o + ba), 2 blocks ----
  if (wave < 2) {
    FragH2<8> ox;
#pragma unroll
    for (int s = 0; s < 8; ++s) { ox.h[s] = xh[(0 * 8 + s) * 64 + lane]; ox.l[s] = xh[(1 * 8 + s) * 64 + lane]; }
    f32x16 acc = zero16();
    mma_wx_h2<8>(acc, stage(wave), ox);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = relu_nan(fmaf(acc[r], kH2Inv, b1[r]));
    __syncthreads();                              // (both readers of `o` are done: m1 goes where it was)
    f16x8 hi, lo;
    split8h(&t[0], hi, lo); xh[(0 * 4 + 2 * wave) * 64 + lane] = hi; xh[(1 * 4 + 2 * wave) * 64 + lane] = lo;
    split8h(&t[8], hi, lo); xh[(0 * 4 + 2 * wave + 1) * 64 + lane] = hi; xh[(1 * 4 + 2 * wave + 1) * 64 + lane] = lo;
  } else {
    __syncthreads();
  }
  __syncthreads();
  // ---- level 2: m2 = ReLU(Wb m1 + bb), 2 blocks ----
  if (wave < 2) {
    FragH2<4> m1x;
#pragma unroll
    for (int s = 0; s < 4; ++s) { m1x.h[s] = xh[(0 * 4 + s) * 64 + lane]; m1x.l[s] = xh[(1 * 4 + s) * 64 + lane]; }
    f32x16 acc = zero16();
    mma_wx_h2<4>(acc, stage(2) + wave * (2 * 4 * 64), m1x);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = relu_nan(fmaf(acc[r], kH2Inv, b2[r]));
    __syncthreads();
    f16x8 hi, lo;
    split8h(&t[0], hi, lo); xh[(0 * 4 + 2 * wave) * 64 + lane] = hi; xh[(1 * 4 + 2 * wave) * 64 + lane] = lo;
    split8h(&t[8], hi, lo); xh[(0 * 4 + 2 * wave + 1) * 64 + lane] = hi; xh[(1 * 4 + 2 * wave + 1) * 64 + lane] = lo;
  } else {
    __syncthreads();
  }
  __syncthreads();
  // ---- level 3: feat = Wc m2 + bc + x2, 4 blocks ----
  float tt[16];
  {
    FragH2<4> m2x;
#pragma unroll
    for (int s = 0; s < 4; ++s) { m2x.h[s] = xh[(0 * 4 + s) * 64 + lane]; m2x.l[s] = xh[(1 * 4 + s) * 64 + lane]; }
    f32x16 acc = zero16();
    mma_wx_h2<4>(acc, stage(3 + (wave >> 1)) + (wave & 1) * (2 * 4 * 64), m2x);
#pragma unroll
    for (int r = 0; r < 16; ++r) tt[r] = fmaf(acc[r], kH2Inv, b3[r]) + x2b[r];
  }
  if (!next_wst) {                                // last layer: the block output itself
    store_block_p32(out + toff, wave, tt, lane);
    return;
  }
  __syncthreads();                                // every wave has read m2
  store_block_h2(xch, wave, tt, lane);
  __syncthreads();
  // ---- level 4: f_next = ReLU(Wp feat + bp), 4 blocks ----
  {
    FragH2<8> fx;
#pragma unroll
    for (int s = 0; s < 8; ++s) { fx.h[s] = xh[(0 * 8 + s) * 64 + lane]; fx.l[s] = xh[(1 * 8 + s) * 64 + lane]; }
    f32x16 acc = zero16();
    mma_wx_h2<8>(acc, stage(5 + wave), fx);
    float t[16], ssq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { t[r] = relu_nan(fmaf(acc[r], kH2Inv, b4[r])); ssq = fmaf(t[r], t[r], ssq); }
    store_block_p32(out + toff, wave, t, lane);
    // [r5] the next layer's "pv_fp8" statistic: a row's |f|^2 is the sum of the four waves' blocks - through the first weight
    // stage's LDS (level 1 is long done with it), added in wave order as the whole-item epilogue adds its blocks
    if (stat_next) {
      ssq = xhalf_sum(ssq);
      if (h == 0) lds[wave * 32 + i] = ssq;
      __syncthreads();
      if (wave == 0) {
        float s4 = ((lds[i] + lds[32 + i]) + lds[64 + i]) + lds[96 + i];
        pv_stat_raise(stat_next, pair, 0.5f * s4, tile * 32 + i < n_rows, lane);     // (both lane halves hold the row's sum: halves add up to it)
      }
    }
  }
}

// =========================================================================================
// k_ctx_prep: context side of a FusionLayer, once per (weight set, pair, token tile):
//   ctx' = LCPE(ctx) [PE] ; cn = LayerNorm_ctx(ctx') ; Kc = cn Wk^T ; Vc = cn Wv^T
//   output per tile: 4096 floats = Kc as P32 (K=64) | Vc as T image (D=64)
//   stages (4): Wk[2] | Wv[2]        vecs: w0|w1|w2|b (LCPE content taps) | gamma | beta
//   grid: (ceil(ttiles/4), B, L)   wst/vecs/out advance by set_stride per blockIdx.z
// =========================================================================================
template <bool PE>
__global__ void __launch_bounds__(256, 2)
k_ctx_prep(const float* __restrict__ ctx, const float* __restrict__ wst, const float* __restrict__ vecs,
           float* __restrict__ out, int T, int ttiles, int wst_stride, int vec_stride) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
  const int pair = blockIdx.y, set = blockIdx.z, nb = gridDim.y;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < ttiles;
  const int tile = active ? tile_raw : ttiles - 1;
  wst += (size_t)set * wst_stride;
  vecs += (size_t)set * vec_stride;
  const float* pair_base = ctx + (size_t)pair * ttiles * (32 * C);
  float* dst = out + (((size_t)set * nb + pair) * ttiles + tile) * kStageFloats;

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst, 4);
  ss.prime();

  float x[CF], cn[CF];
  if (PE) lcpe_frag(x, pair_base, tile * 32 + i, T, vecs, h);
  else load_frag_p32<CF>(x, pair_base + (size_t)tile * (32 * C), lane);
  layernorm_frag<CF>(cn, x, vecs + 4 * C, vecs + 5 * C, h);

#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<CF>(acc, lw, cn);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r];
    if (active) store_block_p32(dst, mb, t, lane);
  }
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_xw<CF>(acc, lw, cn);
    if (active) store_block_timg(dst + 32 * DH, db, acc, 0.f, lane);
  }
}

// =========================================================================================
// k_fusion_attn: query side of a FusionLayer up to the first residual:
//   x' = LCPE(x) [PE] ; q = LayerNorm(x') Wq'^T (Wq' pre-scaled by log2(e)/sqrt(DH)) ;
//   a = softmax(q Kc^T) Vc ; x1 = a Wo^T + bo + x'
//   stages: Wq'[2] | ctx tiles [ttiles] | Wo[2] (2 blocks of 32x64 each)
//   vecs: w0|w1|w2|b (LCPE q taps) | gamma | beta | bo
// =========================================================================================
template <bool PE>
__global__ void __launch_bounds__(256, 2)
k_fusion_attn(const float* __restrict__ xin, const float* __restrict__ ctx_img, const float* __restrict__ wst,
              const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const float* pair_base = xin + (size_t)pair * tiles * (32 * C);
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst, 2,
          ctx_img + (size_t)pair * ttiles * kStageFloats, ttiles, wst + 2 * kStageFloats, 2);
  ss.prime();

  float xp[CF];
  if (PE) lcpe_frag(xp, pair_base, tile * 32 + i, N, vecs, h);
  else load_frag_p32<CF>(xp, pair_base + (size_t)tile * (32 * C), lane);

  float qf[DHF];
  {
    float xn[CF];
    layernorm_frag<CF>(xn, xp, vecs + 4 * C, vecs + 5 * C, h);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float4* lw = ss.acquire();
      f32x16 acc = zero16();
      mma_wx<CF>(acc, lw, xn);
#pragma unroll
      for (int r = 0; r < 16; ++r) qf[16 * mb + r] = acc[r];
    }
  }

  f32x16 oacc[2];
  oacc[0] = zero16(); oacc[1] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  for (int t = 0; t < ttiles; ++t) {
    const float4* lk = ss.acquire();
    const float4* lv = lk + (32 * DH / 4);
    f32x16 s = zero16();
    mma_wx<DHF>(s, lk, qf);
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
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_new); ls += x[r]; }
    l_half = fmaf(l_half, alpha, ls);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 vv = lv[(db * 4 + q) * 64];
        oacc[db] = mfma32(vv.x, x[4 * q + 0], oacc[db]);
        oacc[db] = mfma32(vv.y, x[4 * q + 1], oacc[db]);
        oacc[db] = mfma32(vv.z, x[4 * q + 2], oacc[db]);
        oacc[db] = mfma32(vv.w, x[4 * q + 3], oacc[db]);
      }
    }
  }
  float o[DHF];
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[16 * db + r] = oacc[db][r] * inv;
  }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const float4* lw = ss.acquire();
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int mb = 2 * st + hb;
      f32x16 acc = zero16();
      mma_wx<DHF>(acc, lw + hb * (32 * DH / 4), o);
      float b[16], t[16];
      load_vec_block(b, vecs + 6 * C, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] + b[r] + xp[16 * mb + r];
      if (active) store_block_p32(x1_out + toff, mb, t, lane);
    }
  }
}

// =========================================================================================
// k_fusion_ff:  x2 = x1 + W2 ( (W1a xn + b1a) * gelu(W1g xn + b1g) ) + b2 ,  xn = LayerNorm(x1)
//   The 1024-wide hidden layer never leaves registers: per chunk of 32 hidden units the two
//   W1 products are 16 accumulator registers each, GEGLU is elementwise on them, and the result
//   is directly the B operand of the W2 product.
//   stages (48): for c in 0..15: W1a_c (32x128) | W1g_c (32x128) | W2_c (4 blocks of 32x32)
//   vecs: gamma | beta | b1a[512] | b1g[512] | b2[128]
// =========================================================================================
__global__ void __launch_bounds__(256, 2)
k_fusion_ff(const float* __restrict__ x1, const float* __restrict__ wst, const float* __restrict__ vecs,
            float* __restrict__ x2_out, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst, 3 * (FFH / 32));
  ss.prime();

  float xn[CF];
  {
    float x[CF];
    load_frag_p32<CF>(x, x1 + toff, lane);
    layernorm_frag<CF>(xn, x, vecs, vecs + C, h);
  }
  f32x16 y[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) y[mb] = zero16();
  const float* b1a = vecs + 2 * C;
  const float* b1g = vecs + 2 * C + FFH;

  for (int c = 0; c < FFH / 32; ++c) {
    float ga[16];
    {
      const float4* lw = ss.acquire();
      f32x16 acc = zero16();
      mma_wx<CF>(acc, lw, xn);
      float b[16];
      load_vec_block(b, b1a, c, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] = acc[r] + b[r];
    }
    {
      const float4* lw = ss.acquire();
      f32x16 acc = zero16();
      mma_wx<CF>(acc, lw, xn);
      float b[16];
      load_vec_block(b, b1g, c, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] *= gelu_erf(acc[r] + b[r]);
    }
    {
      const float4* lw = ss.acquire();
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) mma_wx<16>(y[mb], lw + mb * (32 * 32 / 4), ga);
    }
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    float b[16], xr[16], t[16];
    load_vec_block(b, vecs + 2 * C + 2 * FFH, mb, h);
    load_block_p32(xr, x1 + toff, mb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = y[mb][r] + b[r] + xr[r];
    if (active) store_block_p32(x2_out + toff, mb, t, lane);
  }
}

// =========================================================================================
// k_head: classifier 128->32 ReLU ->32 ReLU ->1 and row L2 normalisation.
//   stages (2): Wc1 (32x128) | Wc2 (32x32, padded)     vecs: b1[32] | b2[32] | w3[32] | b3
//   outputs (row-major, the layout the pose head and the caller consume):
//     logits [B,N], feat_n [B,N,128] (unit rows), feat [B,N,128] (optional, may be null)
// =========================================================================================
__global__ void __launch_bounds__(256, 2)
k_head(const float* __restrict__ feat_img, const float* __restrict__ wst, const float* __restrict__ vecs,
       float* __restrict__ logits, float* __restrict__ feat_n, float* __restrict__ feat_rm, int N, int tiles, int* status,
       const PairTab* __restrict__ ptab, const unsigned* __restrict__ pv_stat, const float* __restrict__ pv_thr2, int n_layers) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
  const int pair = blockIdx.y;
  // [r5] informational: a layer of this pair ran the three-product form of P V under the "pv_fp8" guard (GMF_STATUS_PV_GUARDED)
  if (pv_stat && status && blockIdx.x == 0 && threadIdx.x == 0) {
    bool tripped = false;
    for (int l = 0; l < n_layers; ++l) tripped |= !(__uint_as_float(pv_stat[((size_t)l * gridDim.y + pair) * kPvStatStride]) <= pv_thr2[l]);
    if (tripped) __hip_atomic_fetch_or(status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const size_t row0 = pair_row0(ptab, pair, N);            // ragged batch: the outputs are packed [sum n, ...]
  N = pair_rows(ptab, pair, N);
  const int tiles_p = (N + 31) >> 5;
  if ((int)blockIdx.x * kWavesPerWG >= tiles_p) return;
  const int tile_raw = blockIdx.x * kWavesPerWG + wave;
  const bool active = tile_raw < tiles_p;
  const int tile = active ? tile_raw : tiles_p - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);
  const int row = tile * 32 + i;

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWavesPerWG, lane, wst, 2);
  ss.prime();

  float x[CF];
  load_frag_p32<CF>(x, feat_img + toff, lane);
  float h1[16], h2[16];
  {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<CF>(acc, lw, x);
    float b[16];
    load_vec_block(b, vecs, 0, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = fmaxf(acc[r] + b[r], 0.f);
  }
  {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<16>(acc, lw, h1);
    float b[16];
    load_vec_block(b, vecs + 32, 0, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = fmaxf(acc[r] + b[r], 0.f);
  }
  float w3[16];
  load_vec_block(w3, vecs + 64, 0, h);
  float part = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) part = fmaf(h2[r], w3[r], part);
  const float logit = xhalf_sum(part) + vecs[96];

  float ss2 = 0.f;
#pragma unroll
  for (int k = 0; k < CF; ++k) ss2 = fmaf(x[k], x[k], ss2);
  const float nrm2 = xhalf_sum(ss2);
  // an activation beyond the fp16 range of the split operands (or a non-finite input) surfaces here as inf / NaN: the status
  // word says so; the logit is stored as it is, the row's unit features as zeros - the pose head derives neighbour INDICES
  // from them, and a library must not turn a bad input into an out-of-bounds gather
  const bool bad_row = not_finite(nrm2);
  flag_status(status, active && row < N && (not_finite(logit) || bad_row), 1);
  const float inv = 1.0f / fmaxf(sqrtf(nrm2), 1e-12f);

  if (active && row < N) {
    const size_t ro = row0 + row;
    if (h == 0) logits[ro] = logit;
    float4* pn = reinterpret_cast<float4*>(feat_n + ro * C) + h;
#pragma unroll
    for (int g = 0; g < CF / 4; ++g)
      pn[2 * g] = bad_row ? make_float4(0.f, 0.f, 0.f, 0.f)
                          : make_float4(x[4 * g] * inv, x[4 * g + 1] * inv, x[4 * g + 2] * inv, x[4 * g + 3] * inv);
    if (feat_rm) {
      float4* pr = reinterpret_cast<float4*>(feat_rm + ro * C) + h;
#pragma unroll
      for (int g = 0; g < CF / 4; ++g) pr[2 * g] = make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
    }
  }
}

// =========================================================================================
// k_seed_dist: feature-space distances of the seed rows to every correspondence,
//   dist[seed][j] = 2 - 2 <f_seed, f_j>   (models/common.py:64-66 restricted to the rows PointDSC.py:329 keeps).
//   One wave = 32 seeds (fragment gathered from the P32 image of the unit features), key tiles streamed
//   through LDS; [r3] D = mfma(A = key tile, B = seed fragment) (split-fp16 operands) puts the SEED on the lane and four
//   consecutive keys in four consecutive registers: a lane stores its 16 distances of a tile as four 16-byte stores (the
//   key-on-lane form issued sixteen 4-byte stores per lane and tile - the kernel was bound by issuing them, not by the HBM: 148
//   -> ~95 us).  The four stores of a lane fill one 128-byte line of the seed's row.  grid (ceil(S/128), B); dist [B, S, ld] with
//   ld = 32 * tiles (rows padded to whole tiles: every store is aligned and unconditional in the key index).
// =========================================================================================
__global__ void __launch_bounds__(256, 2)
k_seed_dist(const float* __restrict__ featn_img, const int* __restrict__ seeds, float* __restrict__ dist,
            int N, int tiles, int S, int chunks, const PairTab* __restrict__ ptab) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStageFloats];     // 4-slot ring: three key tiles in flight
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y / chunks, chunk = blockIdx.y - pair * chunks;   // key tiles are split over `chunks` workgroups
  // ragged batch: the pair's own rows / tiles / seeds; the image slot and the distance rows keep the strides of the largest pair
  const int Nmax = N, Smax = S;
  N = pair_rows(ptab, pair, N);
  S = ptab ? ptab[pair].S : S;
  const int tiles_p = (N + 31) >> 5;
  if ((int)blockIdx.x * kWavesPerWG * 32 >= S) return;        // (uniform per workgroup, before any barrier)
  const int per = (tiles_p + chunks - 1) / chunks;
  const int t0 = chunk * per, t1 = min(tiles_p, t0 + per);
  const int seed_base = (blockIdx.x * kWavesPerWG + wave) * 32;
  const float* pair_img = featn_img + (size_t)pair * tiles * (32 * C);
  const int my = seed_base + i;
  const int row = (my < S) ? seeds[(size_t)pair * Smax + my] : 0;
  if (t0 >= t1) return;                         // uniform over the workgroup
  // split-fp16 operands (3 products on the f16 MFMA, fp32-equivalent: mfma_core.hpp): 24 MFMAs of 32 cycles per key tile
  // instead of 64 of 64.  [r3] featn_img is the split-fp16 plane image k_pack_rows_h2 writes (16-byte unit ((plane * 8 + s) * 64 +
  // lane) of a tile = the 8 halves of k-step s for lane (h, i)): a key tile goes LDS -> MFMA operand with no conversion (every
  // wave used to split the tile's 64 values per lane again: ~100 vector instructions per tile), and a seed's fragment is the
  // 16 units of its row's lane slot.
  FragH2<8> sx;
  {
    const f16x8* rp = reinterpret_cast<const f16x8*>(pair_img + (size_t)(row >> 5) * kStageFloats) + h * 32 + (row & 31);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) { sx.h[s8] = rp[(0 * 8 + s8) * 64]; sx.l[s8] = rp[(1 * 8 + s8) * 64]; }
  }

  StageRing<4> ss;                              // (a tile is 24 MFMAs: one stage of look-ahead left its L2 round trip exposed)
  ss.init(lds, wave, lane, pair_img + (size_t)t0 * kStageFloats, t1 - t0);
  ss.prime();
  const int ld = tiles * 32;
  float* drow = dist + ((size_t)pair * Smax + seed_base + i) * ld + 4 * h;      // this lane's seed row, its half of each 8-key group
  for (int t = t0; t < t1; ++t) {
    const f16x8* lk = as_h2(ss.acquire());
    f32x16 acc = zero16();
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {            // (the products in the order of the key-on-lane form: s_l k_h, s_h k_l, s_h k_h)
      const f16x8 kh = lk[(0 * 8 + s8) * 64], kl = lk[(1 * 8 + s8) * 64];
      acc = mfma_h16(kh, sx.l[s8], acc);
      acc = mfma_h16(kl, sx.h[s8], acc);
      acc = mfma_h16(kh, sx.h[s8], acc);
    }
    if (my < S) {                               // register r = key 8 (r >> 2) + 4 h + (r & 3) of the tile, lane = seed
      float4* out = reinterpret_cast<float4*>(drow + (size_t)t * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        out[2 * q] = make_float4(2.0f - 2.0f * acc[4 * q], 2.0f - 2.0f * acc[4 * q + 1], 2.0f - 2.0f * acc[4 * q + 2], 2.0f - 2.0f * acc[4 * q + 3]);
    }
  }
}

// =========================================================================================
// layout conversion kernels (the drop-in boundary hands over row-major / channel-major tensors)
// =========================================================================================
// strided [B, n_rows, K] (element (b,r,k) at b*sb + r*sr + k*sk) -> P32 image [B, tiles, 32*K]; rows >= n_rows are 0
__global__ void k_pack_p32(const float* __restrict__ src, float* __restrict__ dst, int n_rows, int tiles, int K,
                           long sb, long sr, long sk, long total4, const PairTab* __restrict__ ptab) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int lane = idx & 63;
  const long gi = idx >> 6;
  const int kg = K / 8;
  const int g = gi % kg;
  const long bt = gi / kg;
  const int tile = bt % tiles;
  const long b = bt / tiles;
  const int row = tile * 32 + (lane & 31);
  const int k0 = 8 * g + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < pair_rows(ptab, (int)b, n_rows)) {      // ragged batch: pair b's rows start at row0 of the packed [sum n, K] tensor
    const float* p = src + (ptab ? (long)ptab[b].row0 * sr : b * sb) + (long)row * sr + (long)k0 * sk;
    v = make_float4(p[0], p[sk], p[2 * sk], p[3 * sk]);
  }
  reinterpret_cast<float4*>(dst)[idx] = v;
}

// row-major [B, n_rows, 128] (ragged: packed rows, PairTab) -> split-fp16 plane image [B, tiles, 4096 floats]: per tile the
// 16-byte unit ((plane * 8 + s) * 64 + lane) holds, for lane (h, i) = row 32 tile + i, the 8 halves of k-step s in fragment
// order (f = 8 s + j -> feature 32 (f >> 4) + 8 ((f & 15) >> 2) + 4 h + (f & 3)); plane 0 = fp16(x), plane 1 = fp16(x - hi).
// Rows >= n_rows are zero.  The operand image of k_seed_dist (the unit features of the pose head).
// [r5] copy_src / copy_dst: the first n_copy threads also copy one float each - the pose head's NMS keys start as a copy of the
// scores (launch_nms_keys with candidate splits), and a hipMemcpyAsync of those few KB is a launch of its own on small grids.
__global__ void k_pack_rows_h2(const float* __restrict__ src, float* __restrict__ dst, int n_rows, int tiles, long total,
                               const PairTab* __restrict__ ptab, const float* __restrict__ copy_src, float* __restrict__ copy_dst,
                               long n_copy) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one (tile, k-step, lane) per thread
  if (idx < n_copy) copy_dst[idx] = copy_src[idx];
  if (idx >= total) return;
  const int lane = idx & 63, s8 = (idx >> 6) & 7;
  const long bt = idx >> 9;
  const int tile = bt % tiles;
  const int b = (int)(bt / tiles);
  const int h = lane >> 5, row = tile * 32 + (lane & 31);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < pair_rows(ptab, b, n_rows)) {
    const float* p = src + (pair_row0(ptab, b, n_rows) + row) * C + 32 * (s8 >> 1) + 16 * (s8 & 1) + 4 * h;
    const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 8);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  }
  f16x8 hi, lo;
  split8h(v, hi, lo);
  f16x8* out = reinterpret_cast<f16x8*>(dst + bt * kStageFloats);
  out[(0 * 8 + s8) * 64 + lane] = hi;
  out[(1 * 8 + s8) * 64 + lane] = lo;
}

// P32 image -> strided [B, n_rows, K]
__global__ void k_unpack_p32(const float* __restrict__ src, float* __restrict__ dst, int n_rows, int tiles, int K,
                             long sb, long sr, long sk, long total4, int* status) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int lane = idx & 63;
  const long gi = idx >> 6;
  const int kg = K / 8;
  const int g = gi % kg;
  const long bt = gi / kg;
  const int tile = bt % tiles;
  const long b = bt / tiles;
  const int row = tile * 32 + (lane & 31);
  const int k0 = 8 * g + 4 * (lane >> 5);
  if (row < n_rows) {
    const float4 v = reinterpret_cast<const float4*>(src)[idx];
    float* p = dst + b * sb + (long)row * sr + (long)k0 * sk;
    p[0] = v.x; p[sk] = v.y; p[2 * sk] = v.z; p[3 * sk] = v.w;
    if (status && (not_finite(v.x) || not_finite(v.y) || not_finite(v.z) || not_finite(v.w)))
      __hip_atomic_fetch_or(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// src,tgt [B,N,3] -> pts8 [B, Npad, 8] (zero padded rows); pack_pts8_body: enc_common.hpp
__global__ void k_pack_pts8(const float* __restrict__ src, const float* __restrict__ tgt, float* __restrict__ dst,
                            int N, int Npad, long total, const PairTab* __restrict__ ptab, unsigned* __restrict__ zero_words,
                            int n_zero) {
  pack_pts8_body((long)blockIdx.x * blockDim.x + threadIdx.x, src, tgt, dst, N, Npad, total, ptab, zero_words, n_zero);
}

}  // namespace gmf

// -----------------------------------------------------------------------------------------

// -----------------------------------------------------------------------------------------
// host-side launchers (C++ linkage inside the library; the C ABI in gmf_api.cpp calls these)
// -----------------------------------------------------------------------------------------
#include "launchers.hpp"

namespace gmf {

// compute units of the current device (256 on MI355X); queried once per process and device
static int cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  static int cached[16] = {0};
  if (dev >= 0 && dev < 16 && cached[dev] > 0) return cached[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < 16) cached[dev] = n;
  return n;
}

static inline dim3 tile_grid(int tiles, int B, int sets = 1) { return dim3((tiles + kWavesPerWG - 1) / kWavesPerWG, B, sets); }

hipError_t launch_front(int mode, const float* in, const float* wst, const float* vecs, float* f, float* q, float* k,
                        float* v, int B, int N, int tiles, hipStream_t s) {
  const dim3 g = tile_grid(tiles, B);
  if (mode == 1) hipLaunchKernelGGL(k_front<1>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles);
  else if (mode == 2) hipLaunchKernelGGL(k_front<2>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles);
  else hipLaunchKernelGGL(k_front<0>, g, dim3(256), 0, s, in, wst, vecs, f, q, k, v, N, tiles);
  return hipGetLastError();
}

hipError_t launch_compat_build(const float* pts8, float* c_dense, int B, int N, int tiles, float sigma_d, int fmt, hipStream_t s,
                               const PairTab* ptab) {
  const dim3 grid(tiles, (tiles + 4 * kJPerWave - 1) / (4 * kJPerWave), B);
  const float inv = 1.0f / (sigma_d * sigma_d);
  if (fmt == 1) hipLaunchKernelGGL(k_compat_build<1>, grid, dim3(256), 0, s, pts8, c_dense, N, tiles, inv, ptab);
  else if (fmt == 2) hipLaunchKernelGGL(k_compat_build<2>, grid, dim3(256), 0, s, pts8, c_dense, N, tiles, inv, ptab);
  else hipLaunchKernelGGL(k_compat_build<0>, grid, dim3(256), 0, s, pts8, c_dense, N, tiles, inv, ptab);
  return hipGetLastError();
}

// fp32 images (P32 Q', K; T-image V), fp32 MFMA, c recomputed in-kernel: the single-stage entry point gmf_scattn_forward
// and the whole-encoder path when no split-fp16 weight images were given (or `scattn_variant` 0).
hipError_t launch_scattn_fp32(const float* q, const float* k, const float* v, const float* pts8, const float* fus,
                              const float* wst, const float* vecs, float* out, int B, int N, int tiles, float sigma_d,
                              hipStream_t s) {
  hipLaunchKernelGGL(k_scattn<false>, tile_grid(tiles, B), dim3(256), 0, s, q, k, v, pts8, fus, wst, vecs, out, N, tiles,
                     1.0f / (sigma_d * sigma_d));
  return hipGetLastError();
}

// Work split of the attention grid (attn_item): W items on `slots` resident workgroups (2 per CU).  n_full = items per XCD
// that run whole; the rest are split `ksplits` ways by key range (ksplits = 1: none).
void plan_attn_split(const Tuning& tune, int W, int tiles, int max_splits, int* n_full_out, int* ksplits_out) {
  const int slots = 2 * cu_count();
  const int per_xcd = (W >> 3) + ((W & 7) ? 1 : 0);
  int n_full = per_xcd, ksplits = 1;                                    // default: every item whole
  if (max_splits > 1 && tiles >= 8) {
    const int cap = std::min(max_splits, std::max(1, tiles / 4));
    if (tune.key_splits > 1) { n_full = 0; ksplits = std::min(tune.key_splits, cap); }             // forced: every item split
    else if (tune.key_splits == 0) {
      // small grid: ONE workgroup per CU - a workgroup alone on its CU runs its tiles almost twice as fast as two
      // co-resident ones (B = 1, N = 5000: 6 splits = 240 workgroups 1.64 ms per forward, 8 splits = 320 workgroups 1.75 ms)
      if (W < slots / 2) { n_full = 0; ksplits = std::min(std::max(2, (slots / 2) / W), cap); }
      else if (W < 3 * slots / 4 && tiles >= 64) { n_full = 0; ksplits = std::min(2, cap); }       // measured break-even
      else if (W > slots && tune.tail_split) {
        // large grid: whole rounds run whole; a last partial round of at most half the slots is split to fill them
        const int full = (W / slots) * slots, rest = W - full;
        if (rest > 0 && 2 * rest <= slots && tiles >= 16) { n_full = full / 8; ksplits = std::min(std::min(slots / rest, 4), cap); }
      }
    }
    if (ksplits <= 1) { n_full = per_xcd; ksplits = 1; }
  }
  *n_full_out = n_full;
  *ksplits_out = ksplits;
}

// Small grids, launches two and three of a layer (after k_small_front_fattn): key-split attention workgroups beside the
// hidden-split feed-forward workgroups, then the merge kernel (partials of both -> block output / next layer's f).
// Preconditions (checked by the caller with plan_attn_split / plan_ff_split): every attention item is split, ff_hs > 1.
hipError_t launch_small_attn_ff_merge(const float* q, const float* k, const float* v, const float* x1, const float* ff_wst,
                                      const float* ff_vecs, float* ff_part, int ff_hs, const float* tail_vecs, float* out, int B,
                                      int N, int tiles, int ksplits, hipStream_t s, const CompatCache* cc, bool tile_merge) {
  const int wpp = (tiles + 3) / 4, W = wpp * B;
  const int per_xcd = (W >> 3) + ((W & 7) ? 1 : 0);
  const int n_attn = 8 * per_xcd * ksplits, n_ff = W * ff_hs;
  if (cc->fmt == 2 && cc->v_scale)
    hipLaunchKernelGGL((k_small_attn_ff<2, true>), dim3(n_attn + n_ff), dim3(256), 0, s, n_attn, q, k, v, cc->tail_wst_h2, tail_vecs, N, tiles, wpp,
                       cc->dense, W, ksplits, cc->part_o, cc->part_ml, x1, ff_wst, ff_vecs, ff_part, B, ff_hs, cc->v_scale, cc->guard, cc->ptab);
  else if (cc->fmt == 2)
    hipLaunchKernelGGL(k_small_attn_ff<2>, dim3(n_attn + n_ff), dim3(256), 0, s, n_attn, q, k, v, cc->tail_wst_h2, tail_vecs, N, tiles, wpp,
                       cc->dense, W, ksplits, cc->part_o, cc->part_ml, x1, ff_wst, ff_vecs, ff_part, B, ff_hs, (const unsigned*)nullptr, PvGuard{}, cc->ptab);
  else if (cc->v_scale)
    hipLaunchKernelGGL((k_small_attn_ff<0, true>), dim3(n_attn + n_ff), dim3(256), 0, s, n_attn, q, k, v, cc->tail_wst_h2, tail_vecs, N, tiles, wpp,
                       cc->dense, W, ksplits, cc->part_o, cc->part_ml, x1, ff_wst, ff_vecs, ff_part, B, ff_hs, cc->v_scale, cc->guard, cc->ptab);
  else
    hipLaunchKernelGGL(k_small_attn_ff<0>, dim3(n_attn + n_ff), dim3(256), 0, s, n_attn, q, k, v, cc->tail_wst_h2, tail_vecs, N, tiles, wpp,
                       cc->dense, W, ksplits, cc->part_o, cc->part_ml, x1, ff_wst, ff_vecs, ff_part, B, ff_hs, (const unsigned*)nullptr, PvGuard{}, cc->ptab);
  if (tile_merge)
    hipLaunchKernelGGL(k_scattn_merge_tile, dim3(tiles, B), dim3(256), 0, s, cc->part_o, cc->part_ml, cc->tail_wst_h2, tail_vecs, out,
                       tiles, ksplits, cc->next_wst_h2, cc->next_bias, (const float*)ff_part, ff_hs, x1, ff_vecs + 2 * C + 2 * FFH,
                       cc->guard.stat_next, N, cc->ptab);
  else
    hipLaunchKernelGGL(k_scattn_merge, dim3(8 * per_xcd), dim3(256), 0, s, cc->part_o, cc->part_ml, x1, cc->tail_wst_h2, tail_vecs, out,
                       tiles, wpp, W, 0, ksplits, cc->next_wst_h2, cc->next_bias, (const float*)ff_part, ff_hs, x1,
                       ff_vecs + 2 * C + 2 * FFH, cc->guard.stat_next, N, cc->ptab);
  return hipGetLastError();
}

// Split-fp16 plane images of Q', K, V (k_front_h2 / k_linear_h2).  tune.scattn_variant:
//   18 (default) = k_scattn_h2p: c streamed from the compat cache, tile loop software-pipelined inside each wave, split-fp16
//                  fc_message epilogue; on small grids the keys of a query block are divided over several workgroups
//                  (KSPLIT form) and k_scattn_merge finishes.  Needs cc->dense and cc->tail_wst_h2; otherwise:
//   9            = k_scattn_h2: not pipelined, c from the cache when there is one, else recomputed per (i, j) from pts8
//                  (the fallback when the cache would not fit); fp32-MFMA fc_message epilogue (wst = fp32 images).
hipError_t launch_scattn_h2(const Tuning& tune, const float* q, const float* k, const float* v, const float* pts8,
                            const float* fus, const float* wst, const float* vecs, float* out, int B, int N, int tiles,
                            float sigma_d, hipStream_t s, const CompatCache* cc) {
  const float inv = 1.0f / (sigma_d * sigma_d);
  const int wpp = (tiles + 3) / 4;
  const dim3 grid4(wpp * B);
  const float* cd = (cc && tune.use_cache) ? cc->dense : nullptr;
  if (tune.scattn_variant == 18 && cd && cc->tail_wst_h2) {
    const int W = wpp * B;
    const int per_xcd = (W >> 3) + ((W & 7) ? 1 : 0);
    int n_full, ksplits;
    // ragged batch: planned on the SMALLEST pair's tiles (every pair then has at least four key tiles per split); either every item is
    // split - small grids: a handful of whole items walking all their keys alone left most of the chip idle (4 pairs x 1000: 1.50 ms
    // against 0.74 for the uniform batch) - or none is (the split tail of large grids assumes items of one length)
    plan_attn_split(tune, W, cc->ptab ? cc->min_tiles : tiles, cc->part_o ? cc->max_splits : 0, &n_full, &ksplits);
    if (cc->ptab && n_full != 0) { n_full = per_xcd; ksplits = 1; }
    const int max_tail = std::max(0, per_xcd - n_full);
    const dim3 grid(8 * (std::min(n_full, per_xcd) + max_tail * ksplits));
    if (cc->half && max_tail == 0)   // throughput numerics mode, whole items only: the three-tiles-in-flight form
      hipLaunchKernelGGL(k_scattn_fast, dim3(8 * per_xcd), dim3(256), 0, s, q, k, v, fus, cc->tail_wst_h2, vecs, out, N, tiles, wpp, cd, W,
                         cc->next_wst_h2, cc->next_bias);
    else if (cc->half)  // ... with a split tail: one fp16 product, c streamed as fp16 (the cache was built that way)
      hipLaunchKernelGGL((k_scattn_h2p<1, 1>), grid, dim3(256), 0, s, q, k, v, fus, cc->tail_wst_h2, vecs, out, N, tiles, wpp, cd, W,
                         n_full, ksplits, cc->part_o, cc->part_ml, cc->next_wst_h2, cc->next_bias, cc->ptab, (const unsigned*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, PvGuard{});
    else if (cc->fmt == 2 && cc->v_scale)   // parity arithmetic, c streamed as 16-bit fixed point; fp8 cross products of P V
      hipLaunchKernelGGL((k_scattn_h2p<3, 2, true>), grid, dim3(256), 0, s, q, k, v, fus, cc->tail_wst_h2, vecs, out, N, tiles, wpp, cd,
                         W, n_full, ksplits, cc->part_o, cc->part_ml, cc->next_wst_h2, cc->next_bias, cc->ptab, cc->v_scale, cc->qf_img, cc->qw_wst, cc->qw_bias, cc->guard);
    else if (cc->fmt == 2)
      hipLaunchKernelGGL((k_scattn_h2p<3, 2>), grid, dim3(256), 0, s, q, k, v, fus, cc->tail_wst_h2, vecs, out, N, tiles, wpp, cd,
                         W, n_full, ksplits, cc->part_o, cc->part_ml, cc->next_wst_h2, cc->next_bias, cc->ptab, (const unsigned*)nullptr, cc->qf_img, cc->qw_wst, cc->qw_bias, PvGuard{});
    else if (cc->v_scale)    // the default: V image with e4m3 cross planes (k_linear_h2 wrote it that way)
      hipLaunchKernelGGL((k_scattn_h2p<3, 0, true>), grid, dim3(256), 0, s, q, k, v, fus, cc->tail_wst_h2, vecs, out, N, tiles, wpp, cd,
                         W, n_full, ksplits, cc->part_o, cc->part_ml, cc->next_wst_h2, cc->next_bias, cc->ptab, cc->v_scale, cc->qf_img, cc->qw_wst, cc->qw_bias, cc->guard);
    else
      hipLaunchKernelGGL((k_scattn_h2p<3, 0>), grid, dim3(256), 0, s, q, k, v, fus, cc->tail_wst_h2, vecs, out, N, tiles, wpp, cd,
                         W, n_full, ksplits, cc->part_o, cc->part_ml, cc->next_wst_h2, cc->next_bias, cc->ptab, (const unsigned*)nullptr, cc->qf_img, cc->qw_wst, cc->qw_bias, PvGuard{});
    if (max_tail > 0)
      hipLaunchKernelGGL(k_scattn_merge, dim3(8 * max_tail), dim3(256), 0, s, cc->part_o, cc->part_ml, fus, cc->tail_wst_h2, vecs, out,
                         tiles, wpp, W, n_full, ksplits, cc->next_wst_h2, cc->next_bias, (const float*)nullptr, 0,
                         (const float*)nullptr, (const float*)nullptr, cc->guard.stat_next, N, cc->ptab);
  }
  else if (cd) hipLaunchKernelGGL(k_scattn_h2<true>, grid4, dim3(256), 0, s, q, k, v, pts8, fus, wst, vecs, out, N, tiles, inv, wpp, cd);
  else hipLaunchKernelGGL(k_scattn_h2<false>, grid4, dim3(256), 0, s, q, k, v, pts8, fus, wst, vecs, out, N, tiles, inv, wpp, cd);
  return hipGetLastError();
}

hipError_t launch_scattn_dense(const float* q, const float* k, const float* v, const float* compat, const float* fus,
                               const float* wst, const float* vecs, float* out, int B, int N, int tiles, hipStream_t s) {
  hipLaunchKernelGGL(k_scattn<true>, tile_grid(tiles, B), dim3(256), 0, s, q, k, v, compat, fus, wst, vecs, out, N, tiles, 0.f);
  return hipGetLastError();
}

hipError_t launch_ctx_prep(bool pe, const float* ctx, const float* wst, const float* vecs, float* out, int B, int T,
                           int ttiles, int sets, int wst_stride, int vec_stride, hipStream_t s) {
  if (pe) hipLaunchKernelGGL(k_ctx_prep<true>, tile_grid(ttiles, B, sets), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles, wst_stride, vec_stride);
  else hipLaunchKernelGGL(k_ctx_prep<false>, tile_grid(ttiles, B, sets), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles, wst_stride, vec_stride);
  return hipGetLastError();
}

hipError_t launch_fusion_attn(bool pe, const float* x, const float* ctx_img, const float* wst, const float* vecs,
                              float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s) {
  if (pe) hipLaunchKernelGGL(k_fusion_attn<true>, tile_grid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  else hipLaunchKernelGGL(k_fusion_attn<false>, tile_grid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  return hipGetLastError();
}

hipError_t launch_fusion_ff(const float* x1, const float* wst, const float* vecs, float* x2, int B, int tiles, hipStream_t s) {
  hipLaunchKernelGGL(k_fusion_ff, tile_grid(tiles, B), dim3(256), 0, s, x1, wst, vecs, x2, tiles);
  return hipGetLastError();
}

hipError_t launch_head(const float* feat_img, const float* wst, const float* vecs, float* logits, float* feat_n,
                       float* feat_rm, int B, int N, int tiles, hipStream_t s, int* status, const PairTab* ptab, const unsigned* pv_stat,
                       const float* pv_thr2, int n_layers) {
  hipLaunchKernelGGL(k_head, tile_grid(tiles, B), dim3(256), 0, s, feat_img, wst, vecs, logits, feat_n, feat_rm, N, tiles, status, ptab,
                     pv_stat, pv_thr2, n_layers);
  return hipGetLastError();
}

hipError_t launch_seed_dist(const float* featn_img, const int* seeds, float* dist, int B, int N, int S, hipStream_t s, const PairTab* ptab) {
  const int tiles = (N + 31) / 32;
  const int sblocks = (S + 127) / 128;
  int chunks = 1;                                  // enough workgroups to fill 256 CUs twice over
  while (chunks < 16 && sblocks * B * chunks < 1024 && tiles / (2 * chunks) >= 4) chunks *= 2;
  hipLaunchKernelGGL(k_seed_dist, dim3(sblocks, B * chunks), dim3(256), 0, s, featn_img, seeds, dist, N, tiles, S, chunks, ptab);
  return hipGetLastError();
}

hipError_t launch_pack_p32(const float* src, float* dst, int B, int n_rows, int K, long sb, long sr, long sk, hipStream_t s, const PairTab* ptab) {
  const int tiles = (n_rows + 31) / 32;
  const long total4 = (long)B * tiles * (K / 8) * 64;
  hipLaunchKernelGGL(k_pack_p32, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, src, dst, n_rows, tiles, K, sb, sr, sk, total4, ptab);
  return hipGetLastError();
}

hipError_t launch_pack_rows_h2(const float* src, float* dst, int B, int n_rows, hipStream_t s, const PairTab* ptab, const float* copy_src,
                               float* copy_dst, long n_copy) {
  const int tiles = (n_rows + 31) / 32;
  const long total = (long)B * tiles * 8 * 64;
  if (!copy_src || !copy_dst) n_copy = 0;
  if (n_copy > total) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pack_rows_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, n_rows, tiles, total, ptab, copy_src,
                     copy_dst, n_copy);
  return hipGetLastError();
}

hipError_t launch_unpack_p32(const float* src, float* dst, int B, int n_rows, int K, long sb, long sr, long sk, hipStream_t s, int* status) {
  const int tiles = (n_rows + 31) / 32;
  const long total4 = (long)B * tiles * (K / 8) * 64;
  hipLaunchKernelGGL(k_unpack_p32, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, src, dst, n_rows, tiles, K, sb, sr, sk, total4, status);
  return hipGetLastError();
}

hipError_t launch_pack_pts8(const float* src, const float* tgt, float* dst, int B, int N, hipStream_t s, const PairTab* ptab,
                            unsigned* zero_words, int n_zero) {
  const int Npad = ((N + 31) / 32) * 32;
  const long total = (long)B * Npad;
  hipLaunchKernelGGL(k_pack_pts8, dim3((unsigned)((std::max(total, (long)n_zero) + 255) / 256)), dim3(256), 0, s, src, tgt, dst, N, Npad, total, ptab,
                     zero_words, n_zero);
  return hipGetLastError();
}

}  // namespace gmf
