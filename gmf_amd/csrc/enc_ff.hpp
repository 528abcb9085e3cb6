// The GEGLU feed-forward of a FusionLayer on the f16 MFMA with split-fp16 operands, as device functions shared by
// k_fusion_ff_h2p / k_linear_h2 (encoder_h2.hip) and the small-grid layer kernels (encoder_kernels.hip).
#pragma once
#include "enc_common.hpp"

namespace gmf {

// The GEGLU chunk loop of the feed-forward, software-pipelined inside the wave (shared by k_fusion_ff_h2p and k_linear_h2):
//   y[mb] += W2[:, chunk] (value_c * GELU(gate_c)) for the hidden chunks [c_begin, c_begin + NCH), value / gate = W1 nx + b1.
// `lds` is a ring of NB = 4 stages owned by the workgroup (the caller has made sure no wave still reads it); on return
// every DMA piece this wave issued has landed.  Stage order A0 G0 | A1 G1 W2_0 | ... addressed into the unchanged blob
// (16 x (A | G | W2)); past the end the last stage is re-fetched into a free slot instead of branching.
template <int NP = 3>
GMF_DEVINL void ff_chunks(const FragH2<8>& nx, f32x16 (&y)[4], float* lds, const float* __restrict__ wst,
                          const float* __restrict__ b1a, const float* __restrict__ b1g, const int wave, const int lane,
                          const int h, const int c_begin, const int NCH) {
  constexpr int NB = 4;
  // stage n of the consumption order -> stage index in the blob (16 x (A | G | W2))
  int n_issued = 0, n_used = 0;
  auto blob_stage = [&](int n) {
    n = min(n, 3 * NCH - 1);
    if (n < 2) return 3 * c_begin + n;
    if (n == 3 * NCH - 1) return 3 * c_begin + n;
    const int m = n - 2, c = m / 3, k = m - 3 * c;
    return 3 * c_begin + ((k == 2) ? 3 * c + 2 : 3 * c + 3 + k);
  };
  auto issue_one = [&]() {
    const float* g = wst + (size_t)blob_stage(n_issued) * kStageFloats;
    float* dst = lds + (n_issued & (NB - 1)) * kStageFloats;
    dma_4k_s(g + wave * 1024, dst + wave * 1024, (unsigned)lane * 16u);   // (statement form, one M0 setup per stage: see StageRing)
    ++n_issued;
  };
  auto acquire = [&]() {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but this wave's pieces of the 2 younger stages have landed
    __syncthreads();
    const f16x8* cur = reinterpret_cast<const f16x8*>(lds + (n_used & (NB - 1)) * kStageFloats) + lane;
    ++n_used;
    issue_one();
    return cur;
  };
  issue_one(); issue_one(); issue_one();

  auto bias_acc = [&](const float* bvec, int c) {
    float b[16];
    load_vec_block(b, bvec, c, h);
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = b[r];
    return a;
  };

  f32x16 a0 = bias_acc(b1a, c_begin), g0 = bias_acc(b1g, c_begin), a1, g1;
  {
    const f16x8* lw = acquire();
    mma_wx_h2n<8, NP>(a0, lw, nx);
    lw = acquire();
    mma_wx_h2n<8, NP>(g0, lw, nx);
  }
  // chunk c: gated values from (a_cur, g_cur); W1 of chunk c+1 accumulates into (a_nxt, g_nxt) meanwhile
  auto chunk = [&](const int c, f32x16& a_cur, const f32x16& g_cur, f32x16& a_nxt, f32x16& g_nxt, const bool has_next) {
    FragH2<2> gx;
    if (has_next) {
      a_nxt = bias_acc(b1a, c + 1);
      g_nxt = bias_acc(b1g, c + 1);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f16x8* lw = acquire();
        f16x8 wh = lw[0], wl = wh;
        if (NP == 3) wl = lw[8 * 64];
        f16x8 wh_n = wh, wl_n = wl;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int u = 8 * half + s;
          if (s < 7) { wh_n = lw[(0 * 8 + s + 1) * 64]; if (NP == 3) wl_n = lw[(1 * 8 + s + 1) * 64]; }
          if (half == 0) mma_np<NP>(a_nxt, wh, wl, nx.h[s], nx.l[s]);
          else mma_np<NP>(g_nxt, wh, wl, nx.h[s], nx.l[s]);
          wh = wh_n; wl = wl_n;
          a_cur[u] *= gelu_erf_1r(g_cur[u]);
          if (half == 1 && (s & 1)) { const int j = s - 1; split2h(a_cur[j], a_cur[j + 1], gx.h[0], gx.l[0], j); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) a_cur[u] *= gelu_erf_1r(g_cur[u]);
#pragma unroll
      for (int j = 0; j < 8; j += 2) split2h(a_cur[j], a_cur[j + 1], gx.h[0], gx.l[0], j);
    }
    {
      const f16x8* lw = acquire();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const f16x8* lb = lw + mb * (2 * 2 * 64);
          if (NP == 3) mma3(y[mb], lb[(0 * 2 + s) * 64], lb[(1 * 2 + s) * 64], gx.h[s], gx.l[s]);
          else y[mb] = mfma_h16(lb[(0 * 2 + s) * 64], gx.h[s], y[mb]);
          if (s == 0) { const int j = 2 * mb; split2h(a_cur[8 + j], a_cur[8 + j + 1], gx.h[1], gx.l[1], j); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  for (int c = c_begin; c + 2 < c_begin + NCH; c += 2) {
    chunk(c, a0, g0, a1, g1, true);
    chunk(c + 1, a1, g1, a0, g0, true);
  }
  chunk(c_begin + NCH - 2, a0, g0, a1, g1, true);
  chunk(c_begin + NCH - 1, a1, g1, a0, g0, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-fetched tail stages
}

// (bx, pair, n_pairs): row block, pair, pairs in the launch; (z, hs): this workgroup's hidden split and their number
// (SPLIT: workgroup z handles the hidden chunks [z, z+1) * 16 / hs and writes its partial Linear-2 output - no bias, no
// residual - to part[z]; k_ff_reduce or the attention merge kernel adds the partials in index order).
template <bool SPLIT>
GMF_DEVINL void fusion_ff_h2p_body(float* lds, const int bx, const int pair, const int n_pairs, const int z, const int hs,
                                   const float* __restrict__ x1, const float* __restrict__ wst, const float* __restrict__ vecs,
                                   float* __restrict__ x2_out, int tiles, float* __restrict__ part) {
  // (compile-time trip counts for the common un-split form: a run-time chunk count costs it 9 %)
  const int HS = SPLIT ? hs : 1, NCH = SPLIT ? (FFH / 32) / HS : FFH / 32, c_begin = SPLIT ? z * NCH : 0;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_raw = bx * kWavesPerWG + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);

  FragH2<8> nx;
  {
    float x[CF], xn[CF];
    load_frag_p32<CF>(x, x1 + toff, lane);
    layernorm_frag<CF>(xn, x, vecs, vecs + C, h);
    nx.set(xn);
  }
  f32x16 y[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) y[mb] = zero16();
  ff_chunks(nx, y, lds, wst, vecs + 2 * C, vecs + 2 * C + FFH, wave, lane, h, c_begin, NCH);
  if (HS > 1) {
    float* pt = part + ((size_t)z * n_pairs * tiles + (size_t)pair * tiles + tile) * (32 * C);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = y[mb][r];
      if (active) store_block_p32(pt, mb, t, lane);
    }
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    float b[16], xr[16], t[16];
    load_vec_block(b, vecs + 2 * C + 2 * FFH, mb, h);
    load_block_p32(xr, x1 + toff, mb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(y[mb][r], kH2Inv, b[r]) + xr[r];
    if (active) store_block_p32(x2_out + toff, mb, t, lane);
  }
}


}  // namespace gmf
