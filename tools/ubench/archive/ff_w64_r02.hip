// Round-2 experiment, not part of the library: the GEGLU feed-forward with TWO row tiles per wave (one wave per SIMD, 512
// registers, 256 rows per workgroup, one workgroup per CU).  Same results as k_fusion_ff_h2p (logits agree to 3e-5).
// Measured in one process at 32 pairs x 5000 (tools/ab_w64.py, since removed): +0.44 ms per step, i.e. ~238 us per launch
// against 201 us - and 640 workgroups on 256 CUs are 2.5 rounds, so even perfectly balanced it would only tie.  The
// compiler-scheduled single wave does not hide its ~7.5 vector instructions per MFMA (GELU 16, splits, accumulator <-> VGPR
// copies: 790 v_accvgpr moves per two chunks) the way two co-resident waves do.  Fragment of encoder_h2.hip:
// =========================================================================================
// Two row tiles per wave ("w64"): one wave per SIMD with the whole 512-register file, 4 waves = 256 rows per workgroup, one
// workgroup per CU.  Every weight fragment read from LDS feeds two MFMAs (tile A, tile B), a 16 KiB stage serves 256 rows
// instead of 128 (half the L2 -> LDS traffic, half the LDS-DMA pieces, LDS reads and stage barriers per MFMA), and the four
// waves of a workgroup - alone on their SIMDs - stay in step, so the stage barriers cost little.  The two tiles are
// independent instruction streams inside the wave: tile B's MFMAs fill the dependency gaps of tile A's.
// ff_chunks_w64 is ff_chunks for NT = 2 tiles.
// =========================================================================================
GMF_DEVINL void ff_chunks_w64(const FragH2<8> (&nx)[2], f32x16 (&y)[2][4], float* lds, const float* __restrict__ wst,
                              const float* __restrict__ b1a, const float* __restrict__ b1g, const int wave, const int lane,
                              const int h) {
  constexpr int NB = 4, NCH = FFH / 32;
  int n_issued = 0, n_used = 0;
  auto blob_stage = [&](int n) {
    n = min(n, 3 * NCH - 1);
    if (n < 2) return n;
    if (n == 3 * NCH - 1) return n;
    const int m = n - 2, c = m / 3, k = m - 3 * c;
    return (k == 2) ? 3 * c + 2 : 3 * c + 3 + k;
  };
  auto issue_one = [&]() {
    const float* g = wst + (size_t)blob_stage(n_issued) * kStageFloats;
    float* dst = lds + (n_issued & (NB - 1)) * kStageFloats;
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_piece_1k(g + (wave + 4 * q) * 256, dst + (wave + 4 * q) * 256, lane);
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
  f32x16 a0[2], g0[2], a1[2], g1[2];
  a0[0] = bias_acc(b1a, 0); g0[0] = bias_acc(b1g, 0);
  a0[1] = a0[0]; g0[1] = g0[0];
  {
    const f16x8* lw = acquire();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f16x8 wh = lw[(0 * 8 + s) * 64], wl = lw[(1 * 8 + s) * 64];
      mma3(a0[0], wh, wl, nx[0].h[s], nx[0].l[s]);
      mma3(a0[1], wh, wl, nx[1].h[s], nx[1].l[s]);
    }
    lw = acquire();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f16x8 wh = lw[(0 * 8 + s) * 64], wl = lw[(1 * 8 + s) * 64];
      mma3(g0[0], wh, wl, nx[0].h[s], nx[0].l[s]);
      mma3(g0[1], wh, wl, nx[1].h[s], nx[1].l[s]);
    }
  }
  auto chunk = [&](const int c, f32x16 (&a_cur)[2], const f32x16 (&g_cur)[2], f32x16 (&a_nxt)[2], f32x16 (&g_nxt)[2],
                   const bool has_next) {
    FragH2<2> gx[2];
    if (has_next) {
      a_nxt[0] = bias_acc(b1a, c + 1); g_nxt[0] = bias_acc(b1g, c + 1);
      a_nxt[1] = a_nxt[0]; g_nxt[1] = g_nxt[0];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f16x8* lw = acquire();
        f16x8 wh = lw[0], wl = lw[8 * 64];
        f16x8 wh_n = wh, wl_n = wl;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int u = 8 * half + s;
          if (s < 7) { wh_n = lw[(0 * 8 + s + 1) * 64]; wl_n = lw[(1 * 8 + s + 1) * 64]; }
          f32x16& tA = (half == 0) ? a_nxt[0] : g_nxt[0];
          f32x16& tB = (half == 0) ? a_nxt[1] : g_nxt[1];
          tA = mfma_h16(wl, nx[0].h[s], tA);
          tB = mfma_h16(wl, nx[1].h[s], tB);
          tA = mfma_h16(wh, nx[0].l[s], tA);
          tB = mfma_h16(wh, nx[1].l[s], tB);
          tA = mfma_h16(wh, nx[0].h[s], tA);
          tB = mfma_h16(wh, nx[1].h[s], tB);
          wh = wh_n; wl = wl_n;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            a_cur[t][u] *= gelu_erf_1r(g_cur[t][u]);
            if (half == 1 && (s & 1)) { const int j = s - 1; split2h(a_cur[t][j], a_cur[t][j + 1], gx[t].h[0], gx[t].l[0], j); }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a_cur[t][u] *= gelu_erf_1r(g_cur[t][u]);
#pragma unroll
        for (int j = 0; j < 8; j += 2) split2h(a_cur[t][j], a_cur[t][j + 1], gx[t].h[0], gx[t].l[0], j);
      }
    }
    {
      const f16x8* lw = acquire();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const f16x8* lb = lw + mb * (2 * 2 * 64);
          const f16x8 wh = lb[(0 * 2 + s) * 64], wl = lb[(1 * 2 + s) * 64];
          y[0][mb] = mfma_h16(wl, gx[0].h[s], y[0][mb]);
          y[1][mb] = mfma_h16(wl, gx[1].h[s], y[1][mb]);
          y[0][mb] = mfma_h16(wh, gx[0].l[s], y[0][mb]);
          y[1][mb] = mfma_h16(wh, gx[1].l[s], y[1][mb]);
          y[0][mb] = mfma_h16(wh, gx[0].h[s], y[0][mb]);
          y[1][mb] = mfma_h16(wh, gx[1].h[s], y[1][mb]);
          if (s == 0) {
            const int j = 2 * mb;
#pragma unroll
            for (int t = 0; t < 2; ++t) split2h(a_cur[t][8 + j], a_cur[t][8 + j + 1], gx[t].h[1], gx[t].l[1], j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  for (int c = 0; c + 2 < NCH; c += 2) {
    chunk(c, a0, g0, a1, g1, true);
    chunk(c + 1, a1, g1, a0, g0, true);
  }
  chunk(NCH - 2, a0, g0, a1, g1, true);
  chunk(NCH - 1, a1, g1, a0, g0, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// k_fusion_ff_w64: the feed-forward of k_fusion_ff_h2p (same blobs, same results) in the two-tiles-per-wave form.
// grid (ceil(tiles / 8), B), block 256, one workgroup per CU.
__global__ void __launch_bounds__(256, 1)
k_fusion_ff_w64(const float* __restrict__ x1, const float* __restrict__ wst, const float* __restrict__ vecs,
                float* __restrict__ x2_out, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStageFloats];
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  int tile[2]; bool act[2]; size_t toff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int raw = blockIdx.x * 8 + 2 * wave + t;
    act[t] = raw < tiles;
    tile[t] = act[t] ? raw : tiles - 1;
    toff[t] = ((size_t)pair * tiles + tile[t]) * (32 * C);
  }
  FragH2<8> nx[2];
  f32x16 y[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float x[CF], xn[CF];
    load_frag_p32<CF>(x, x1 + toff[t], lane);
    layernorm_frag<CF>(xn, x, vecs, vecs + C, h);
    nx[t].set(xn);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float b[16];
      load_vec_block(b, vecs + 2 * C + 2 * FFH, mb, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) y[t][mb][r] = (x[16 * mb + r] + b[r]) * 256.0f;
    }
  }
  ff_chunks_w64(nx, y, lds, wst, vecs + 2 * C, vecs + 2 * C + FFH, wave, lane, h);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = y[t][mb][r] * kH2Inv;
      if (act[t]) store_block_p32(x2_out + toff[t], mb, o, lane);
    }
}

