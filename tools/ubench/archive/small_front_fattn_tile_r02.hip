// ARCHIVED (round 2, measured and not shipped): the first launch of the small-grid layer with one workgroup per 32-row tile,
// its four waves splitting the output blocks and the context tiles, weights loaded straight into registers.  Correct (logits
// within 1.9e-5 of the role kernel's, Q'/K/V bit-identical) but no faster: 19.2-20.9 us against 19.0 us per launch at B = 1,
// N = 5000.  Stamps inside it (s_memtime between scheduling barriers) show why: issuing the prologue's loads takes 5.7 us of
// its 17.5 - 157 workgroups each pull all 368 KiB of the layer's weight and context blocks through the L2 for 32 rows of work
// (58 MB per launch, ~5.6 TB/s while it lasts); LCPE 2.7 us and LayerNorm + to_q 2.2 us on the critical waves; each Q'/K/V
// block 0.7 us.  The role kernel (k_small_front_fattn, encoder_h2.hip) reads a quarter of that per row.  The merge step's
// per-tile form (k_scattn_merge_tile, 144 KiB of weights per tile) is on the right side of the same trade and ships.
// (Excerpt of encoder_h2.hip at that commit; not compiled.)

// k_small_front_fattn_tile: the first launch of the small-grid layer with ONE workgroup per 32-row TILE (grids of up to 256
// tiles: B = 1 up to N = 8192).  k_small_front_fattn gives a row block of four tiles to four workgroups by ROLE (Q' | K | V |
// cross-attention), each wave walking a chain of 4 .. 11 dependent weight stages; here the four waves of a workgroup share one
// tile and split every level of the work between them:
//   Q', K, V          12 output blocks of 32: wave w multiplies block w of each
//   cross-attention   LCPE in every wave (cheap), LayerNorm + to_q's two blocks in waves 0 and 1, the context tiles dealt
//                     round-robin to the waves (each keeps its own running maximum / sum / accumulators; merged through the
//                     LDS like key-split partials), to_out's four blocks one per wave
// No wave shares a weight block with another, so nothing is staged through the LDS: a wave loads its own blocks straight
// into registers (a 32 x 128 block is 16 lane-linear 16-byte loads per lane), the next block's loads issued before the current
// block is multiplied - no stage barriers; three workgroup barriers in all (vectors / halo visible, q exchanged, partials
// exchanged).  Q', K, V are bit-identical to the role kernel's; the cross-attention adds its context tiles in another order
// (fp32 rounding).   grid (tiles, B), block 256.
__global__ void __launch_bounds__(256, 1)
k_small_front_fattn_tile(const float* __restrict__ f_in, const float* __restrict__ front_wst, const float* __restrict__ front_vec,
                         const float* __restrict__ ctx_img, const float* __restrict__ attn_wst, const float* __restrict__ attn_vec,
                         float* __restrict__ q_out, float* __restrict__ k_out, float* __restrict__ v_out, float* __restrict__ x1_out,
                         int N, int tiles, int T, int ttiles) {
  // LDS: q exchange (8 KiB) | softmax partials [wave][34][64] (34 KiB) | attention vectors (3.5 KiB) | LCPE halo rows (1 KiB)
  __shared__ __attribute__((aligned(16))) float lds[2048 + 4 * 34 * 64 + 7 * C + 2 * C];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, pair = blockIdx.y;
  const float* pair_base = f_in + (size_t)pair * tiles * (32 * C);
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * C);
  float* const xq = lds;
  float* const px = xq + 2048;
  float* const lvec = px + 4 * 34 * 64;
  float* const halo = lvec + 7 * C;
  const float* ctx_pair = ctx_img + (size_t)pair * ttiles * kStageFloats;

  // NF 16-byte fragments of a lane-linear image: unit u of `g` for this lane
  auto load_frags = [&](f16x8 (&dst)[16], const float* g, auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    const f16x8* p = reinterpret_cast<const f16x8*>(g) + lane;
#pragma unroll
    for (int u = 0; u < n; ++u) dst[u] = p[u * 64];
  };
  const std::integral_constant<int, 16> k16;
  const std::integral_constant<int, 8> k8;
  f16x8 wa[16], wb[16];                            // two weight / context blocks in flight ([plane][step] as in the stage images)

  // The blocks this wave loads LATER are touched now, one dword per 128-byte line: between two layers the compat stream has
  // swept the L2, so a block's first load is a trip to HBM (~2.5 us, measured by early-exit builds: each 24-MFMA block cost one
  // such trip) - after the touch it is an L2 hit by the time it is wanted.
  // (each touch goes to a variable of its own, summed at the kernel's END: an accumulation here would wait for every load)
  float tv[9];
  {
    const float* gk = front_wst + (size_t)(8 + (wave < 2 ? wave : 0)) * kStageFloats;     // Wk block (waves 2, 3 load theirs at once)
    const float* gv = front_wst + (size_t)(12 + wave) * kStageFloats;                     // Wv block
    const float* c0 = ctx_pair + (size_t)min(wave, ttiles - 1) * kStageFloats;            // context tiles wave, wave + 4 (clamped)
    const float* c1 = ctx_pair + (size_t)min(wave + 4, ttiles - 1) * kStageFloats;
    const float* go = attn_wst + (size_t)(2 + (wave >> 1)) * kStageFloats + (wave & 1) * 2048;   // Wo block (64 lines)
    tv[0] = gk[lane * 32]; tv[1] = gk[(lane + 64) * 32];
    tv[2] = gv[lane * 32]; tv[3] = gv[(lane + 64) * 32];
    tv[4] = c0[lane * 32]; tv[5] = c0[(lane + 64) * 32];
    tv[6] = c1[lane * 32]; tv[7] = c1[(lane + 64) * 32];
    tv[8] = go[lane * 32];
  }
  // ---- requests of the prologue ----
  dma_vec(attn_vec, lvec, 7 * C, wave, kWavesPerWG, lane);
  if (wave == 0) LcpeHalo<CF>::issue(pair_base, tile, tiles, halo, lane);
  float xp[CF];
  load_frag_p32<CF>(xp, f_in + toff, lane);
  float bq[16], bk[16];
  load_vec_block(bq, front_vec + 1 * C, wave, h);
  load_vec_block(bk, front_vec + 2 * C, wave, h);
  const float bv = front_vec[3 * C + 32 * wave + i];
  load_frags(wa, front_wst + (size_t)(4 + wave) * kStageFloats, k16);                 // Wq' block
  if (wave < 2) load_frags(wb, attn_wst + (size_t)wave * kStageFloats, k16);           // Wq'' block (waves 0, 1)
  else load_frags(wb, front_wst + (size_t)(8 + wave) * kStageFloats, k16);             // Wk block
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // vectors and halo rows are in the LDS
  FragH2<8> fx;
  fx.set(xp);
  LcpeHalo<CF>::apply(xp, halo, lvec, tile * 32 + i, N, lane);
  float res[16], bo[16];                           // x' and the output bias, feature block `wave`: the residual of to_out
  load_vec_block(bo, lvec + 6 * C, wave, h);
#pragma unroll
  for (int r = 0; r < 16; ++r) res[r] = xp[16 * wave + r];
  // ---- to_q (waves 0, 1) ----
  if (wave < 2) {
    FragH2<8> nx;
    {
      float xn[CF];
      layernorm_frag<CF>(xn, xp, lvec + 4 * C, lvec + 5 * C, h);
      nx.set(xn);
    }
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(acc, wb[s], wb[8 + s], nx.h[s], nx.l[s]);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
    f16x8* xh = reinterpret_cast<f16x8*>(xq);
    f16x8 hi, lo;
    split8h(&t[0], hi, lo); xh[(0 * 4 + 2 * wave) * 64 + lane] = hi; xh[(1 * 4 + 2 * wave) * 64 + lane] = lo;
    split8h(&t[8], hi, lo); xh[(0 * 4 + 2 * wave + 1) * 64 + lane] = hi; xh[(1 * 4 + 2 * wave + 1) * 64 + lane] = lo;
    load_frags(wb, front_wst + (size_t)(8 + wave) * kStageFloats, k16);               // Wk block
  }
  // ---- Q' ----
  {
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(acc, wa[s], wa[8 + s], fx.h[s], fx.l[s]);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bq[r]);
    store_block_h2(q_out + toff, wave, t, lane);
    load_frags(wa, front_wst + (size_t)(12 + wave) * kStageFloats, k16);              // Wv block
  }
  // ---- K ----
  {
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(acc, wb[s], wb[8 + s], fx.h[s], fx.l[s]);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bk[r]);
    store_block_h2(k_out + toff, wave, t, lane);
    if (wave < ttiles) load_frags(wb, ctx_pair + (size_t)wave * kStageFloats, k16);    // this wave's first context tile (Kc | Vc)
  }
  // ---- V (feature on lane) ----
  {
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(acc, fx.h[s], fx.l[s], wa[s], wa[8 + s]);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, bv);
    store_block_h2(v_out + toff, wave, t, lane);
    if (wave + 4 < ttiles) load_frags(wa, ctx_pair + (size_t)(wave + 4) * kStageFloats, k16);
  }
  __syncthreads();                                 // q is in the LDS
  FragH2<4> qx;
  {
    const f16x8* xh = reinterpret_cast<const f16x8*>(xq);
#pragma unroll
    for (int s = 0; s < 4; ++s) { qx.h[s] = xh[(0 * 4 + s) * 64 + lane]; qx.l[s] = xh[(1 * 4 + s) * 64 + lane]; }
  }
  // ---- context tiles wave, wave + 4, ... ([0..7] Kc planes x steps, [8..15] Vc planes x slots) ----
  f32x16 oacc[2];
  oacc[0] = zero16(); oacc[1] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  auto ctx_tile = [&](const int t, const f16x8 (&w)[16]) {
    f32x16 sc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) mma3(sc, w[s], w[4 + s], qx.h[s], qx.l[s]);
    float x[16];
    float mx = -INFINITY;
    const int jbase = t * 32 + 4 * h;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int jl = 8 * (rr >> 2) + (rr & 3);
      x[rr] = (jbase + jl < T) ? sc[rr] : -INFINITY;
      mx = fmaxf(mx, x[rr]);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const float m_off = m_new - 10.0f;
    float ls = 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) { x[rr] = __builtin_amdgcn_exp2f(x[rr] - m_off); ls += x[rr]; }
    l_half = fmaf(l_half, alpha, ls);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) oacc[db][rr] *= alpha;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int sl = 2 * db + s2;
        mma3(oacc[db], w[8 + sl], w[12 + sl], ph, pl);
      }
    }
  };
  if (wave < ttiles) ctx_tile(wave, wb);
  // Wo block `wave` (32 x 64: 8 fragments) takes the first buffer's place
  load_frags(wb, attn_wst + (size_t)(2 + (wave >> 1)) * kStageFloats + (wave & 1) * (2 * 4 * 64 * 4), k8);
  if (wave + 4 < ttiles) ctx_tile(wave + 4, wa);
  for (int t = wave + 8; t < ttiles; t += 4) {     // (more than 256 context tokens: one more tile per wave and pass)
    load_frags(wa, ctx_pair + (size_t)t * kStageFloats, k16);
    ctx_tile(t, wa);
  }
  // ---- merge the four waves' partial softmaxes (like key-split partials, in wave order), to_out ----
  {
    float* mine = px + wave * (34 * 64) + lane;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) mine[(16 * db + rr) * 64] = oacc[db][rr];
    mine[32 * 64] = m_run;
    mine[33 * 64] = l_half;
    __syncthreads();
    float mw[4], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = px[w * (34 * 64) + 32 * 64 + lane]; M = fmaxf(M, mw[w]); }
    float o[32], l = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) o[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a = __builtin_amdgcn_exp2f(mw[w] - M);          // (a wave without a tile: 2^-inf = 0)
      l = fmaf(px[w * (34 * 64) + 33 * 64 + lane], a, l);
#pragma unroll
      for (int e = 0; e < 32; ++e) o[e] = fmaf(px[w * (34 * 64) + e * 64 + lane], a, o[e]);
    }
    const float inv = 1.0f / xhalf_sum(l);
    FragH2<4> ox;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float t[16];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) t[rr] = o[16 * db + rr] * inv;
      ox.set_block(db, t);
    }
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) mma3(acc, wb[s], wb[4 + s], ox.h[s], ox.l[s]);
    float t[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) t[rr] = fmaf(acc[rr], kH2Inv, bo[rr]) + res[rr];
    store_block_p32(x1_out + toff, wave, t, lane);
  }
  {                                                // (never true: keeps the touches alive without a wait before this point)
    float touched = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) touched += tv[q];
    if (touched == 1.2345e38f) x1_out[toff] = touched;
  }
}

