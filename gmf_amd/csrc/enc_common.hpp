// Constants and small device helpers shared by the encoder kernel files.
#pragma once
#include "mfma_core.hpp"
#include "launchers.hpp"

namespace gmf {

constexpr int C = 128;        // correspondence feature width
constexpr int CF = C / 2;     // fragment length of a C-wide row
constexpr int DH = 64;        // cross-attention head width (PointDSC: C/2)
constexpr int DHF = DH / 2;
constexpr int FFH = 512;      // GEGLU hidden width (4*C)
constexpr int kWavesPerWG = 4;

// ---- small helpers ---------------------------------------------------------------------
// bias (or any per-feature vector) for out-block mb in fragment order
GMF_DEVINL void load_vec_block(float (&v)[16], const float* __restrict__ vec, int mb, int h) {
  const float4* p = reinterpret_cast<const float4*>(vec + 32 * mb) + h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = p[2 * q];
    v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}

// store a Y^T block (16 regs, rows on lanes) as the P32 columns [32mb, 32mb+32) of a K-wide tile
GMF_DEVINL void store_block_p32(float* __restrict__ tile_base, int mb, const float (&t)[16], int lane) {
  float4* p = reinterpret_cast<float4*>(tile_base) + (4 * mb) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q * 64] = make_float4(t[4 * q + 0], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}

GMF_DEVINL void load_block_p32(float (&t)[16], const float* __restrict__ tile_base, int mb, int lane) {
  const float4* p = reinterpret_cast<const float4*>(tile_base) + (4 * mb) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = p[q * 64];
    t[4 * q + 0] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
  }
}

// store a Y block (feature on lane) as d-block db of a T image tile
GMF_DEVINL void store_block_timg(float* __restrict__ tile_base, int db, const f32x16& a, float bias, int lane) {
  float4* p = reinterpret_cast<float4*>(tile_base) + (4 * db) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    p[q * 64] = make_float4(a[4 * q + 0] + bias, a[4 * q + 1] + bias, a[4 * q + 2] + bias, a[4 * q + 3] + bias);
}

// fp16x2 image of a 32 x 128 tile: 16-byte unit ((plane*8 + slot)*64 + lane), planes hi | lo.
GMF_DEVINL void store_block_h2(float* __restrict__ tile_base, int blk, const float (&t)[16], int lane) {
  f16x8* base = reinterpret_cast<f16x8*>(tile_base);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f16x8 hi, lo;
    split8h(&t[8 * half], hi, lo);
    const int slot = 2 * blk + half;
    base[(0 * 8 + slot) * 64 + lane] = hi;
    base[(1 * 8 + slot) * 64 + lane] = lo;
  }
}

// [r5] The same routine writes the K image of that form (t: the lane's KEY, 16 channels of block db; one scale per (key, 32-channel
// block)): the cross products of S = K Q'^T run on the fp8 pipe too (q_planes8 below is the B operand's side).
// The V image of the parity attention kernel's "pv_fp8" form (k_scattn_h2p<3, CFMT, true>): the high fp16 plane as above; in
// place of the low fp16 plane, per feature block db, the two e4m3 operands of the CROSS products of O += P V -
//   16-byte unit (1*8 + 2 db)*64 + lane :  e4m3((v - hi) / (s / 2^11))   of the lane's 16 keys, register order
//   16-byte unit (1*8 + 2 db + 1)*64 + lane :  e4m3(v / s)
// with ONE power-of-two scale s per (feature, 32-key tile): the tile's largest |v| of that feature lands in [128, 256).  The
// E8M0 bytes of the two blocks go to `scale_word` (byte db: lanes 0..31 carry the low block's, lanes 32..63 the high block's -
// where v_mfma_scale_f32_32x32x64_f8f6f4 reads them), one dword per lane and tile.  t: the lane's feature, 16 keys (V^T block).
GMF_DEVINL void store_block_v8(float* __restrict__ tile_base, int db, const float (&t)[16], int lane, unsigned& scale_word) {
  f16x8* base = reinterpret_cast<f16x8*>(tile_base);
  float mx = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, __builtin_fabsf(t[r]));
  mx = xhalf_max_swap(mx);
  // biased exponent e of the maximum: max in [2^(e-127), 2^(e-126)); s = 2^(e-134).  The scale has a FLOOR of 2^-22
  // (sb >= 105): below |v| = 2^-14 the high plane fp16(v) is subnormal, so the residual v - hi is no longer 2^-12 |v| but
  // absolute, up to 2^-25 - with the scale following a tile maximum below 2^-15 that residual / (s / 2^11) left e4m3's
  // range (448) and the conversion returned NaN bytes, which poisoned the feature of O for every query of the pair
  // (ADVICE r3).  With the floor the scaled residual is at most 2^-25 / 2^-33 = 256; a tile of values that small keeps
  // absolute errors of 2^-42 in its cross terms - far below anything else in the product.
  const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  const int sb = max(e - 7, 105);
  const float s_hi = __uint_as_float((unsigned)sb << 23), s_lo = __uint_as_float((unsigned)(sb - 11) << 23);
  scale_word |= (unsigned)((lane & 32) ? sb : sb - 11) << (8 * db);
  i32x4 v8, l8;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f16x8 hi;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const float x0 = t[8 * half + j], x1 = t[8 * half + j + 1];
      const f32x2 x = {x0, x1};
      const f16x2 hh = __builtin_convertvector(x, f16x2);
      const f32x2 rl = resid2h(hh, x0, x1);
      hi[j] = hh[0]; hi[j + 1] = hh[1];
      const int w = 2 * half + (j >> 2);
      if (j & 2) { v8[w] = cvt2_fp8_f32<true>(v8[w], x0, x1, s_hi); l8[w] = cvt2_fp8_f32<true>(l8[w], rl[0], rl[1], s_lo); }
      else { v8[w] = cvt2_fp8_f32<false>(0, x0, x1, s_hi); l8[w] = cvt2_fp8_f32<false>(0, rl[0], rl[1], s_lo); }
    }
    base[(0 * 8 + 2 * db + half) * 64 + lane] = hi;
  }
  reinterpret_cast<i32x4*>(tile_base)[(1 * 8 + 2 * db) * 64 + lane] = l8;
  reinterpret_cast<i32x4*>(tile_base)[(1 * 8 + 2 * db + 1) * 64 + lane] = v8;
}

// [r5] The same planes for the B operand of S = K Q'^T (scattn_h2p_body, PVF8): block cb of a query row's Q' (its 16 values on this
// lane, given as the split-fp16 planes the kernel already holds: x = hi + lo) as the 32 operand bytes
//     bytes 0..15  e4m3((hi + lo) / s)          - meets the K image's e4m3(k - k_hi) bytes (store_block_v8 on a K block)
//     bytes 16..31 e4m3(lo / (s / 2^11))        - meets its e4m3(k) bytes
// with ONE power-of-two scale per (query, 32-channel block): the block's largest |q| in [128, 256) s, floor 2^-22 as for V.
// Byte cb of `scale_word`: lanes 0..31 carry block 0's E8M0 (s), lanes 32..63 block 1's (s / 2^11).  Built from (hi, lo) - not
// from the fp32 value - so that a Q' read back from an image gives the same bytes as one projected in the prologue.
GMF_DEVINL void q_planes8(const f16x8& h0, const f16x8& l0, const f16x8& h1, const f16x8& l1, int cb, int lane, i32x8& out,
                          unsigned& scale_word) {
  float q[16], ql[16];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ql[e] = (float)l0[e]; q[e] = (float)h0[e] + ql[e];
    ql[8 + e] = (float)l1[e]; q[8 + e] = (float)h1[e] + ql[8 + e];
  }
  float mx = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, __builtin_fabsf(q[r]));
  mx = xhalf_max_swap(mx);
  const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  const int sb = max(e - 7, 105);
  const float s_hi = __uint_as_float((unsigned)sb << 23), s_lo = __uint_as_float((unsigned)(sb - 11) << 23);
  scale_word |= (unsigned)((lane & 32) ? sb - 11 : sb) << (8 * cb);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int a = cvt2_fp8_f32<false>(0, q[4 * w], q[4 * w + 1], s_hi);
    out[w] = cvt2_fp8_f32<true>(a, q[4 * w + 2], q[4 * w + 3], s_hi);
    int b = cvt2_fp8_f32<false>(0, ql[4 * w], ql[4 * w + 1], s_lo);
    out[4 + w] = cvt2_fp8_f32<true>(b, ql[4 * w + 2], ql[4 * w + 3], s_lo);
  }
}

// ---- the "pv_fp8" guard (PvGuard, launchers.hpp) ---------------------------------------------------------------
// Does this pair's V image of this layer carry e4m3 cross planes (true) or its low fp16 plane (false)?  Uniform per workgroup, two
// scalar loads; the kernels that WRITE the image and the ones that multiply it evaluate the same words.  (A NaN statistic -
// non-finite features - compares false: three products; the status word of k_head reports the NaN itself.)
GMF_DEVINL bool pv_planes_on(const unsigned* __restrict__ v_scale, const PvGuard& g, int pair) {
  if (!v_scale) return false;
  if (!g.stat) return true;
  return __uint_as_float(g.stat[(size_t)pair * kPvStatStride]) <= *g.thr2;
}
// Raises stat[pair] to the largest |f|^2 of this wave's 32 rows (rows on lanes; ssq_half = this lane's 64 features of its row, the
// other 64 live in lane ^ 32).  Sums of squares are >= 0, so the float bits order like unsigned integers.
GMF_DEVINL void pv_stat_raise(unsigned* __restrict__ stat, int pair, float ssq_half, bool row_valid, int lane) {
  float s = xhalf_sum(ssq_half);
  s = row_valid ? s : 0.f;
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s = fmaxf(s, __shfl_xor(s, m, 64));
  // A wave looks first - a relaxed device-scope load - and only sends the atomic when its value would raise the word (a stale read
  // costs a redundant atomic, never a wrong maximum; ~ln(waves) atomics per pair get through); every pair has a cache line of its own
  // (kPvStatStride).
  if (lane == 0) {
    unsigned* const w = stat + (size_t)pair * kPvStatStride;
    const unsigned v = __float_as_uint(s);
    if (v > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, v);
  }
}

// LCPE (fusion_layer.py:118-128): y[row] = x[row] + b + w0*x[row-1] + w1*x[row] + w2*x[row+1],
// zero padding outside [0, n_rows).  taps = w0[C] | w1[C] | w2[C] | b[C].
GMF_DEVINL void lcpe_frag(float (&y)[CF], const float* __restrict__ pair_base, int row, int n_rows,
                          const float* __restrict__ taps, int h) {
  const float4* base = reinterpret_cast<const float4*>(pair_base);
  const bool has_m = row >= 1, has_p = row + 1 < n_rows;
  const int rm = has_m ? row - 1 : row, rp = has_p ? row + 1 : row;
  const float4* pc = base + (size_t)(row >> 5) * (CF / 4) * 64 + h * 32 + (row & 31);
  const float4* pm = base + (size_t)(rm >> 5) * (CF / 4) * 64 + h * 32 + (rm & 31);
  const float4* pp = base + (size_t)(rp >> 5) * (CF / 4) * 64 + h * 32 + (rp & 31);
  const float4* t0 = reinterpret_cast<const float4*>(taps) + h;
  const float fm = has_m ? 1.f : 0.f, fp = has_p ? 1.f : 0.f;
#pragma unroll
  for (int g = 0; g < CF / 4; ++g) {
    const float4 xc = pc[g * 64], xm = pm[g * 64], xp = pp[g * 64];
    const float4 w0 = t0[2 * g], w1 = t0[2 * g + C / 4], w2 = t0[2 * g + 2 * (C / 4)], b = t0[2 * g + 3 * (C / 4)];
    y[4 * g + 0] = xc.x + b.x + w0.x * (fm * xm.x) + w1.x * xc.x + w2.x * (fp * xp.x);
    y[4 * g + 1] = xc.y + b.y + w0.y * (fm * xm.y) + w1.y * xc.y + w2.y * (fp * xp.y);
    y[4 * g + 2] = xc.z + b.z + w0.z * (fm * xm.z) + w1.z * xc.z + w2.z * (fp * xp.z);
    y[4 * g + 3] = xc.w + b.w + w0.w * (fm * xm.w) + w1.w * xc.w + w2.w * (fp * xp.w);
  }
}

// ---- LCPE with the neighbour rows taken from the neighbour LANES -------------------------------------------------------
// lcpe_frag above loads the rows row - 1, row, row + 1 and the four tap vectors from global memory: 7 float4 loads per
// 4 features, which the compiler - short of registers around them - issues as one dependent round trip after the other
// (k_fusion_attn_w_h2: 40 x `s_waitcnt vmcnt(0)`, 25 of the kernel's 48 us at one wave per SIMD).  Here a lane's rows
// row +- 1 are its neighbour lanes' registers (v_mov_b32_dpp wave_shr / wave_shl); only the row above the tile's first row and
// the row below its last one come from memory - one LDS-DMA gather per K = 256 floats into a per-wave halo area - and the taps
// are read from a copy of the kernel's per-feature vectors in the LDS.
//   halo (per wave, 2 * K floats): float4 unit r * (K/4) + 2 g + h = features 8 g + 4 h .. + 3 of the row above (r = 0) / below (r = 1)
//   taps_lds: w0[K] | w1[K] | w2[K] | b[K]
template <int KF>
struct LcpeHalo {
  static constexpr int K = 2 * KF, U = K / 4;
  GMF_DEVINL static void issue(const float* __restrict__ pair_base, int tile, int tiles, float* halo, int lane) {
#pragma unroll
    for (int q = 0; q < (2 * U) / 64; ++q) {
      const int w = q * 64 + lane, r = w / U, u = w % U;
      const bool other = (r == 0) ? (tile > 0) : (tile + 1 < tiles);        // else: any finite value of the own tile (it is masked)
      const int st = other ? (r == 0 ? tile - 1 : tile + 1) : tile;
      const int si = (r == 0) == other ? 31 : 0;
      const float* src = pair_base + (size_t)st * (32 * K) + (size_t)(((u >> 1) * 64 + (u & 1) * 32 + si) * 4);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(halo + q * 256), 16, 0, 0);
    }
  }
  // the same two rows from a ROW-MAJOR [n_rows, K] tensor (the kernels that read their input without a packing pass)
  GMF_DEVINL static void issue_rowmajor(const float* __restrict__ rows_base, int tile, int n_rows, float* halo, int lane) {
#pragma unroll
    for (int q = 0; q < (2 * U) / 64; ++q) {
      const int w = q * 64 + lane, r = w / U, u = w % U, g = u >> 1, hh = u & 1;
      int row = (r == 0) ? tile * 32 - 1 : tile * 32 + 32;
      row = min(max(row, 0), n_rows - 1);                                    // (outside: any finite row, it is masked)
      const float* src = rows_base + (size_t)row * K + (32 * (g >> 2) + 8 * (g & 3) + 4 * hh);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(halo + q * 256), 16, 0, 0);
    }
  }
  GMF_DEVINL static float from_lane_below(float v) {   // lane L gets lane L - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
  }
  GMF_DEVINL static float from_lane_above(float v) {   // lane L gets lane L + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
  }
  // x (the tile's own rows, one per lane) -> x + b + w0 * x[row - 1] + w1 * x + w2 * x[row + 1], zero padding outside [0, n_rows)
  GMF_DEVINL static void apply(float (&x)[KF], const float* halo, const float* taps_lds, int row, int n_rows, int lane) {
    const int h = lane >> 5, i = lane & 31;
    const bool has_m = row >= 1, has_p = row + 1 < n_rows, first = i == 0, last = i == 31;
    const float4* hl = reinterpret_cast<const float4*>(halo) + h;
    const float4* t0 = reinterpret_cast<const float4*>(taps_lds) + h;
#pragma unroll
    for (int g = 0; g < KF / 4; ++g) {
      const float4 up = hl[2 * g], dn = hl[U + 2 * g];
      const float4 w0 = t0[2 * g], w1 = t0[2 * g + K / 4], w2 = t0[2 * g + 2 * (K / 4)], b = t0[2 * g + 3 * (K / 4)];
      const float upv[4] = {up.x, up.y, up.z, up.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w};
      const float w0v[4] = {w0.x, w0.y, w0.z, w0.w}, w1v[4] = {w1.x, w1.y, w1.z, w1.w}, w2v[4] = {w2.x, w2.y, w2.z, w2.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xc = x[4 * g + e];
        float xm = from_lane_below(xc), xq = from_lane_above(xc);
        xm = first ? upv[e] : xm;
        xq = last ? dnv[e] : xq;
        xm = has_m ? xm : 0.f;
        xq = has_p ? xq : 0.f;
        x[4 * g + e] = xc + bv[e] + w0v[e] * xm + w1v[e] * xc + w2v[e] * xq;
      }
    }
  }
};

// [r4] Priority 1 for the LAST 256 workgroups of a grid (256 = the CUs of the part; the last, possibly partial, 256-block of the
// dispatch order).  Two co-resident workgroups of equal priority share a SIMD's issue slots by AGE: the older one runs as if nearly alone,
// the younger fills what is left (profiles/r04_linear_timeline.txt: 146 k against 170-185 k cycles).  While every slot stays busy that
// is as good as any order - but the workgroups dispatched last are the younger one on whatever CU they land on, run slowest, and are what
// the launch waits for at its end.  With priority over their (older) neighbours they finish beside them instead of behind them:
// -0.5 .. -1.8 % per step on EVERY grid from 560 to 2560 workgroups (32 x 5000: 16.96 -> 16.77 ms in one job).  Rules that touch earlier
// workgroups lose on some grids: every other 256-block -1.5 .. +5.7 %, the whole last round of 512 -1.9 .. +1.8 % (profiles/
// r04_compat_stream_probes.txt has the sweeps).  One scalar instruction per workgroup; results do not depend on it.
GMF_DEVINL void tail_priority(const unsigned linear_wg, const unsigned n_wgs) {
  if (linear_wg >= ((n_wgs - 1u) & ~255u)) __builtin_amdgcn_s_setprio(1);
}

// Split-fp16 weight images are stored as 256 W (packing.p32_h2s / p16_h2s): lo = fp16(256 w - hi) then stays a normal
// fp16 number for |w| >= 2^-11 instead of falling into the subnormals (spacing 2^-24: a weight of 0.006 - the folded
// softmax scale makes Wq that small - kept only 16 significant bits).  Consumers fold 2^-8 into the bias add
// (fmaf(acc, kH2Inv, b): same instruction count).  The GEGLU W1 matrices stay unscaled (their accumulators start from the
// bias and feed the GELU directly).
constexpr float kH2Inv = 1.0f / 256.0f;

// a row fragment as two fp16 planes: NS = K/16 k-steps
template <int NS>
struct FragH2 {
  f16x8 h[NS], l[NS];
  GMF_DEVINL void set(const float* x) {
#pragma unroll
    for (int s = 0; s < NS; ++s) split8h(x + 8 * s, h[s], l[s]);
  }
  // fill k-steps [2*blk, 2*blk+2) from one 32-wide block of accumulators / fragment values
  GMF_DEVINL void set_block(int blk, const float (&t)[16]) {
    split8h(&t[0], h[2 * blk], l[2 * blk]);
    split8h(&t[8], h[2 * blk + 1], l[2 * blk + 1]);
  }
};

// acc += W(32 x 16*NS) * X^T   (rows on lanes); weight block image: 16-byte unit ((plane*NS + s)*64 + lane)
template <int NS>
GMF_DEVINL void mma_wx_h2(f32x16& acc, const f16x8* lw, const FragH2<NS>& x) {
#pragma unroll
  for (int s = 0; s < NS; ++s) mma3(acc, lw[(0 * NS + s) * 64], lw[(1 * NS + s) * 64], x.h[s], x.l[s]);
}

// acc += X * W^T   (feature on lane)
template <int NS>
GMF_DEVINL void mma_xw_h2(f32x16& acc, const f16x8* lw, const FragH2<NS>& x) {
#pragma unroll
  for (int s = 0; s < NS; ++s) mma3(acc, x.h[s], x.l[s], lw[(0 * NS + s) * 64], lw[(1 * NS + s) * 64]);
}

// The same products with NP = 3 (parity: split-fp16, three partial products) or NP = 1 (the throughput numerics mode: only the
// high planes are multiplied - the low planes are neither read from the LDS nor formed)
template <int NP>
GMF_DEVINL void mma_np(f32x16& acc, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
  if (NP == 3) mma3(acc, ah, al, bh, bl);
  else acc = mfma_h16(ah, bh, acc);
}
template <int NS, int NP>
GMF_DEVINL void mma_wx_h2n(f32x16& acc, const f16x8* lw, const FragH2<NS>& x) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (NP == 3) mma3(acc, lw[(0 * NS + s) * 64], lw[(1 * NS + s) * 64], x.h[s], x.l[s]);
    else acc = mfma_h16(lw[(0 * NS + s) * 64], x.h[s], acc);
  }
}
template <int NS, int NP>
GMF_DEVINL void mma_xw_h2n(f32x16& acc, const f16x8* lw, const FragH2<NS>& x) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (NP == 3) mma3(acc, x.h[s], x.l[s], lw[(0 * NS + s) * 64], lw[(1 * NS + s) * 64]);
    else acc = mfma_h16(x.h[s], lw[(0 * NS + s) * 64], acc);
  }
}

GMF_DEVINL const f16x8* as_h2(const float4* p) { return reinterpret_cast<const f16x8*>(p); }

GMF_DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Branch-free erf for the GELU of the feed-forward kernels.  ocml's erff compiles to a per-element exec-mask branch
// (two ranges, ~42 instructions plus the mask bookkeeping) that also pins the surrounding MFMAs in place; here both
// ranges are evaluated (13 FMAs, one v_exp_f32) and selected.  Polynomials: N. Juffa's erff (max error < 1 ulp in
// each range with a correctly rounded exp; with v_exp_f32 on r * log2(e) the large-|a| range stays below 2 ulp,
// because exp(r) <= 0.43 there enters 1 - exp(r) with a weight below one half).
GMF_DEVINL float erf_bf(float a) {
  const float t = fabsf(a), s = a * a;
  // |a| > 0.927734375:  erf = 1 - exp(r(t))
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = 1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f);
  // |a| <= 0.927734375:  erf = a + a * p(a^2)
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, t, t);
  return copysignf(t > 0.927734375f ? big : small, a);
}
GMF_DEVINL float gelu_erf_bf(float x) { return 0.5f * x * (1.0f + erf_bf(x * 0.70710678118654752440f)); }

// GELU(x) = x * Phi(x) needs Phi to ~6e-8 ABSOLUTE, not erf to a relative ulp, so one formula covers every x:
//   Phi(x) = 1 - erfc(t)/2 (x >= 0),  erfc(t)/2 (x < 0),  t = |x|/sqrt(2)   =>   GELU(x) = max(x, 0) - 0.5 |x| erfc(t)
//   erfc(t) = exp2(t * p(t)) with p a degree-8 polynomial fitted (weighted minimax, tools/fit_gelu.py) to log2(erfc(t))/t
//   on [0, 4.3]: |erfc error| < 2.3e-9; beyond 4.3 erfc < 1.2e-9 and t is clamped.
// 16 vector instructions with one v_exp_f32 and no select (the two-range erf above: 26).  Max |error| against the exact
// GELU over |x| <= 12 evaluated in fp32: 2.4e-7 - the reference's own fp32 formula 0.5 x (1 + erf(x/sqrt 2)): 4.5e-7.
GMF_DEVINL float gelu_erf_1r(float x) {
  const float ax = fabsf(x);
  const float t = fminf(ax * 0.70710678118654752440f, 4.3f);
  float p = 1.160468673e-05f;
  p = fmaf(p, t, -1.529631263e-04f);
  p = fmaf(p, t, 8.482293342e-04f);
  p = fmaf(p, t, -2.274774713e-03f);
  p = fmaf(p, t, 8.479235839e-05f);
  p = fmaf(p, t, 2.772448584e-02f);
  p = fmaf(p, t, -1.483079195e-01f);
  p = fmaf(p, t, -9.184429049e-01f);
  p = fmaf(p, t, -1.627907276e+00f);
  const float e = __builtin_amdgcn_exp2f(p * t);
  return fmaf(-0.5f * ax, e, fmaxf(x, 0.f));
}


// ---- the compat matrix and the key-point image: bodies shared by their own kernels (encoder_kernels.hip) and the small-grid
// prologue's role kernels (encoder_h2.hip) ----
// d = ||a|| - ||b|| from the squared lengths in the reference's form, sqrt(a2) - sqrt(b2) with correctly rounded square
// roots (sqrt_cr), i.e. the same roundings as torch.norm on the CPU (PointDSC.py:217-219).
GMF_DEVINL float len_diff(float a2, float b2) { return sqrt_cr(a2) - sqrt_cr(b2); }

// compat * score for one element; lp points at this lane-half's first key of the tile (pts8 rows)
GMF_DEVINL float compat_times(const float4* lp, int jl, const float (&si)[3], const float (&ti)[3], float inv_sig2, float sc) {
  const float4 a = lp[2 * jl], b = lp[2 * jl + 1];
  const float ax = si[0] - a.x, ay = si[1] - a.y, az = si[2] - a.z;
  const float bx = ti[0] - b.x, by = ti[1] - b.y, bz = ti[2] - b.z;
  const float d = len_diff(fmaf(az, az, fmaf(ay, ay, ax * ax)), fmaf(bz, bz, fmaf(by, by, bx * bx)));
  return fmaxf(1.0f - d * d * inv_sig2, 0.f) * sc;
}

constexpr int kJPerWave = 8;

// c is exactly symmetric (the squared differences do not see the sign of s_i - s_j), so only tiles J >= I are
// evaluated; a tile with J > I is also written as tile (J, I) after a 32 x 32 transpose through a per-wave LDS buffer
// (16 ds_write_b32 + 16 ds_read_b32 instead of 16 x ~26 vector instructions with two correctly rounded square roots).
// FMT (CompatCache::fmt): 0 = fp32, 4 KiB per tile.  The 16-bit formats are 2 KiB per tile ([q2][lane][8 x 16 bit], registers
// r = 8 q2 .. 8 q2 + 7) - half the stream of the 12 attention launches:
//   1 = c as fp16 (the throughput numerics mode, gmf_set_tuning "precision" = 1);
//   2 = c as 16-bit fixed point, u = rint(65535 c) (uniform absolute error <= 7.6e-6; gmf_set_tuning "compat_format" = 2, opt-in).
// (Measured and dropped, round 3: fp16 of 1 - c - exact where c = 1, coarse below c = 0.5 - and 16-bit fixed point of sqrt(1 - c),
// the error structure of the reference's own fp32 rounding; profiles/r03_compat_formats.txt.)
// (I, by, pair): the workgroup's query tile, its block of 4 * kJPerWave key tiles, its pair; `tr`: kCompatLdsFloats floats.  No
// workgroup barrier inside (the transposes are per wave), so a role kernel may call it from any subset of its workgroups.
constexpr int kCompatLdsFloats = 4 * 32 * 33;
template <int FMT>
GMF_DEVINL void compat_build_body(float* tr, const int I, const int by, const int pair, const float* __restrict__ pts8,
                                  float* __restrict__ c_dense, int N, int tiles, float inv_sig2, const PairTab* __restrict__ ptab) {
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_p = (pair_rows(ptab, pair, N) + 31) >> 5;      // ragged batch: this pair's own tiles (the slot keeps `tiles`)
  if (I >= tiles_p) return;
  const size_t pbase = (size_t)pair * tiles;
  float si[3], ti[3];
  {
    const float4* pp = reinterpret_cast<const float4*>(pts8 + (pbase * 32 + (size_t)I * 32 + i) * 8);
    const float4 a = pp[0], b = pp[1];
    si[0] = a.x; si[1] = a.y; si[2] = a.z; ti[0] = b.x; ti[1] = b.y; ti[2] = b.z;
  }
  constexpr int kTile16 = FMT ? 128 : 256;          // 16-byte pieces per tile
  float4* const cbase = reinterpret_cast<float4*>(c_dense) + pbase * (size_t)tiles * kTile16 + lane;
  float4* const crow = cbase + (size_t)I * tiles * kTile16;
  auto store_tile = [&](float4* ct, const float (&c)[16]) {
    if (FMT == 1) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        f16x8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)c[8 * q2 + e];
        ct[q2 * 64] = __builtin_bit_cast(float4, hv);
      }
    } else if (FMT == 2) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        unsigned wv[4];
        auto enc = [](float cv) { return (unsigned)__builtin_rintf(cv * 65535.0f); };
#pragma unroll
        for (int e = 0; e < 8; e += 2) wv[e >> 1] = enc(c[8 * q2 + e]) | (enc(c[8 * q2 + e + 1]) << 16);
        ct[q2 * 64] = make_float4(__uint_as_float(wv[0]), __uint_as_float(wv[1]), __uint_as_float(wv[2]), __uint_as_float(wv[3]));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) ct[q * 64] = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    }
  };
  float* const mt = tr + wave * 32 * 33;
  const int j0 = (by * 4 + wave) * kJPerWave;
  for (int J = max(j0, I); J < min(tiles_p, j0 + kJPerWave); ++J) {
    const float4* lp = reinterpret_cast<const float4*>(pts8 + (pbase * 32 + (size_t)J * 32) * 8) + 8 * h;
    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = compat_times(lp, 8 * (r >> 2) + (r & 3), si, ti, inv_sig2, 1.0f);
    store_tile(crow + (size_t)J * kTile16, c);
    if (J > I) {
      // tile (J, I): lane (h, i), register r = element (row i of J, column jl of I) = c(I: jl, J: i) = M[jl][i]
#pragma unroll
      for (int r = 0; r < 16; ++r) mt[i * 33 + 8 * (r >> 2) + 4 * h + (r & 3)] = c[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float d[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = mt[(8 * (r >> 2) + 4 * h + (r & 3)) * 33 + i];
      store_tile(cbase + ((size_t)J * tiles + I) * kTile16, d);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// src,tgt [B,N,3] -> pts8 [B, Npad, 8] (zero padded rows)
GMF_DEVINL void pack_pts8_body(const long idx, const float* __restrict__ src, const float* __restrict__ tgt, float* __restrict__ dst,
                               int N, int Npad, long total, const PairTab* __restrict__ ptab, unsigned* __restrict__ zero_words,
                               int n_zero) {
  // [r5] the "pv_fp8" guard's statistics start every forward at zero: cleared here, by the first kernel of the stream that runs
  // before anything raises them (a hipMemsetAsync of these few words costs a 9 us fill kernel of its own)
  if (zero_words && idx < n_zero) zero_words[idx] = 0u;
  if (idx >= total) return;
  const int row = idx % Npad;
  const long b = idx / Npad;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
  if (row < pair_rows(ptab, (int)b, N)) {
    const size_t r0 = pair_row0(ptab, (int)b, N) + row;
    const float* ps = src + r0 * 3;
    const float* pt = tgt + r0 * 3;
    a = make_float4(ps[0], ps[1], ps[2], 0.f);
    c = make_float4(pt[0], pt[1], pt[2], 0.f);
  }
  reinterpret_cast<float4*>(dst)[2 * idx] = a;
  reinterpret_cast<float4*>(dst)[2 * idx + 1] = c;
}


}  // namespace gmf
