// Device-side building blocks shared by every encoder kernel (gfx950 / CDNA4 only).
//
// Execution model used throughout ("rows on lanes"):
//   * a wavefront (64 lanes) owns one TILE of 32 rows (correspondences or image tokens);
//     lane l = (h, i) with i = l & 31 the row inside the tile and h = l >> 5 the K-half;
//   * a row vector of width K lives in registers as a FRAGMENT  float x[K/2]  with
//         x[4*g + e] = X[row i][8*g + 4*h + e]          g in [0,K/8), e in [0,4)
//     which is at the same time the A- and the B-operand layout of
//     v_mfma_f32_32x32x2_f32 (one f32 per lane, lane l holds element (l&31, k = l>>5)),
//     with the contraction index visited in the order k(step s = 4g+e, half h) = 8g+4h+e;
//   * D = mfma(A = weight image, B = fragment) gives  Y^T[m][i]  for 32 output features:
//     lane (h,i), register r holds  m = 8*(r>>2) + 4*h + (r&3)  - exactly fragment
//     element r of the 32-wide block, so the result chains into the next layer with no
//     data movement (no LDS round trip, no shuffles);
//   * D = mfma(A = fragment, B = weight image) gives  Y[i'][m]  with the FEATURE on the
//     lane (m = l & 31) and rows i' = 8*(r>>2)+4*h+(r&3) in the registers: this is the
//     layout the attention kernels want for V (A-operand of O^T += V^T P^T).
//
// Memory images (all fp32, float4 granularity, 1 KiB per (tile, g) = one coalesced
// wave-wide access, and one LDS-DMA instruction):
//   P32 image of X[rows, K]   : float4 index ((tile*(K/8) + g)*64 + lane)
//                               = X[32*tile + i][8g+4h .. 8g+4h+3]
//   T   image of V[rows, D]   : float4 index (((tile*(D/32) + db)*4 + q)*64 + lane)
//                               = V[32*tile + 8q+4h+e][32*db + i], e = 0..3
//   A weight matrix W[M, K] (PyTorch [out, in]) is stored as its own P32 image, so
//   out-block mb (32 output features) is the contiguous 32*K floats at mb*32*K.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launchers.hpp"

#define GMF_DEVINL __device__ __forceinline__

namespace gmf {

// rows of pair `pair` and where they start in the caller's row-major tensors (uniform batch: N and pair * N)
GMF_DEVINL int pair_rows(const PairTab* pt, int pair, int N) { return pt ? pt[pair].n : N; }
GMF_DEVINL size_t pair_row0(const PairTab* pt, int pair, int N) { return pt ? (size_t)pt[pair].row0 : (size_t)pair * N; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kTileRows = 32;
constexpr int kStageFloats = 4096;  // one weight / K / V stage = 16 KiB

GMF_DEVINL f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- split-fp16 ("fp16x2") arithmetic ------------------------------------------------------------
// x = xh + xl with two fp16 planes (11 + 11 significant bits, round-to-nearest-even: |x - xh - xl| <= 2^-22 |x|)
// and three partial products hh, hl, lh (the dropped l*l term is <= 2^-22 |ab|).  Per-term error ~2.4e-7 worst
// case - below the fp32 accumulation noise of a 128-term dot product - at 3/16 of the fp32-MFMA cycles, with
// operand images the size of the fp32 ones (4 B per element).  Ranges: |x| < 65504; low parts that fall into the
// fp16 subnormal range keep an absolute error <= 3e-8.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

GMF_DEVINL f32x16 mfma_h16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// x - float(hi) for a pair, one v_fma_mix_f32 each (the mix form reads the fp16 half directly: no convert-back, and the
// difference is exact either way); bit-identical to (x - (float)hi), checked by tools/ubench/fma_mix_split.hip.
GMF_DEVINL f32x2 resid2h(f16x2 hh, float x0, float x1) {
  const unsigned hu = __builtin_bit_cast(unsigned, hh);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hu), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hu), "v"(x1));
  return f32x2{r0, r1};
}

// HAZARD INVARIANT of resid2h / lo2h (hipcc places no MFMA -> VALU wait states in front of inline assembly - DESIGN.md section 3,
// "What this toolchain does behind hand-scheduled code"): x0 / x1 may be MFMA accumulators, and the asm statements below read
// them.  They are safe BY CONSTRUCTION, not by luck of scheduling: their operand `hh` is the result of the compiler-scheduled
// conversion of the SAME x0 / x1 (every caller forms it with __builtin_convertvector right before), so that conversion - which
// the hazard recogniser does guard - reads the accumulators first, and the statements cannot be issued before their operand
// exists.  Do not call these with an `hh` that was not computed from the very x0 / x1 passed along.  (The C form
// (_Float16)(x - (float)hh) compiles to cvt + cvt + sub + sub + cvt_pk: 6 instructions per pair instead of 3.)
// fp16(x - float(hi)) for a pair, written straight into the two halves of one register: v_fma_mixlo_f16 / v_fma_mixhi_f16
// round the (exact) difference to fp16 themselves, so the low plane of a pair costs 2 instructions instead of 2 + a
// v_cvt_pk (3 per pair with the hi conversion instead of 4).  Bit-identical to convert(resid2h(...)), checked over all
// magnitudes by tools/ubench/fma_mixlo_split.hip.
GMF_DEVINL f16x2 lo2h(f16x2 hh, float x0, float x1) {
  const unsigned hu = __builtin_bit_cast(unsigned, hh);
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hu), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hu), "v"(x1));
  return __builtin_bit_cast(f16x2, d);
}

GMF_DEVINL void split8h(const float* v, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 x = {v[j], v[j + 1]};
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f16x2 ll = lo2h(hh, v[j], v[j + 1]);
    hi[j] = hh[0]; hi[j + 1] = hh[1];
    lo[j] = ll[0]; lo[j + 1] = ll[1];
  }
}

// elements j, j+1 of an 8-element k-step
GMF_DEVINL void split2h(float x0, float x1, f16x8& hi, f16x8& lo, int j) {
  const f32x2 x = {x0, x1};
  const f16x2 hh = __builtin_convertvector(x, f16x2);
  const f16x2 ll = lo2h(hh, x0, x1);
  hi[j] = hh[0]; hi[j + 1] = hh[1];
  lo[j] = ll[0]; lo[j + 1] = ll[1];
}

// ---- e4m3 planes for the block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) ------------------------------------------------
// Measured on gfx950 (tools/ubench/mfma_scale_probe.hip, fp8_cvt_probe.hip): a lane's 32 operand bytes are k-positions
// (half h = lane >> 5, byte b); byte b of half h of A meets byte b of half h of B; bytes 0..15 of BOTH halves form scale block 0,
// whose E8M0 scale is the selected byte of lanes 0..31's scale register (lane = the operand's row / column), bytes 16..31 form
// block 1, scaled by lanes 32..63's.  The conversions compute e4m3(x / scale), round to nearest even, NaN beyond 448; the
// decode multiplies by the scale.  32 f16 MFMAs + 4 of these take 0.80 of the time of 48 f16 MFMAs (the clock rises as well).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
template <bool HI_WORD>
GMF_DEVINL int cvt2_fp8_f16(int d, f16x2 v, float scale) {       // e4m3(v / scale) for a pair, into one 16-bit half of d
  return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(__builtin_bit_cast(i16x2, d), v, scale, HI_WORD));
}
template <bool HI_WORD>
GMF_DEVINL int cvt2_fp8_f32(int d, float x0, float x1, float scale) {
  return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(i16x2, d), x0, x1, scale, HI_WORD));
}
template <bool HI_WORD>
GMF_DEVINL f16x2 dec2_fp8_f16(int d, float scale) {               // the two e4m3 values of one half of d, times scale (exact in fp16)
  return __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d, scale, HI_WORD);
}
GMF_DEVINL f32x16 mfma_f8s(i32x8 a, i32x8 b, f32x16 c, int opsel_a, int scale_a, int scale_b) {
  switch (opsel_a) {      // (the byte select is an immediate)
    case 0: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    case 1: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, scale_a, 0, scale_b);
    case 2: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, scale_a, 0, scale_b);
    default: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, scale_a, 0, scale_b);
  }
}

// ... with a byte select on BOTH scale registers (the same byte: block `sel` of A's and of B's scale word)
GMF_DEVINL f32x16 mfma_f8s2(i32x8 a, i32x8 b, f32x16 c, int sel, int scale_a, int scale_b) {
  switch (sel) {
    case 0: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    case 1: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, scale_a, 1, scale_b);
    case 2: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, scale_a, 2, scale_b);
    default: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, scale_a, 3, scale_b);
  }
}

// [r5] PointDSC's learnable sigma (PointDSC.py:164) either by value (the host read it) or from a device address
// (gmf_set_sigma_device: a training step whose optimizer updates sigma on the device, captured in a HIP graph).  The same
// correctly rounded fp32 operations as the host's 1.0f / (sigma * sigma): both forms give the same bits.
GMF_DEVINL float sigma_inv2(float inv_by_value, const float* __restrict__ sigma_dev) {
  if (!sigma_dev) return inv_by_value;
  const float sg = *sigma_dev;
  return 1.0f / (sg * sg);
}
GMF_DEVINL float sigma_two_inv3(float by_value, const float* __restrict__ sigma_dev) {
  if (!sigma_dev) return by_value;
  const float sg = *sigma_dev;
  return 2.0f / (sg * sg * sg);
}

GMF_DEVINL void mma3(f32x16& acc, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
  acc = mfma_h16(al, bh, acc);
  acc = mfma_h16(ah, bl, acc);
  acc = mfma_h16(ah, bh, acc);
}

// Sticky status word of the handle (host-mapped, gmf_status_read): OR `bits` into it when any lane of the wave saw `bad`.
// One system-scope atomic per wave and only in the failure case: the good path costs a vote.
GMF_DEVINL void flag_status(int* status, bool bad, int bits) {
  if (status && __any(bad)) {
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
      __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// ReLU that lets a NaN THROUGH (fmaxf / v_max_f32 return the other operand): an activation that left the fp16 range of the split
// operands becomes (inf, -inf) planes and NaN in the next product; with fmaxf the next ReLU turned that NaN into 0 and the
// overflow passed silently.  One compare + select instead of one max, in the epilogues only.
GMF_DEVINL float relu_nan(float x) { return x < 0.f ? 0.f : x; }
GMF_DEVINL bool not_finite(float v) { return !(__builtin_fabsf(v) <= 3.4028234e38f); }

GMF_DEVINL f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// acc += Wimg(32 x K) * X^T   ->  Y^T block (rows on lanes).  lw = LDS image + lane.
template <int KF>
GMF_DEVINL void mma_wx(f32x16& acc, const float4* lw, const float (&x)[KF]) {
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) {
    const float4 w = lw[g * 64];
    acc = mfma32(w.x, x[4 * g + 0], acc);
    acc = mfma32(w.y, x[4 * g + 1], acc);
    acc = mfma32(w.z, x[4 * g + 2], acc);
    acc = mfma32(w.w, x[4 * g + 3], acc);
  }
}

// acc += X * Wimg^T  ->  Y block with the output feature on the lane (T layout).
template <int KF>
GMF_DEVINL void mma_xw(f32x16& acc, const float4* lw, const float (&x)[KF]) {
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) {
    const float4 w = lw[g * 64];
    acc = mfma32(x[4 * g + 0], w.x, acc);
    acc = mfma32(x[4 * g + 1], w.y, acc);
    acc = mfma32(x[4 * g + 2], w.z, acc);
    acc = mfma32(x[4 * g + 3], w.w, acc);
  }
}

// fragment element offset helper: r in [0,16) of a 32-wide block -> feature index
GMF_DEVINL int frag_feature(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

// Load a per-feature vector (bias, LN gamma, ...) in fragment order:
//   v[4*g+e] = vec[8g + 4h + e]
template <int KF>
GMF_DEVINL void load_vec_frag(float (&v)[KF], const float* __restrict__ vec, int h) {
  const float4* p = reinterpret_cast<const float4*>(vec) + h;
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) {
    const float4 t = p[2 * g];
    v[4 * g + 0] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
  }
}

// P32 tile load / store.  tile_base points at the tile (32*K floats).
template <int KF>
GMF_DEVINL void load_frag_p32(float (&x)[KF], const float* __restrict__ tile_base, int lane) {
  const float4* p = reinterpret_cast<const float4*>(tile_base) + lane;
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) {
    const float4 t = p[g * 64];
    x[4 * g + 0] = t.x; x[4 * g + 1] = t.y; x[4 * g + 2] = t.z; x[4 * g + 3] = t.w;
  }
}

template <int KF>
GMF_DEVINL void store_frag_p32(float* __restrict__ tile_base, const float (&x)[KF], int lane) {
  float4* p = reinterpret_cast<float4*>(tile_base) + lane;
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) p[g * 64] = make_float4(x[4 * g + 0], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
}

// Load row `row` (any row of the same [rows,K] P32 tensor, pair-local index) as a fragment,
// zero if the row is outside [0, n_rows).  Used for the LCPE neighbours i-1 / i+1.
template <int KF>
GMF_DEVINL void load_row_frag_p32(float (&x)[KF], const float* __restrict__ pair_base, int row, int n_rows, int h) {
  if (row >= 0 && row < n_rows) {
    const float4* p = reinterpret_cast<const float4*>(pair_base) + (size_t)(row >> 5) * (KF / 4) * 64 + h * 32 + (row & 31);
#pragma unroll
    for (int g = 0; g < KF / 4; ++g) {
      const float4 t = p[g * 64];
      x[4 * g + 0] = t.x; x[4 * g + 1] = t.y; x[4 * g + 2] = t.z; x[4 * g + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KF; ++k) x[k] = 0.f;
  }
}

// Correctly rounded square root for the squared distances (normal range or exactly 0): v_sqrt_f32 (1 ulp) plus
// one FMA correction step, y' = y + (x - y*y) * (0.5 * rsq(x)).  Agrees with sqrtf bit for bit on 6.7e7 sampled
// squared distances (tools/ubench/sqrt_check.hip; raw v_sqrt_f32 differs on 15 % of them) at 6 instructions
// instead of the ~14 of hipcc's general sqrtf expansion (which also handles denormals and scaling).
GMF_DEVINL float sqrt_cr(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float hr = 0.5f * __builtin_amdgcn_rsqf(fmaxf(x, 1e-36f));
  return fmaf(fmaf(-y, y, x), hr, y);
}


// Sum / max over the two K-halves of a row (lane l and l^32 hold the two halves).
GMF_DEVINL float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }
GMF_DEVINL float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
// max over the two K-halves of a row without the LDS: swap the upper 32 lanes of one copy with the lower 32 of another
GMF_DEVINL float xhalf_max_swap(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// LayerNorm of a row fragment (eps 1e-5, biased variance, two-pass as nn.LayerNorm).
template <int KF>
GMF_DEVINL void layernorm_frag(float (&y)[KF], const float (&x)[KF], const float* __restrict__ gamma,
                               const float* __restrict__ beta, int h) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KF; ++k) s += x[k];
  const float mu = xhalf_sum(s) * (1.0f / (2 * KF));
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < KF; ++k) { const float d = x[k] - mu; v = fmaf(d, d, v); }
  const float rstd = rsqrtf(xhalf_sum(v) * (1.0f / (2 * KF)) + 1e-5f);
  const float4* pg = reinterpret_cast<const float4*>(gamma) + h;
  const float4* pb = reinterpret_cast<const float4*>(beta) + h;
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) {
    const float4 ga = pg[2 * g], be = pb[2 * g];
    y[4 * g + 0] = fmaf((x[4 * g + 0] - mu) * rstd, ga.x, be.x);
    y[4 * g + 1] = fmaf((x[4 * g + 1] - mu) * rstd, ga.y, be.y);
    y[4 * g + 2] = fmaf((x[4 * g + 2] - mu) * rstd, ga.z, be.z);
    y[4 * g + 3] = fmaf((x[4 * g + 3] - mu) * rstd, ga.w, be.w);
  }
}

// ---------------------------------------------------------------------------------
// 16 KiB stage streaming: global (L2) -> LDS by LDS-DMA, double buffered, one barrier
// per stage.  The LDS image of a stage is byte-identical to its global image, which is
// lane-linear per 1 KiB piece, so global_load_lds_dwordx4 applies with no swizzle.
// ---------------------------------------------------------------------------------
GMF_DEVINL void dma_piece_1k(const float* __restrict__ gsrc_piece, float* lds_piece, int lane) {
  __builtin_amdgcn_global_load_lds(
      (const void __attribute__((address_space(1)))*)(gsrc_piece + lane * 4),
      (void __attribute__((address_space(3)))*)(lds_piece), 16, 0, 0);
}

// The same piece with a wave-uniform source base in SGPRs and the lane offset (lane * 16 bytes) as the 32-bit VGPR offset:
// no 64-bit vector address arithmetic per piece.  M0 (the LDS destination base) is saved and restored inside the statement
// because the compiler tracks its own M0 values around its LDS-DMA builtins.  hipcc does not count this load in its
// s_waitcnt bookkeeping: callers wait with their own vmcnt before a barrier, as for the builtin form.
GMF_DEVINL void dma_piece_1k_s(const float* __restrict__ gsrc_piece_uniform, float* lds_piece, unsigned lane_off16) {
  unsigned keep;
  const unsigned lds_dst = (unsigned)(uintptr_t)(void __attribute__((address_space(3)))*)(lds_piece);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off16), "s"(gsrc_piece_uniform), "s"(lds_dst) : "memory");
}

// FOUR consecutive KiB (a wave's quarter of a 16 KiB stage) under ONE M0 / address setup: the instruction's 13-bit immediate offset is
// added to the global address AND to the LDS address, so pieces 1..3 are the same instruction with offset:1024 / 2048 / 3072 - 4 vector-memory
// and 4 scalar instructions per wave and stage where four dma_piece_1k_s cost 4 + ~32 (a wave issues nothing else while it issues these:
// tools/ubench/mfma_valu_overlap.hip).  Which wave fetches which KiB of a stage is immaterial: a stage is read behind a barrier.
GMF_DEVINL void dma_4k_s(const float* __restrict__ gsrc_uniform, float* lds_dst_base, unsigned lane_off16) {
  unsigned keep;
  const unsigned lds_dst = (unsigned)(uintptr_t)(void __attribute__((address_space(3)))*)(lds_dst_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off16), "s"(gsrc_uniform), "s"(lds_dst) : "memory");
}

// Copy a per-feature vector block of n_floats (a multiple of 4) to the LDS, 1 KiB pieces round-robin over the waves; the last
// piece is cut by the execution mask (nothing is read behind the block).
GMF_DEVINL void dma_vec(const float* __restrict__ gsrc, float* lds_dst, int n_floats, int wave, int n_waves, int lane) {
  for (int p = wave; p * 256 < n_floats; p += n_waves)
    if (p * 256 + lane * 4 < n_floats) dma_piece_1k(gsrc + p * 256, lds_dst + p * 256, lane);
}

// Copy `n_pieces` KiB from gsrc to lds_dst, pieces distributed round-robin over the waves.
GMF_DEVINL void dma_issue(const float* __restrict__ gsrc, float* lds_dst, int n_pieces, int wave, int n_waves, int lane) {
  for (int p = wave; p < n_pieces; p += n_waves) dma_piece_1k(gsrc + p * 256, lds_dst + p * 256, lane);
}

// Streams 16 KiB stages through two LDS buffers.  The stage sequence is the concatenation of up
// to three contiguous global blobs (e.g. projection weights | per-tile context images | output
// weights), consumed strictly in order by every wave of the workgroup.
struct StageStream {
  const float* seg_ptr[3];
  int seg_end[3];   // cumulative stage counts
  int stage_floats = kStageFloats;   // 16 KiB unless the user sets another stage size (match_kernels.hip)
  float* buf0;
  float* buf1;
  int issued;       // stages issued so far
  int consumed;     // stages handed to the consumer so far
  int total;
  int wave, n_waves, lane;

  GMF_DEVINL void init(float* b0, float* b1, int wave_, int n_waves_, int lane_, const float* p0, int n0,
                       const float* p1 = nullptr, int n1 = 0, const float* p2 = nullptr, int n2 = 0) {
    seg_ptr[0] = p0; seg_ptr[1] = p1; seg_ptr[2] = p2;
    seg_end[0] = n0; seg_end[1] = n0 + n1; seg_end[2] = n0 + n1 + n2;
    total = seg_end[2];
    buf0 = b0; buf1 = b1; issued = 0; consumed = 0;
    wave = wave_; n_waves = n_waves_; lane = lane_;
  }
  GMF_DEVINL void issue_one() {
    if (issued < total) {
      const float* g;
      if (issued < seg_end[0]) g = seg_ptr[0] + (size_t)issued * stage_floats;
      else if (issued < seg_end[1]) g = seg_ptr[1] + (size_t)(issued - seg_end[0]) * stage_floats;
      else g = seg_ptr[2] + (size_t)(issued - seg_end[1]) * stage_floats;
      dma_issue(g, (issued & 1) ? buf1 : buf0, stage_floats / 256, wave, n_waves, lane);
      ++issued;
    }
  }
  // Call once before the first acquire() (after a barrier if the buffers were in use): stage 0 in flight.
  GMF_DEVINL void prime() { issue_one(); }
  // Wait for the next stage, make it visible to the whole workgroup, start the one after it,
  // and return the LDS image (+lane, in float4 units) of the stage to consume now.
  GMF_DEVINL const float4* acquire() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed in LDS
    __syncthreads();                                  // everyone's pieces landed; prior readers of the other buffer done
    const float* cur = (consumed & 1) ? buf1 : buf0;
    ++consumed;
    issue_one();
    return reinterpret_cast<const float4*>(cur) + lane;
  }
};

// StageStream with a deeper ring: NBUF slots of 16 KiB, NBUF - 1 stages in flight, for 4-wave workgroups (4 pieces per
// wave and stage).  One stage of look-ahead (StageStream) covers an L2 hit only when a stage carries >= ~1000 cycles of
// MFMAs; the fp16x2 stages carry 24 MFMAs (~800 cycles), so the h2 kernels stalled on every acquire.  acquire() waits
// with vmcnt(4 * (NBUF - 2)): everything older than this wave's pieces of the NBUF - 2 younger stages has landed (other,
// younger vector-memory operations only make the wait more conservative).
template <int NBUF>
struct StageRing {
  const float* seg_ptr[4];
  int seg_end[4];
  float* base;
  int issued, consumed, total;
  int wave, lane;

  GMF_DEVINL void init(float* lds_base, int wave_, int lane_, const float* p0, int n0, const float* p1 = nullptr,
                       int n1 = 0, const float* p2 = nullptr, int n2 = 0, const float* p3 = nullptr, int n3 = 0) {
    seg_ptr[0] = p0; seg_ptr[1] = p1; seg_ptr[2] = p2; seg_ptr[3] = p3;
    seg_end[0] = n0; seg_end[1] = n0 + n1; seg_end[2] = n0 + n1 + n2; seg_end[3] = n0 + n1 + n2 + n3;
    total = seg_end[3];
    base = lds_base; issued = 0; consumed = 0;
    wave = wave_; lane = lane_;
  }
  GMF_DEVINL void issue_one() {
    if (issued < total) {
      const float* g;
      if (issued < seg_end[0]) g = seg_ptr[0] + (size_t)issued * kStageFloats;
      else if (issued < seg_end[1]) g = seg_ptr[1] + (size_t)(issued - seg_end[0]) * kStageFloats;
      else if (issued < seg_end[2]) g = seg_ptr[2] + (size_t)(issued - seg_end[1]) * kStageFloats;
      else g = seg_ptr[3] + (size_t)(issued - seg_end[2]) * kStageFloats;
      float* dst = base + (issued % NBUF) * kStageFloats;
      // The statement form of the DMA (not the builtin): hipcc waits with vmcnt(0) before every barrier / LDS read that follows
      // a builtin LDS-DMA - for the look-ahead stages and for every store in flight as well - which turned each stage of a
      // one-wave-per-SIMD kernel into a full memory round trip (~1 us whatever it multiplied).  acquire() carries the waits.
      dma_4k_s(g + wave * 1024, dst + wave * 1024, (unsigned)lane * 16u);
      ++issued;
    }
  }
  GMF_DEVINL void prime() {
#pragma unroll
    for (int k = 0; k < NBUF - 1; ++k) issue_one();
  }
  // acquire with a caller-counted wait.  CONTRACT: YOUNGER must be a LOWER bound (too small only waits longer; too large lets a
  // stage through before it has landed) - each call site states how it counts next to the call, and tools/soak_determinism.py
  // (bitwise-equal results over hundreds of runs) is the standing check.  YOUNGER is a lower bound of the vector-memory operations (DMA pieces of later stages,
  // loads, stores - vmcnt counts them all, in issue order) this wave has issued after the pieces of the stage being acquired.
  // Lets stores and later stages stay in flight across the stage barrier.
  template <int YOUNGER>
  GMF_DEVINL const float4* acquire_counted() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    __syncthreads();
    const float* cur = base + (consumed % NBUF) * kStageFloats;
    ++consumed;
    issue_one();
    return reinterpret_cast<const float4*>(cur) + lane;
  }
  GMF_DEVINL const float4* acquire() {
    if (issued - consumed == NBUF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NBUF - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // everyone's pieces of this stage landed; the slot of the previous stage is free
    const float* cur = base + (consumed % NBUF) * kStageFloats;
    ++consumed;
    issue_one();
    return reinterpret_cast<const float4*>(cur) + lane;
  }
};

}  // namespace gmf
