// FusionLayer / PerceiverIO kernels for the DGR bottleneck configuration
//   latent_dim (query width) = 256, context dim = 128, one head of d_head = 128, GEGLU hidden 2 x 1024
// (reference: GMF_DeepGlobalRegistration/*/model/resunet_new.py:516-525,660,694-705 calling
//  model/perceiver_io.py:139-221; note to_out maps the head back to the QUERY width, perceiver_io.py:83).
//
// Same "rows on lanes" scheme and the same stage streaming as encoder_kernels.hip; a 256-wide row is a
// 128-register fragment, so these kernels run one wave per SIMD (<= 512 VGPRs) instead of two.
//   k_ctx_prep_w     perceiver_io.py:126-128,46-49,89-91    LCPE(content) + LayerNorm_ctx + to_kv (128 -> 128 | 128)
//   k_fusion_attn_w  perceiver_io.py:121-123,44,87-101,208  LCPE(q) + LayerNorm + to_q + softmax(QK^T)V + to_out + residual
//   k_fusion_ff_w    perceiver_io.py:54-69,211              LayerNorm + Linear(256,2048) + GEGLU + Linear(1024,256) + residual
#include <type_traits>
#include "enc_common.hpp"
#include "launchers.hpp"

namespace gmf {

namespace wide {

constexpr int CX = 128;        // context (image token) width
constexpr int CXF = CX / 2;
constexpr int LAT = 256;       // query / latent width
constexpr int LATF = LAT / 2;
constexpr int DHW = 128;       // head width
constexpr int DHWF = DHW / 2;
constexpr int FFHW = 4 * LAT;  // GEGLU hidden width (value half)
constexpr int kWaves = 4;
#ifndef GMF_RING_W
#define GMF_RING_W 4
#endif
constexpr int kRingW = GMF_RING_W;   // LDS ring depth of the split-fp16 kernels' stage streams (8 measured no faster: the stages are not what these kernels wait for)

GMF_DEVINL void load_vec16(float (&v)[16], const float* __restrict__ vec, int mb, int h) {
  const float4* p = reinterpret_cast<const float4*>(vec + 32 * mb) + h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = p[2 * q];
    v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}

// block mb (32 features) of a K-wide P32 tile
template <int K>
GMF_DEVINL void store_blk(float* __restrict__ tile_base, int mb, const float (&t)[16], int lane) {
  float4* p = reinterpret_cast<float4*>(tile_base) + (4 * mb) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q * 64] = make_float4(t[4 * q + 0], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}

template <int K>
GMF_DEVINL void load_blk(float (&t)[16], const float* __restrict__ tile_base, int mb, int lane) {
  const float4* p = reinterpret_cast<const float4*>(tile_base) + (4 * mb) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = p[q * 64];
    t[4 * q + 0] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
  }
}

GMF_DEVINL void store_timg(float* __restrict__ tile_base, int db, const f32x16& a, int lane) {
  float4* p = reinterpret_cast<float4*>(tile_base) + (4 * db) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q * 64] = make_float4(a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// LCPE over a K-wide P32 tensor (K = 2*KF): taps = w0[K] | w1[K] | w2[K] | b[K]
template <int KF>
GMF_DEVINL void lcpe_w(float (&y)[KF], const float* __restrict__ pair_base, int row, int n_rows,
                       const float* __restrict__ taps, int h) {
  constexpr int K = 2 * KF;
  const float4* base = reinterpret_cast<const float4*>(pair_base);
  const bool has_m = row >= 1, has_p = row + 1 < n_rows;
  const int rm = has_m ? row - 1 : row, rp = has_p ? row + 1 : row;
  const float4* pc = base + (size_t)(row >> 5) * (KF / 4) * 64 + h * 32 + (row & 31);
  const float4* pm = base + (size_t)(rm >> 5) * (KF / 4) * 64 + h * 32 + (rm & 31);
  const float4* pp = base + (size_t)(rp >> 5) * (KF / 4) * 64 + h * 32 + (rp & 31);
  const float4* t0 = reinterpret_cast<const float4*>(taps) + h;
  const float fm = has_m ? 1.f : 0.f, fp = has_p ? 1.f : 0.f;
#pragma unroll
  for (int g = 0; g < KF / 4; ++g) {
    const float4 xc = pc[g * 64], xm = pm[g * 64], xp = pp[g * 64];
    const float4 w0 = t0[2 * g], w1 = t0[2 * g + K / 4], w2 = t0[2 * g + 2 * (K / 4)], b = t0[2 * g + 3 * (K / 4)];
    y[4 * g + 0] = xc.x + b.x + w0.x * (fm * xm.x) + w1.x * xc.x + w2.x * (fp * xp.x);
    y[4 * g + 1] = xc.y + b.y + w0.y * (fm * xm.y) + w1.y * xc.y + w2.y * (fp * xp.y);
    y[4 * g + 2] = xc.z + b.z + w0.z * (fm * xm.z) + w1.z * xc.z + w2.z * (fp * xp.z);
    y[4 * g + 3] = xc.w + b.w + w0.w * (fm * xm.w) + w1.w * xc.w + w2.w * (fp * xp.w);
  }
}

// acc += Wimg(32 x 8*NG) * x[off .. off + 4*NG) : one K-slice of a wider product
template <int NG, int KF>
GMF_DEVINL void mma_slice(f32x16& acc, const float4* lw, const float (&x)[KF], int off) {
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float4 w = lw[g * 64];
    acc = mfma32(w.x, x[off + 4 * g + 0], acc);
    acc = mfma32(w.y, x[off + 4 * g + 1], acc);
    acc = mfma32(w.z, x[off + 4 * g + 2], acc);
    acc = mfma32(w.w, x[off + 4 * g + 3], acc);
  }
}

GMF_DEVINL float gelu_erf_w(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

}  // namespace wide

using namespace wide;

// ---------------------------------------------------------------------------------------------
// output per token tile: 8192 floats = Kc as P32 (K=128) | Vc as T image (D=128)
// stages (8): Wk[4] | Wv[4]        vecs: content taps w0|w1|w2|b (4x128) | gamma_c | beta_c
// ---------------------------------------------------------------------------------------------
template <bool PE>
__global__ void __launch_bounds__(256, 1)
k_ctx_prep_w(const float* __restrict__ ctx, const float* __restrict__ wst, const float* __restrict__ vecs,
             float* __restrict__ out, int T, int ttiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWaves + wave;
  const bool active = tile_raw < ttiles;
  const int tile = active ? tile_raw : ttiles - 1;
  const float* pair_base = ctx + (size_t)pair * ttiles * (32 * CX);
  float* dst = out + ((size_t)pair * ttiles + tile) * (2 * kStageFloats);

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWaves, lane, wst, 8);
  ss.prime();
  float x[CXF], cn[CXF];
  if (PE) lcpe_w<CXF>(x, pair_base, tile * 32 + i, T, vecs, h);
  else load_frag_p32<CXF>(x, pair_base + (size_t)tile * (32 * CX), lane);
  layernorm_frag<CXF>(cn, x, vecs + 4 * CX, vecs + 5 * CX, h);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<CXF>(acc, lw, cn);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r];
    if (active) store_blk<DHW>(dst, mb, t, lane);
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_xw<CXF>(acc, lw, cn);
    if (active) store_timg(dst + kStageFloats, db, acc, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// stages: Wq'[4 blocks x 2 K-halves] | ctx tiles [2 per tile: K, V] | Wo[8]
// vecs: query taps w0|w1|w2|b (4x256) | gamma[256] | beta[256] | bo[256]
// x, x1: P32 images with K = 256 (8192 floats per tile)
// ---------------------------------------------------------------------------------------------
template <bool PE>
__global__ void __launch_bounds__(256, 1)
k_fusion_attn_w(const float* __restrict__ xin, const float* __restrict__ ctx_img, const float* __restrict__ wst,
                const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWaves + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const float* pair_base = xin + (size_t)pair * tiles * (32 * LAT);
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * LAT);

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWaves, lane, wst, 8,
          ctx_img + (size_t)pair * ttiles * (2 * kStageFloats), 2 * ttiles, wst + 8 * kStageFloats, 8);
  ss.prime();

  float xp[LATF];
  if (PE) lcpe_w<LATF>(xp, pair_base, tile * 32 + i, N, vecs, h);
  else load_frag_p32<LATF>(xp, pair_base + (size_t)tile * (32 * LAT), lane);

  float qf[DHWF];
  {
    float xn[LATF];
    layernorm_frag<LATF>(xn, xp, vecs + 4 * LAT, vecs + 5 * LAT, h);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      f32x16 acc = zero16();
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const float4* lw = ss.acquire();
        mma_slice<16, LATF>(acc, lw, xn, 64 * half);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) qf[16 * mb + r] = acc[r];
    }
  }

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  for (int t = 0; t < ttiles; ++t) {
    f32x16 s = zero16();
    {
      const float4* lk = ss.acquire();
      mma_wx<DHWF>(s, lk, qf);
    }
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
    const float4* lv = ss.acquire();
#pragma unroll
    for (int db = 0; db < 4; ++db) {
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
  float o[DHWF];
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[16 * db + r] = oacc[db][r] * inv;
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const float4* lw = ss.acquire();
    f32x16 acc = zero16();
    mma_wx<DHWF>(acc, lw, o);
    float b[16], t[16];
    load_vec16(b, vecs + 6 * LAT, mb, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] + b[r] + xp[16 * mb + r];
    if (active) store_blk<LAT>(x1_out + toff, mb, t, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// stages (192): for c in 0..31: W1a_c [2 K-halves] | W1g_c [2] | W2_c [2: out-blocks 0-3, 4-7]
// vecs: gamma[256] | beta[256] | b1a[1024] | b1g[1024] | b2[256]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 1)
k_fusion_ff_w(const float* __restrict__ x1, const float* __restrict__ wst, const float* __restrict__ vecs,
              float* __restrict__ x2_out, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageFloats];
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWaves + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * LAT);

  StageStream ss;
  ss.init(lds, lds + kStageFloats, wave, kWaves, lane, wst, 6 * (FFHW / 32));
  ss.prime();
  float xn[LATF];
  {
    float x[LATF];
    load_frag_p32<LATF>(x, x1 + toff, lane);
    layernorm_frag<LATF>(xn, x, vecs, vecs + LAT, h);
  }
  f32x16 y[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) y[mb] = zero16();
  const float* b1a = vecs + 2 * LAT;
  const float* b1g = vecs + 2 * LAT + FFHW;

  for (int c = 0; c < FFHW / 32; ++c) {
    float ga[16];
    {
      f32x16 acc = zero16();
#pragma unroll
      for (int half = 0; half < 2; ++half) { const float4* lw = ss.acquire(); mma_slice<16, LATF>(acc, lw, xn, 64 * half); }
      float b[16];
      load_vec16(b, b1a, c, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] = acc[r] + b[r];
    }
    {
      f32x16 acc = zero16();
#pragma unroll
      for (int half = 0; half < 2; ++half) { const float4* lw = ss.acquire(); mma_slice<16, LATF>(acc, lw, xn, 64 * half); }
      float b[16];
      load_vec16(b, b1g, c, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] *= gelu_erf_w(acc[r] + b[r]);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float4* lw = ss.acquire();
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) mma_wx<16>(y[4 * half + m4], lw + m4 * (32 * 32 / 4), ga);
    }
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    float b[16], xr[16], t[16];
    load_vec16(b, vecs + 2 * LAT + 2 * FFHW, mb, h);
    load_blk<LAT>(xr, x1 + toff, mb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = y[mb][r] + b[r] + xr[r];
    if (active) store_blk<LAT>(x2_out + toff, mb, t, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// k_fusion_ff_w_h2: the 256-wide GEGLU feed-forward on the f16 MFMA with split-fp16 operands (5.3x fewer matrix cycles than
// k_fusion_ff_w; arithmetic notes in mfma_core.hpp).  Weight images from packing.p32_h2, same blob size and stage count:
//   for c in 0..31: W1a_c hi plane | W1a_c lo plane | W1g_c hi | W1g_c lo | W2_c out-blocks 0-3 | W2_c out-blocks 4-7
//   (a 32 x 256 block is [plane][16 k-steps][lane][8 halfs]: the hi-plane stage carries the products ah*xl and ah*xh, the
//   lo-plane stage al*xh).  Single-range GELU (enc_common.hpp).  One wave per SIMD: x (128 VGPRs as two fp16 planes) and
//   the eight output accumulators (128) stay in registers.
// ---------------------------------------------------------------------------------------------
// Small grids (M < ~6000 voxels is the usual size of the DGR bottleneck): gridDim.z = hs workgroups per row block take
// 32 / hs hidden chunks each and store their raw partial sums to part[z]; k_ff_reduce_w adds them in index order.
__global__ void __launch_bounds__(256, 1)
k_fusion_ff_w_h2(const float* __restrict__ x1, const float* __restrict__ wst, const float* __restrict__ vecs,
                 float* __restrict__ x2_out, int tiles, float* __restrict__ part) {
  using namespace wide;
  // LDS: the stage ring | the per-feature vectors (gamma | beta | b1 value | b1 gate | b2: 11 KiB) - a bias fetched from global
  // memory at the top of every hidden chunk was a memory round trip per chunk with nothing to hide behind (one wave per SIMD)
  __shared__ __attribute__((aligned(16))) float lds[kRingW * kStageFloats + 2 * LAT + 2 * FFHW + LAT];
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWaves + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * LAT);

  const int hs = gridDim.z, n_chunks = (FFHW / 32) / hs, c0 = blockIdx.z * n_chunks;
  float* const lvec = lds + kRingW * kStageFloats;
  dma_issue(vecs, lvec, 11, wave, kWaves, lane);
  StageRing<kRingW> ss;
  ss.init(lds, wave, lane, wst + (size_t)c0 * 6 * kStageFloats, 6 * n_chunks);
  ss.prime();
  FragH2<16> nx;
  {
    float x[LATF], xn[LATF];
    load_frag_p32<LATF>(x, x1 + toff, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    layernorm_frag<LATF>(xn, x, lvec, lvec + LAT, h);
    nx.set(xn);
  }
  f32x16 y[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) y[mb] = zero16();
  const float* b1a = lvec + 2 * LAT;
  const float* b1g = lvec + 2 * LAT + FFHW;

  // acc (preloaded with the bias) += W(32 x 256) x^T from the two plane stages of one weight block
  auto w1_block = [&](f32x16& acc) {
    const f16x8* lw = as_h2(ss.acquire());            // hi plane, 16 k-steps
#pragma unroll
    for (int s = 0; s < 16; ++s) { const f16x8 wh = lw[s * 64]; acc = mfma_h16(wh, nx.l[s], acc); acc = mfma_h16(wh, nx.h[s], acc); }
    lw = as_h2(ss.acquire());                         // lo plane
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma_h16(lw[s * 64], nx.h[s], acc);
  };
  for (int c = c0; c < c0 + n_chunks; ++c) {
    float ga[16];
    {
      float b[16];
      load_vec16(b, b1a, c, h);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = b[r];
      w1_block(acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] = acc[r];
    }
    {
      float b[16];
      load_vec16(b, b1g, c, h);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = b[r];
      w1_block(acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] *= gelu_erf_1r(acc[r]);
    }
    FragH2<2> gx;
    gx.set(ga);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f16x8* lw = as_h2(ss.acquire());
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) mma_wx_h2<2>(y[4 * half + m4], lw + m4 * (2 * 2 * 64), gx);
    }
  }
  if (hs > 1) {
    float* dst = part + (size_t)blockIdx.z * ((size_t)gridDim.y * tiles * (32 * LAT)) + toff;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
      if (active) store_timg(dst, mb, y[mb], lane);          // (same 16-byte units as store_blk)
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    float b[16], xr[16], t[16];
    load_vec16(b, lvec + 2 * LAT + 2 * FFHW, mb, h);
    load_blk<LAT>(xr, x1 + toff, mb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(y[mb][r], kH2Inv, b[r]) + xr[r];
    if (active) store_blk<LAT>(x2_out + toff, mb, t, lane);
  }
}

// x2 = sum_z part[z] * 2^-8 + b2 + x1 (z in index order).  grid (ceil(tiles/4), B, 8): a wave adds one 32 x 32 block.
// out_rm (may be null): the caller's strided [B, n_rows, 256] output - written here instead of the image + an unpacking pass.
__global__ void __launch_bounds__(256)
k_ff_reduce_w(const float* __restrict__ part, const float* __restrict__ x1, const float* __restrict__ vecs,
              float* __restrict__ x2_out, int tiles, int hs, float* __restrict__ out_rm, long o_sb, long o_sr, long o_sk, int n_rows,
              int* status) {
  using namespace wide;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int tile = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (tile >= tiles) return;
  const int mb = blockIdx.z;
  const size_t n_all = (size_t)gridDim.y * tiles * (32 * LAT);
  const size_t toff = ((size_t)blockIdx.y * tiles + tile) * (32 * LAT);
  float b[16], xr[16], p[8][16];
  load_vec16(b, vecs + 2 * LAT + 2 * FFHW, mb, h);
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < hs) load_blk<LAT>(p[z], part + (size_t)z * n_all + toff, mb, lane);
  load_blk<LAT>(xr, x1 + toff, mb, lane);
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
  if (out_rm) {
    const int row = tile * 32 + (lane & 31);
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) bad = bad || not_finite(t[r]);
    flag_status(status, bad && row < n_rows, 1);
    if (row < n_rows) {
      float* p = out_rm + (long)blockIdx.y * o_sb + (long)row * o_sr;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) p[(long)(32 * mb + 8 * q + 4 * h + e) * o_sk] = t[4 * q + e];
    }
    return;
  }
  store_blk<LAT>(x2_out + toff, mb, t, lane);
}

// ---------------------------------------------------------------------------------------------
// k_ctx_prep_w_h2 / k_fusion_attn_w_h2: the context preparation and the cross-attention of the 256-wide layer on the f16 MFMA
// with split-fp16 operands, like its feed-forward above (same arithmetic as k_ctx_prep_h2 / k_fusion_attn_h2 of the
// 128-wide layer, encoder_h2.hip; 5.3x fewer matrix cycles than k_ctx_prep_w / k_fusion_attn_w).  Weight images from
// packing.p32_h2s (256 W; 2^-8 folded into the epilogues), same blob sizes and stage counts as the fp32 images:
//   ctx : Wk[4] | Wv[4]                     a 32 x 128 block is one stage [plane][8 k-steps][lane][8 halfs]
//   attn: Wq''[4 x (hi plane | lo plane)]   a 32 x 256 block is two stages [16 k-steps][lane][8 halfs] (hi: wh*xl + wh*xh, lo: wl*xh)
//         | Wo[8]                           32 x 128 blocks
// Context image per token tile (2 stages, as before): Kc | Vc as fp16x2 images for d_head = 128:
//   Kc unit ((plane*8 + s)*64 + lane), Vc unit ((plane*8 + slot)*64 + lane), slot = 2*db + s2.
// ---------------------------------------------------------------------------------------------
// ROWMAJOR: `ctx` is the caller's [B, T, 128] tensor itself (no packing pass in front of this 3-workgroup kernel).
template <bool PE, bool ROWMAJOR>
__global__ void __launch_bounds__(256, 1)
k_ctx_prep_w_h2(const float* __restrict__ ctx, const float* __restrict__ wst, const float* __restrict__ vecs,
                float* __restrict__ out, int T, int ttiles) {
  __shared__ __attribute__((aligned(16))) float lds[kRingW * kStageFloats + 6 * CX + kWaves * 2 * CX];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWaves + wave;
  const bool active = tile_raw < ttiles;
  const int tile = active ? tile_raw : ttiles - 1;
  const float* pair_base = ROWMAJOR ? ctx + (size_t)pair * T * CX : ctx + (size_t)pair * ttiles * (32 * CX);
  float* dst = out + ((size_t)pair * ttiles + tile) * (2 * kStageFloats);
  float* const lvec = lds + kRingW * kStageFloats;          // taps (4 x 128) | gamma | beta: 3 KiB
  float* const halo = lvec + 6 * CX + wave * (2 * CX);

  dma_issue(vecs, lvec, 3, wave, kWaves, lane);
  if (PE) {
    if (ROWMAJOR) LcpeHalo<CXF>::issue_rowmajor(pair_base, tile, T, halo, lane);
    else LcpeHalo<CXF>::issue(pair_base, tile, ttiles, halo, lane);
  }
  StageRing<kRingW> ss;
  ss.init(lds, wave, lane, wst, 8);
  ss.prime();
  FragH2<8> cx;
  {
    float x[CXF], cn[CXF];
    if (ROWMAJOR) {
      const int row = tile * 32 + i;
      const float4* p = reinterpret_cast<const float4*>(pair_base + (size_t)min(row, T - 1) * CX) + h;
#pragma unroll
      for (int g = 0; g < CXF / 4; ++g) {           // fragment group g = features 32 (g >> 2) + 8 (g & 3) + 4 h .. + 3
        const float4 t = p[8 * (g >> 2) + 2 * (g & 3)];
        const bool ok = row < T;
        x[4 * g + 0] = ok ? t.x : 0.f; x[4 * g + 1] = ok ? t.y : 0.f; x[4 * g + 2] = ok ? t.z : 0.f; x[4 * g + 3] = ok ? t.w : 0.f;
      }
    } else {
      load_frag_p32<CXF>(x, pair_base + (size_t)tile * (32 * CX), lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (PE) LcpeHalo<CXF>::apply(x, halo, lvec, tile * 32 + i, T, lane);
    layernorm_frag<CXF>(cn, x, lvec + 4 * CX, lvec + 5 * CX, h);
    cx.set(cn);
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const f16x8* lw = as_h2(ss.acquire());
    f32x16 acc = zero16();
    mma_wx_h2<8>(acc, lw, cx);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
    if (active) store_block_h2(dst, mb, t, lane);
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const f16x8* lw = as_h2(ss.acquire());
    f32x16 acc = zero16();
    mma_xw_h2<8>(acc, lw, cx);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
    if (active) store_block_h2(dst + kStageFloats, db, t, lane);
  }
}

template <bool PE>
__global__ void __launch_bounds__(256, 1)
k_fusion_attn_w_h2(const float* __restrict__ xin, const float* __restrict__ ctx_img, const float* __restrict__ wst,
                   const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles) {
  // LDS: the stage ring | the kernel's per-feature vectors (7 KiB) | per wave the two halo rows of the LCPE (2 KiB)
  __shared__ __attribute__((aligned(16))) float lds[kRingW * kStageFloats + 7 * LAT + kWaves * 2 * LAT];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int tile_raw = blockIdx.x * kWaves + wave;
  const bool active = tile_raw < tiles;
  const int tile = active ? tile_raw : tiles - 1;
  const float* pair_base = xin + (size_t)pair * tiles * (32 * LAT);
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * LAT);
  float* const lvec = lds + kRingW * kStageFloats;
  float* const halo = lvec + 7 * LAT + wave * (2 * LAT);

  // everything the prologue needs is requested at once (older than the ring's pieces, so the ring's counted waits cover it)
  dma_issue(vecs, lvec, 7, wave, kWaves, lane);
  if (PE) LcpeHalo<LATF>::issue(pair_base, tile, tiles, halo, lane);
  StageRing<kRingW> ss;
  ss.init(lds, wave, lane, wst, 8, ctx_img + (size_t)pair * ttiles * (2 * kStageFloats), 2 * ttiles, wst + 8 * kStageFloats, 8);
  ss.prime();

  // one wave per SIMD (512 VGPRs): x' (128), its two fp16 planes (128) and the Q fragment (64) are live together
  float xp[LATF];
  load_frag_p32<LATF>(xp, pair_base + (size_t)tile * (32 * LAT), lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (PE) LcpeHalo<LATF>::apply(xp, halo, lvec, tile * 32 + i, N, lane);
  FragH2<8> qx;
  {
    FragH2<16> nx;
    {
      float xn[LATF];
      layernorm_frag<LATF>(xn, xp, lvec + 4 * LAT, lvec + 5 * LAT, h);
      nx.set(xn);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      f32x16 acc = zero16();
      const f16x8* lw = as_h2(ss.acquire());              // hi plane, 16 k-steps
#pragma unroll
      for (int s = 0; s < 16; ++s) { const f16x8 wh = lw[s * 64]; acc = mfma_h16(wh, nx.l[s], acc); acc = mfma_h16(wh, nx.h[s], acc); }
      lw = as_h2(ss.acquire());                           // lo plane
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = mfma_h16(lw[s * 64], nx.h[s], acc);
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
      qx.set_block(mb, t);
    }
  }

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  for (int t = 0; t < ttiles; ++t) {
    f32x16 s = zero16();
    {
      const f16x8* lk = as_h2(ss.acquire());
      mma_wx_h2<8>(s, lk, qx);
    }
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
    const float m_off = m_new - 10.0f;                    // P' = 2^10 P: its lo plane stays a normal fp16 number
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = __builtin_amdgcn_exp2f(x[r] - m_off); ls += x[r]; }
    l_half = fmaf(l_half, alpha, ls);
    const f16x8* lv = as_h2(ss.acquire());
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int slot = 2 * db + s2;
        mma3(oacc[db], lv[(0 * 8 + slot) * 64], lv[(1 * 8 + slot) * 64], ph, pl);
      }
    }
  }
  FragH2<8> ox;
  {
    const float inv = 1.0f / xhalf_sum(l_half);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = oacc[db][r] * inv;
      ox.set_block(db, t);
    }
  }
  // to_out: 8 stages, each followed by 4 stores per lane.  The ring's plain acquire() would wait for the previous stage's
  // stores at every barrier (vmcnt counts stores too; a store's acknowledgement takes about as long as a whole stage), so the
  // waits are counted: younger than stage j's pieces are the pieces of the up to two stages behind it and the stores of the up
  // to three stages before it.  (Padding waves store their copy of the last tile as well - same values - so that the count
  // is the same in every wave.)
  auto out_stage = [&](auto jc) {
    constexpr int mb = decltype(jc)::value;
    constexpr int younger = 4 * (mb <= 5 ? 2 : 7 - mb) + 4 * (mb < 3 ? mb : 3);
    const f16x8* lw = as_h2(ss.template acquire_counted<younger>());
    f32x16 acc = zero16();
    mma_wx_h2<8>(acc, lw, ox);
    float b[16], t[16];
    load_vec16(b, lvec + 6 * LAT, mb, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, b[r]) + xp[16 * mb + r];
    store_blk<LAT>(x1_out + toff, mb, t, lane);
  };
  out_stage(std::integral_constant<int, 0>{}); out_stage(std::integral_constant<int, 1>{});
  out_stage(std::integral_constant<int, 2>{}); out_stage(std::integral_constant<int, 3>{});
  out_stage(std::integral_constant<int, 4>{}); out_stage(std::integral_constant<int, 5>{});
  out_stage(std::integral_constant<int, 6>{}); out_stage(std::integral_constant<int, 7>{});
}

// ---------------------------------------------------------------------------------------------
// k_fusion_attn_w_tile: the cross-attention of the 256-wide layer with ONE workgroup per 32-row tile, for grids of up to 256
// tiles (the usual size of the DGR bottleneck: 1000 - 8000 voxels are 32 - 250 tiles, i.e. 8 - 63 workgroups of
// k_fusion_attn_w_h2, each wave walking 36 dependent stages).  The four waves share the tile and split every level:
//   LCPE + LayerNorm   each wave its quarter of the row's 256 features; the LayerNorm statistics are exchanged through the LDS
//   to_q               4 blocks of 32 outputs, one per wave (K = 256)
//   context tiles      dealt round-robin to the waves, partial softmaxes merged through the LDS (as key-split partials)
//   to_out             8 blocks of 32 outputs, two per wave, + bias + the wave's quarter of x'
// No wave shares a weight block with another, so the blocks are loaded straight into registers (the next block requested
// before the current one is multiplied); the LDS only carries the exchanges.  Same arithmetic per product as
// k_fusion_attn_w_h2; the LayerNorm sums and the context tiles are added in another order (fp32 rounding).
// grid (tiles, B), block 256.
// ---------------------------------------------------------------------------------------------
template <bool PE>
__global__ void __launch_bounds__(256, 1)
k_fusion_attn_w_tile(const float* __restrict__ xin, const float* __restrict__ ctx_img, const float* __restrict__ wst,
                     const float* __restrict__ vecs, float* __restrict__ x1_out, int N, int tiles, int T, int ttiles) {
  using namespace wide;
  // LDS: vectors (7 KiB) | halo rows (2 KiB) | LayerNorm partial sums [2][4][64] | exchange: LN'd row image (32 KiB) | q image
  //      (16 KiB); later the same exchange area holds the softmax partials [wave][66][64] (66 KiB)
  __shared__ __attribute__((aligned(16))) float lds[7 * LAT + 2 * LAT + 512 + 4 * 66 * 64];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, pair = blockIdx.y;
  const float* pair_base = xin + (size_t)pair * tiles * (32 * LAT);
  const size_t toff = ((size_t)pair * tiles + tile) * (32 * LAT);
  float* const lvec = lds;
  float* const halo = lvec + 7 * LAT;
  float* const stats = halo + 2 * LAT;
  float* const xch = stats + 512;
  f16x8* const nxh = reinterpret_cast<f16x8*>(xch);                 // [plane][16 steps][64 lanes]
  f16x8* const qh = reinterpret_cast<f16x8*>(xch + 8192);           // [plane][8 steps][64 lanes]
  const float* ctx_pair = ctx_img + (size_t)pair * ttiles * (2 * kStageFloats);
  const int row = tile * 32 + i;
  const int g0 = 8 * wave;                                         // this wave's fragment groups [g0, g0 + 8): 32 of the lane's 128 features

  auto load16 = [&](f16x8 (&dst)[32], const int at, const float* g) {      // one 16 KiB stage: 16 fragments
    const f16x8* p = reinterpret_cast<const f16x8*>(g) + lane;
#pragma unroll
    for (int u = 0; u < 16; ++u) dst[at + u] = p[u * 64];
  };
  f16x8 wa[32], wb[32];

  // ---- prologue requests ----
  dma_issue(vecs, lvec, 7, wave, kWaves, lane);
  if (PE && wave == 0) LcpeHalo<LATF>::issue(pair_base, tile, tiles, halo, lane);
  float xq[32];
  {
    const float4* p = reinterpret_cast<const float4*>(pair_base + (size_t)tile * (32 * LAT)) + lane;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 t = p[(g0 + g) * 64];
      xq[4 * g + 0] = t.x; xq[4 * g + 1] = t.y; xq[4 * g + 2] = t.z; xq[4 * g + 3] = t.w;
    }
  }
  load16(wa, 0, wst + (size_t)(2 * wave) * kStageFloats);           // Wq'' block `wave`: hi plane | lo plane
  load16(wa, 16, wst + (size_t)(2 * wave + 1) * kStageFloats);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                 // (A) vectors and halo rows are in the LDS
  if (wave < ttiles) {                                             // first context tile of this wave: K stage | V stage
    load16(wb, 0, ctx_pair + (size_t)(2 * wave) * kStageFloats);
    load16(wb, 16, ctx_pair + (size_t)(2 * wave + 1) * kStageFloats);
  }
  // ---- LCPE on the quarter (neighbour rows from the neighbour lanes, LcpeHalo) ----
  if (PE) {
    const bool has_m = row >= 1, has_p = row + 1 < N, first = i == 0, last = i == 31;
    const float4* hl = reinterpret_cast<const float4*>(halo) + h;
    const float4* t0 = reinterpret_cast<const float4*>(lvec) + h;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int gg = g0 + g;
      const float4 up = hl[2 * gg], dn = hl[LAT / 4 + 2 * gg];
      const float4 w0 = t0[2 * gg], w1 = t0[2 * gg + LAT / 4], w2 = t0[2 * gg + 2 * (LAT / 4)], b = t0[2 * gg + 3 * (LAT / 4)];
      const float upv[4] = {up.x, up.y, up.z, up.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w};
      const float w0v[4] = {w0.x, w0.y, w0.z, w0.w}, w1v[4] = {w1.x, w1.y, w1.z, w1.w}, w2v[4] = {w2.x, w2.y, w2.z, w2.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xc = xq[4 * g + e];
        float xm = LcpeHalo<LATF>::from_lane_below(xc), xn = LcpeHalo<LATF>::from_lane_above(xc);
        xm = first ? upv[e] : xm;
        xn = last ? dnv[e] : xn;
        xm = has_m ? xm : 0.f;
        xn = has_p ? xn : 0.f;
        xq[4 * g + e] = xc + bv[e] + w0v[e] * xm + w1v[e] * xc + w2v[e] * xn;
      }
    }
  }
  // ---- LayerNorm over the 256 features of a row: two passes, partial sums of the four waves through the LDS ----
  float mu, rstd;
  {
    float sp = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) sp += xq[e];
    sp = xhalf_sum(sp);
    stats[(0 * 4 + wave) * 64 + lane] = sp;
    __syncthreads();                                               // (B)
    mu = ((stats[(0 * 4 + 0) * 64 + lane] + stats[(0 * 4 + 1) * 64 + lane]) + stats[(0 * 4 + 2) * 64 + lane]) + stats[(0 * 4 + 3) * 64 + lane];
    mu *= 1.0f / LAT;
    float vp = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) { const float d = xq[e] - mu; vp = fmaf(d, d, vp); }
    vp = xhalf_sum(vp);
    stats[(1 * 4 + wave) * 64 + lane] = vp;
    __syncthreads();                                               // (C)
    const float var = ((stats[(1 * 4 + 0) * 64 + lane] + stats[(1 * 4 + 1) * 64 + lane]) + stats[(1 * 4 + 2) * 64 + lane]) + stats[(1 * 4 + 3) * 64 + lane];
    rstd = rsqrtf(var * (1.0f / LAT) + 1e-5f);
  }
  {
    const float4* pg = reinterpret_cast<const float4*>(lvec + 4 * LAT) + h;
    const float4* pb = reinterpret_cast<const float4*>(lvec + 5 * LAT) + h;
    float xn[32];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 ga = pg[2 * (g0 + g)], be = pb[2 * (g0 + g)];
      xn[4 * g + 0] = fmaf((xq[4 * g + 0] - mu) * rstd, ga.x, be.x);
      xn[4 * g + 1] = fmaf((xq[4 * g + 1] - mu) * rstd, ga.y, be.y);
      xn[4 * g + 2] = fmaf((xq[4 * g + 2] - mu) * rstd, ga.z, be.z);
      xn[4 * g + 3] = fmaf((xq[4 * g + 3] - mu) * rstd, ga.w, be.w);
    }
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {                               // the quarter is k-steps 4 w .. 4 w + 3 of the row fragment
      f16x8 hi, lo;
      split8h(&xn[8 * ss], hi, lo);
      nxh[(0 * 16 + 4 * wave + ss) * 64 + lane] = hi;
      nxh[(1 * 16 + 4 * wave + ss) * 64 + lane] = lo;
    }
  }
  __syncthreads();                                                 // (D) the LayerNorm'd row is in the LDS
  // ---- to_q, block `wave` (K = 256; the row fragment is read from the LDS step by step) ----
  {
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f16x8 xh = nxh[(0 * 16 + s) * 64 + lane], xl = nxh[(1 * 16 + s) * 64 + lane];
      acc = mfma_h16(wa[s], xl, acc);
      acc = mfma_h16(wa[s], xh, acc);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma_h16(wa[16 + s], nxh[(0 * 16 + s) * 64 + lane], acc);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r] * kH2Inv;
    f16x8 hi, lo;
    split8h(&t[0], hi, lo); qh[(0 * 8 + 2 * wave) * 64 + lane] = hi; qh[(1 * 8 + 2 * wave) * 64 + lane] = lo;
    split8h(&t[8], hi, lo); qh[(0 * 8 + 2 * wave + 1) * 64 + lane] = hi; qh[(1 * 8 + 2 * wave + 1) * 64 + lane] = lo;
  }
  if (wave + 4 < ttiles) {                                         // second context tile of this wave
    load16(wa, 0, ctx_pair + (size_t)(2 * (wave + 4)) * kStageFloats);
    load16(wa, 16, ctx_pair + (size_t)(2 * (wave + 4) + 1) * kStageFloats);
  }
  __syncthreads();                                                 // (E) q is in the LDS
  FragH2<8> qx;
#pragma unroll
  for (int s = 0; s < 8; ++s) { qx.h[s] = qh[(0 * 8 + s) * 64 + lane]; qx.l[s] = qh[(1 * 8 + s) * 64 + lane]; }
  // ---- context tiles wave, wave + 4, ...: [0..15] K (hi | lo planes x 8 steps), [16..31] V (hi | lo planes x 8 slots) ----
  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) oacc[db] = zero16();
  float m_run = -INFINITY, l_half = 0.f;
  auto ctx_tile = [&](const int t, const f16x8 (&w)[32]) {
    f32x16 sc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(sc, w[s], w[8 + s], qx.h[s], qx.l[s]);
    float x[16];
    float mx = -INFINITY;
    const int jbase = t * 32 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = 8 * (r >> 2) + (r & 3);
      x[r] = (jbase + jl < T) ? sc[r] : -INFINITY;
      mx = fmaxf(mx, x[r]);
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
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 ph, pl;
      split8h(&x[8 * s2], ph, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int slot = 2 * db + s2;
        mma3(oacc[db], w[16 + slot], w[24 + slot], ph, pl);
      }
    }
  };
  if (wave < ttiles) ctx_tile(wave, wb);
  if (wave + 8 < ttiles) {
    load16(wb, 0, ctx_pair + (size_t)(2 * (wave + 8)) * kStageFloats);
    load16(wb, 16, ctx_pair + (size_t)(2 * (wave + 8) + 1) * kStageFloats);
  }
  if (wave + 4 < ttiles) ctx_tile(wave + 4, wa);
  load16(wa, 0, wst + (size_t)(8 + 2 * wave) * kStageFloats);      // Wo blocks 2 w, 2 w + 1
  load16(wa, 16, wst + (size_t)(8 + 2 * wave + 1) * kStageFloats);
  if (wave + 8 < ttiles) ctx_tile(wave + 8, wb);
  for (int t = wave + 12; t < ttiles; t += 4) {                    // (more than 384 context tokens)
    load16(wb, 0, ctx_pair + (size_t)(2 * t) * kStageFloats);
    load16(wb, 16, ctx_pair + (size_t)(2 * t + 1) * kStageFloats);
    ctx_tile(t, wb);
  }
  // ---- merge the waves' partial softmaxes (wave order), to_out blocks 2 w and 2 w + 1 ----
  __syncthreads();                                                 // (F) every wave has read q: the exchange area is free
  {
    float* mine = xch + wave * (66 * 64) + lane;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(16 * db + r) * 64] = oacc[db][r];
    mine[64 * 64] = m_run;
    mine[65 * 64] = l_half;
  }
  __syncthreads();                                                 // (G)
  FragH2<8> ox;
  {
    float mw[4], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = xch[w * (66 * 64) + 64 * 64 + lane]; M = fmaxf(M, mw[w]); }
    float a[4], l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a[w] = __builtin_amdgcn_exp2f(mw[w] - M); l = fmaf(xch[w * (66 * 64) + 65 * 64 + lane], a[w], l); }
    const float inv = 1.0f / xhalf_sum(l);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) o = fmaf(xch[w * (66 * 64) + (16 * db + r) * 64 + lane], a[w], o);
        t[r] = o * inv;
      }
      ox.set_block(db, t);
    }
  }
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    const int mb = 2 * wave + hb;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 8; ++s) mma3(acc, wa[16 * hb + s], wa[16 * hb + 8 + s], ox.h[s], ox.l[s]);
    float b[16], t[16];
    load_vec16(b, lvec + 6 * LAT, mb, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fmaf(acc[r], kH2Inv, b[r]) + xq[16 * hb + r];
    store_blk<LAT>(x1_out + toff, mb, t, lane);
  }
}

// ---------------------------------------------------------------------------------------------
static inline dim3 wgrid(int tiles, int B) { return dim3((tiles + kWaves - 1) / kWaves, B); }

hipError_t launch_ctx_prep_w(bool pe, const float* ctx, const float* wst, const float* vecs, float* out, int B, int T,
                             int ttiles, hipStream_t s) {
  if (pe) hipLaunchKernelGGL(k_ctx_prep_w<true>, wgrid(ttiles, B), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles);
  else hipLaunchKernelGGL(k_ctx_prep_w<false>, wgrid(ttiles, B), dim3(256), 0, s, ctx, wst, vecs, out, T, ttiles);
  return hipGetLastError();
}

hipError_t launch_fusion_attn_w(bool pe, const float* x, const float* ctx_img, const float* wst, const float* vecs,
                                float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s) {
  if (pe) hipLaunchKernelGGL(k_fusion_attn_w<true>, wgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  else hipLaunchKernelGGL(k_fusion_attn_w<false>, wgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst, vecs, x1, N, tiles, T, ttiles);
  return hipGetLastError();
}

hipError_t launch_ctx_prep_w_h2(bool pe, const float* ctx, const float* wst_h2, const float* vecs, float* out, int B, int T,
                                int ttiles, hipStream_t s, bool rowmajor) {
  if (rowmajor) {
    if (pe) hipLaunchKernelGGL((k_ctx_prep_w_h2<true, true>), wgrid(ttiles, B), dim3(256), 0, s, ctx, wst_h2, vecs, out, T, ttiles);
    else hipLaunchKernelGGL((k_ctx_prep_w_h2<false, true>), wgrid(ttiles, B), dim3(256), 0, s, ctx, wst_h2, vecs, out, T, ttiles);
  } else {
    if (pe) hipLaunchKernelGGL((k_ctx_prep_w_h2<true, false>), wgrid(ttiles, B), dim3(256), 0, s, ctx, wst_h2, vecs, out, T, ttiles);
    else hipLaunchKernelGGL((k_ctx_prep_w_h2<false, false>), wgrid(ttiles, B), dim3(256), 0, s, ctx, wst_h2, vecs, out, T, ttiles);
  }
  return hipGetLastError();
}

hipError_t launch_fusion_attn_w_h2(bool pe, const float* x, const float* ctx_img, const float* wst_h2, const float* vecs,
                                   float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s, bool tile_form) {
  if (tile_form && tiles * B <= 256) {             // a single round of one-tile workgroups
    if (pe) hipLaunchKernelGGL(k_fusion_attn_w_tile<true>, dim3(tiles, B), dim3(256), 0, s, x, ctx_img, wst_h2, vecs, x1, N, tiles, T, ttiles);
    else hipLaunchKernelGGL(k_fusion_attn_w_tile<false>, dim3(tiles, B), dim3(256), 0, s, x, ctx_img, wst_h2, vecs, x1, N, tiles, T, ttiles);
    return hipGetLastError();
  }
  if (pe) hipLaunchKernelGGL(k_fusion_attn_w_h2<true>, wgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst_h2, vecs, x1, N, tiles, T, ttiles);
  else hipLaunchKernelGGL(k_fusion_attn_w_h2<false>, wgrid(tiles, B), dim3(256), 0, s, x, ctx_img, wst_h2, vecs, x1, N, tiles, T, ttiles);
  return hipGetLastError();
}

// hidden splits of the wide feed-forward: one workgroup per CU, so as many as keep the grid within the 256 CUs
int plan_ff_split_w(int base_wgs) {
  int hs = 1;
  while (hs < 8 && base_wgs * hs * 2 <= 256) hs *= 2;
  // [r5] a second round that is less than half full (257 .. 384 workgroups, e.g. 40 000 voxels): two half-length workgroups per row block
  // fill both rounds, and the partials' reduction writes the caller's tensor itself (no unpacking pass): 0.457 -> 0.404 ms at 40 000
  if (base_wgs > 256 && base_wgs <= 384) hs = 2;
  return hs;
}

hipError_t launch_fusion_ff_w_h2(const float* x1, const float* wst_h2, const float* vecs, float* x2, int B, int tiles, hipStream_t s,
                                 float* part, int hs, float* out_rm, long o_sb, long o_sr, long o_sk, int n_rows, int* status) {
  const dim3 g = wgrid(tiles, B);
  if (part && hs > 1) {
    hipLaunchKernelGGL(k_fusion_ff_w_h2, dim3(g.x, g.y, hs), dim3(256), 0, s, x1, wst_h2, vecs, x2, tiles, part);
    hipLaunchKernelGGL(k_ff_reduce_w, dim3(g.x, g.y, 8), dim3(256), 0, s, part, x1, vecs, x2, tiles, hs, out_rm, o_sb, o_sr, o_sk, n_rows, status);
  } else {
    hipLaunchKernelGGL(k_fusion_ff_w_h2, g, dim3(256), 0, s, x1, wst_h2, vecs, x2, tiles, (float*)nullptr);
  }
  return hipGetLastError();
}

hipError_t launch_fusion_ff_w(const float* x1, const float* wst, const float* vecs, float* x2, int B, int tiles, hipStream_t s) {
  hipLaunchKernelGGL(k_fusion_ff_w, wgrid(tiles, B), dim3(256), 0, s, x1, wst, vecs, x2, tiles);
  return hipGetLastError();
}

}  // namespace gmf
