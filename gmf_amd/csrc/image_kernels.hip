// Image encoder (ResNet-34 truncated after layer2, GMF_PointDSC/models/resnet.py:195-216).
// The convolutions stay with MIOpen (PyTorch-ROCm, NHWC fp32); what the framework leaves un-fused around them - the folded
// BatchNorm bias, the residual add and the ReLU, three element-wise passes per convolution - is one pass here:
//   y = max(y + bias[c] (+ residual), 0)   in place on an NHWC tensor (resnet.py:59-75, BasicBlock.forward).
#include <hip/hip_runtime.h>

#include "mfma_core.hpp"
#include "launchers.hpp"

namespace gmf {

// =========================================================================================
// k_conv_nhwc_h2: 3x3 (pad 1) or 1x1 convolution, stride 1 or 2, NHWC fp32 in and out, as an implicit GEMM on the f16 MFMA
// with split-fp16 operands (the arithmetic of the attention kernel: three products per multiply-add, fp32 accumulate,
// weights stored as 256 W).  M = output pixels (32 per wave, 128 per workgroup), N = COUT (all blocks of 32 per wave),
// K = taps x CIN in k-steps of 16 = (tap, 16 input channels).
//   A operand: lane (pixel i, half h) loads input channels 16 cb + 8 h .. + 7 of the tap's input pixel (two float4s, zeros
//              outside the image), splits them to fp16 hi/lo in registers; the next k-step's loads are in flight meanwhile.
//   B operand: the weight image streams through a ring of 16 KiB LDS stages by DMA; per k-step and output block one
//              (hi, lo) pair of 1 KiB units, per 64-channel half of the output: unit ((ks * 2 + blk) * 2 + plane) * 64 + lane,
//              lane (h, i) holding W[64 half + 32 blk + i][k = 16 ks + 8 h .. + 7], k = tap * CIN + cin (packing.conv_image).
//   epilogue:  y = acc / 256 + bias[cout] (+ residual) -> optional ReLU; a register row of a lane is one pixel, lanes run
//              over 32 consecutive output channels: 128-byte rows.
// =========================================================================================
// CBT: k order (channel block, tap) instead of (tap, channel block) - the order of the weight images of the stride-1 3x3
// shapes, which k_conv3x3_patch_h2 shares.
template <int CIN, int COUT, int KS, int STRIDE, bool CBT = false>
__global__ void __launch_bounds__(256, 2)
k_conv_nhwc_h2(const float* __restrict__ x, const float* __restrict__ wimg, const float* __restrict__ bias,
               const float* __restrict__ residual, float* __restrict__ y, int B, int H, int W, int Ho, int Wo, int relu) {
  // a workgroup: 128 output pixels x 64 output channels (blockIdx.y selects the 64-channel half: twice the workgroups for
  // COUT = 128, whose 15 x 20 maps give only 150 pixel groups for 64 images); a stage = 4 k-steps x 2 blocks x (hi, lo)
  constexpr int NBLK = 2, CB = CIN / 16, NK = KS * KS * CB, KPS = 4, NST = NK / KPS, PAD = KS / 2, NB = 3;
  static_assert(NK % KPS == 0 && COUT % 64 == 0, "k-steps come in stages of four, output channels in halves of 64");
  __shared__ __attribute__((aligned(16))) float lds[NB * kStageFloats];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int co0 = 64 * blockIdx.y;
  wimg += (size_t)blockIdx.y * NST * kStageFloats;
  const long P = (long)B * Ho * Wo;
  const long pix0 = ((long)blockIdx.x * 4 + wave) * 32;
  const long pix = pix0 + i;
  const bool pvalid = pix < P;
  int yo = 0, xo = 0;
  long bimg = 0;
  if (pvalid) { bimg = pix / ((long)Ho * Wo); const int rem = (int)(pix - bimg * Ho * Wo); yo = rem / Wo; xo = rem - yo * Wo; }
  const int yc = yo * STRIDE - PAD, xc = xo * STRIDE - PAD;     // top-left input pixel of the window

  auto issue_stage = [&](int st) {
    const float* g = wimg + (size_t)st * kStageFloats;
    float* dst = lds + (st % NB) * kStageFloats;
    dma_4k_s(g + wave * 1024, dst + wave * 1024, (unsigned)lane * 16u);   // (4 vector-memory operations, one setup: mfma_core.hpp, StageRing)
  };
  // every lane always loads (clamped address, value zeroed afterwards): the number of vector-memory operations in flight
  // must not depend on the data, the s_waitcnt counts below rely on it
  const float* xb = x + (size_t)(pvalid ? bimg : 0) * H * W * CIN + 8 * h;
  auto load_a = [&](int ks, float (&a)[8]) {
    const int tap = CBT ? ks % (KS * KS) : ks / CB, cb = CBT ? ks / (KS * KS) : ks - tap * CB;
    const int dy = tap / KS, dx = tap - dy * KS;
    const int yi = yc + dy, xi = xc + dx;
    const bool ok = pvalid && yi >= 0 && yi < H && xi >= 0 && xi < W;
    const int yl = min(max(yi, 0), H - 1), xl = min(max(xi, 0), W - 1);
    const float4* p = reinterpret_cast<const float4*>(xb + ((size_t)yl * W + xl) * CIN + 16 * cb);
    const float4 u = p[0], v = p[1];
    a[0] = ok ? u.x : 0.f; a[1] = ok ? u.y : 0.f; a[2] = ok ? u.z : 0.f; a[3] = ok ? u.w : 0.f;
    a[4] = ok ? v.x : 0.f; a[5] = ok ? v.y : 0.f; a[6] = ok ? v.z : 0.f; a[7] = ok ? v.w : 0.f;
  };

  f32x16 acc[NBLK];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) acc[blk] = zero16();
  // the activations of a whole stage (4 k-steps) are requested one stage ahead, AFTER that stage's DMA pieces were issued:
  // in the vmcnt queue they are then younger than the weights they will meet and older than the next stage's DMA
  float a_cur[KPS][8], a_nxt[KPS][8];
  issue_stage(0);
  if (NST > 1) issue_stage(1);
#pragma unroll
  for (int kk = 0; kk < KPS; ++kk) load_a(kk, a_cur[kk]);
  for (int st = 0; st < NST; ++st) {
    // stage st landed for this wave when at most [DMA(st+1): 4] + [A(st): 8, waited for by the compiler anyway] are out
    if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();                               // everyone's pieces of stage st landed; slot (st + 2) % 3 is free
    if (st + 2 < NST) issue_stage(st + 2);
    if (st + 1 < NST) {
#pragma unroll
      for (int kk = 0; kk < KPS; ++kk) load_a((st + 1) * KPS + kk, a_nxt[kk]);
    }
    const f16x8* lw = reinterpret_cast<const f16x8*>(lds + (st % NB) * kStageFloats) + lane;
#pragma unroll
    for (int kk = 0; kk < KPS; ++kk) {
      f16x8 ah, al;
      split8h(a_cur[kk], ah, al);
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk)
        mma3(acc[blk], ah, al, lw[((kk * NBLK + blk) * 2 + 0) * 64], lw[((kk * NBLK + blk) * 2 + 1) * 64]);
    }
#pragma unroll
    for (int kk = 0; kk < KPS; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) a_cur[kk][e] = a_nxt[kk][e];
  }
  // register r of lane (h, i): pixel pix0 + 8 (r >> 2) + 4 h + (r & 3), output channel co0 + 32 blk + i
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const float bv = bias[co0 + 32 * blk + i];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long p = pix0 + 8 * (r >> 2) + 4 * h + (r & 3);
      if (p < P) {
        const size_t o = (size_t)p * COUT + co0 + 32 * blk + i;
        float v = fmaf(acc[blk][r], 1.0f / 256.0f, bv);
        if (residual) v += residual[o];
        y[o] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// =========================================================================================
// k_conv_small_h2 [r5]: the same convolutions on SMALL grids - a few images, e.g. the two images of ONE scene pair, which is what the
// reference's evaluation loop feeds (evaluation/test_3DMatch.py:69; resnet.py:195-216).  With 128 output pixels x 64 channels per
// workgroup two 120 x 160 images give 19 (layer1) / 10 (layer2) workgroups, each a chain of 36 .. 72 dependent k-steps: a launch
// costs its chain (8 - 12 us), which is why such batches used MIOpen's kernels until round 4.  Here a workgroup owns 32 output
// pixels x 32 output channels and its NW waves SPLIT THE K RANGE (k-steps [w NK / NW, (w + 1) NK / NW) of the same tile; NW = 4 for 64
// input channels, 8 for 128: nine k-steps per wave either way): four times the workgroups per pixel and twice per channel (150 / 76
// for two images); the partial tiles meet in LDS and are added in wave order (deterministic).  No LDS staging of the weights: a
// wave requests ALL its k-steps' operands - two (hi, lo) weight fragments from the L2-resident image, two float4 of activations
// each - before its first product (one memory round trip per launch instead of one per stage: 0.37 -> 0.19 ms per two-image pass).
// Same weight images, same arithmetic per product as k_conv_nhwc_h2 (split-fp16 operands, three products, fp32 accumulate); the
// accumulation ORDER over k differs (four partial sums), so the two kernels agree to fp32 rounding, not bit for bit.
// grid (ceil(P / 32), COUT / 32), block 64 NW.
// =========================================================================================
template <int CIN, int COUT, int KS, int STRIDE, bool CBT, int NW>
__global__ void __launch_bounds__(64 * NW)
k_conv_small_h2(const float* __restrict__ x, const float* __restrict__ wimg, const float* __restrict__ bias,
                const float* __restrict__ residual, float* __restrict__ y, int B, int H, int W, int Ho, int Wo, int relu) {
  constexpr int CB = CIN / 16, NK = KS * KS * CB, NKW = NK / NW, PAD = KS / 2;
  constexpr int CHK = NKW, NCH = 1;               // every k-step of the wave in flight at once (9 x 16 + 9 x 8 registers)
  static_assert(NK % NW == 0 && COUT % 64 == 0 && 16 % NW == 0 && NKW <= 9, "the k range splits evenly over the waves");
  __shared__ float red[NW][16][64];
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = blockIdx.y >> 1, blk = blockIdx.y & 1, co0 = 32 * blockIdx.y;
  const long P = (long)B * Ho * Wo;
  const long pix0 = (long)blockIdx.x * 32;
  const long pix = pix0 + i;
  const bool pvalid = pix < P;
  int yo = 0, xo = 0;
  long bimg = 0;
  if (pvalid) { bimg = pix / ((long)Ho * Wo); const int rem = (int)(pix - bimg * Ho * Wo); yo = rem / Wo; xo = rem - yo * Wo; }
  const int yc = yo * STRIDE - PAD, xc = xo * STRIDE - PAD;
  const float* xb = x + (size_t)(pvalid ? bimg : 0) * H * W * CIN + 8 * h;
  auto load_a = [&](int ks, float (&a)[8]) {
    const int tap = CBT ? ks % (KS * KS) : ks / CB, cb = CBT ? ks / (KS * KS) : ks - tap * CB;
    const int dy = tap / KS, dx = tap - dy * KS;
    const int yi = yc + dy, xi = xc + dx;
    const bool ok = pvalid && yi >= 0 && yi < H && xi >= 0 && xi < W;
    const int yl = min(max(yi, 0), H - 1), xl = min(max(xi, 0), W - 1);
    const float4* p = reinterpret_cast<const float4*>(xb + ((size_t)yl * W + xl) * CIN + 16 * cb);
    const float4 u = p[0], v = p[1];
    a[0] = ok ? u.x : 0.f; a[1] = ok ? u.y : 0.f; a[2] = ok ? u.z : 0.f; a[3] = ok ? u.w : 0.f;
    a[4] = ok ? v.x : 0.f; a[5] = ok ? v.y : 0.f; a[6] = ok ? v.z : 0.f; a[7] = ok ? v.w : 0.f;
  };
  // weight image (packing.conv_image): 16-byte unit (((half NK + ks) 2 + blk) 2 + plane) 64 + lane
  const f16x8* wf = reinterpret_cast<const f16x8*>(wimg) + lane;
  auto load_b = [&](int ks, f16x8& bh, f16x8& bl) {
    const size_t u = (((size_t)half * NK + ks) * 2 + blk) * 2;
    bh = wf[u * 64];
    bl = wf[(u + 1) * 64];
  };
  const int k0 = wave * NKW;
  f32x16 acc = zero16();
  float a[2][CHK][8];
  f16x8 bh[2][CHK], bl[2][CHK];
#pragma unroll
  for (int kk = 0; kk < CHK; ++kk) { load_a(k0 + kk, a[0][kk]); load_b(k0 + kk, bh[0][kk], bl[0][kk]); }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) {
#pragma unroll
      for (int kk = 0; kk < CHK; ++kk) {
        load_a(k0 + (c + 1) * CHK + kk, a[(c + 1) & 1][kk]);
        load_b(k0 + (c + 1) * CHK + kk, bh[(c + 1) & 1][kk], bl[(c + 1) & 1][kk]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < CHK; ++kk) {
      f16x8 ah, al;
      split8h(a[c & 1][kk], ah, al);
      mma3(acc, ah, al, bh[c & 1][kk], bl[c & 1][kk]);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // wave w finishes registers (16 / NW) w ..: the NW partial sums in wave order, then bias (+ residual) and the ReLU.
  // register r of lane (h, i): pixel pix0 + 8 (r >> 2) + 4 h + (r & 3), output channel co0 + i
  const float bv = bias[co0 + i];
#pragma unroll
  for (int q = 0; q < 16 / NW; ++q) {
    const int r = (16 / NW) * wave + q;
    float sum = red[0][r][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[w][r][lane];
    const long p = pix0 + 8 * (r >> 2) + 4 * h + (r & 3);
    if (p < P) {
      const size_t o = (size_t)p * COUT + co0 + i;
      float v = fmaf(sum, 1.0f / 256.0f, bv);
      if (residual) v += residual[o];
      y[o] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

// =========================================================================================
// k_conv3x3_patch_h2: the stride-1 3x3 shapes (64 -> 64, 128 -> 128) with the activations staged through LDS.
//   For stride 1 the input pixels a workgroup's 128 consecutive (flattened NHWC) output pixels need are themselves one
//   contiguous flat range [p0 - W - 1, p0 + 128 + W]: per 16-channel block cb it is loaded once (coalesced 64-byte pieces),
//   split to fp16 hi / lo once and kept in LDS; the nine taps read it back with shifted addresses (two ds_read_b128 per lane
//   and k-step) instead of nine global gathers and nine splits of the same values.  Taps outside the image read a zero slot.
//   k order = (channel block, tap); weight image as packing.conv_image(W, order="cb_tap"), streamed as in k_conv_nhwc_h2.
//   Two patch buffers: block cb + 1 is fetched to registers before block cb is multiplied and written to LDS after it.
// =========================================================================================
// KPS: k-steps per weight stage (4: 16 KiB stages; 3: 12 KiB stages that align with the nine taps of a channel block).
// WMAX: largest image width the patch buffers hold.  DBUF: two patch buffers, or one with an extra barrier per channel block.
// <64, 64, 3, 40, false, 3> needs 49.5 KiB of LDS: three workgroups per CU, so that the 600 workgroups of the 64 -> 64 shape
// (64 images of 30 x 40) run in one round of the 768 slots instead of two rounds of 512.
template <int CIN, int COUT, int KPS, int WMAX, bool DBUF, int WGS>
__global__ void __launch_bounds__(256, WGS)
k_conv3x3_patch_h2(const float* __restrict__ x, const float* __restrict__ wimg, const float* __restrict__ bias,
                   const float* __restrict__ residual, float* __restrict__ y, int B, int H, int W, int relu) {
  constexpr int NBLK = 2, CB = CIN / 16, NK = 9 * CB, NST = NK / KPS, NB = 3;
  constexpr int kStage = KPS * NBLK * 512;         // floats per weight stage: KPS k-steps x 2 blocks x (hi, lo) KiB
  constexpr int kPatchMax = 128 + 2 * WMAX + 2;
  constexpr int kPatchUnits = kPatchMax * 2 + 2;   // 16-byte units per plane: pixel * 2 + half; the last two are the zero slot
  constexpr int NPB = DBUF ? 2 : 1;
  static_assert(NK % KPS == 0, "k-steps come in whole stages");
  __shared__ __attribute__((aligned(16))) float lds[NB * kStage];
  __shared__ __attribute__((aligned(16))) f16x8 patch[NPB][2][kPatchUnits];     // [buffer][plane][unit]
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int co0 = 64 * blockIdx.y;
  wimg += (size_t)blockIdx.y * NST * kStage;
  const long P = (long)B * H * W;
  const long p0 = (long)blockIdx.x * 128;          // first output pixel of the workgroup
  const long q0 = p0 - W - 1;                      // flat input pixel of patch index 0
  const int npatch = 128 + 2 * W + 2;
  const long pix0 = p0 + wave * 32;
  const long pix = pix0 + i;
  const bool pvalid = pix < P;
  int yo = 0, xo = 0;
  if (pvalid) { const int rem = (int)(pix % ((long)H * W)); yo = rem / W; xo = rem - yo * W; }
  // patch unit of tap (dy, dx) for this lane, or the zero slot
  int unit[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3, dx = t - 3 * dy;
    const int yi = yo + dy - 1, xi = xo + dx - 1;
    const bool ok = pvalid && yi >= 0 && yi < H && xi >= 0 && xi < W;
    unit[t] = ok ? 2 * (wave * 32 + i + dy * W + dx) + h : 2 * kPatchMax + h;
  }
  if (threadIdx.x < 4 * NPB) {                     // the zero slot of every buffer and plane
    f16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (_Float16)0;
    patch[threadIdx.x >> 2][(threadIdx.x >> 1) & 1][2 * kPatchMax + (threadIdx.x & 1)] = z;
  }

  auto issue_stage = [&](int st) {
    const float* g = wimg + (size_t)st * kStage;
    float* dst = lds + (st % NB) * kStage;
#pragma unroll
    for (int q = 0; q < KPS; ++q) dma_piece_1k_s(g + (wave + 4 * q) * 256, dst + (wave + 4 * q) * 256, (unsigned)lane * 16u);
  };
  // patch block cb: thread t moves the float4 quarter (t & 3) of patch pixels (t >> 2) + 64 j, j = 0 .. 3 (always four
  // loads, clamped address: the vmcnt counts below rely on it)
  constexpr int PJ = 4;
  static_assert(64 * PJ >= kPatchMax, "four pixels per thread cover the patch");
  auto load_patch = [&](int cb, float4 (&v)[PJ]) {
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int pp = (threadIdx.x >> 2) + 64 * j;
      long q = q0 + pp;
      q = q < 0 ? 0 : (q >= P ? P - 1 : q);
      v[j] = *reinterpret_cast<const float4*>(x + (size_t)q * CIN + 16 * cb + 4 * (threadIdx.x & 3));
    }
  };
  auto store_patch = [&](int buf, const float4 (&v)[PJ]) {
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int pp = (threadIdx.x >> 2) + 64 * j;
      if (pp < npatch) {
        const f32x2 a = {v[j].x, v[j].y}, b = {v[j].z, v[j].w};
        const f16x2 ah = __builtin_convertvector(a, f16x2), bh = __builtin_convertvector(b, f16x2);
        const f16x2 al = lo2h(ah, v[j].x, v[j].y), bl = lo2h(bh, v[j].z, v[j].w);
        // channel 4 * quarter .. + 3 of the pixel: half (quarter >> 1) of the unit, elements 4 (quarter & 1) ..
        const int u = 2 * pp + ((threadIdx.x & 3) >> 1), e0 = 4 * (threadIdx.x & 1);
        _Float16* dh = reinterpret_cast<_Float16*>(&patch[buf][0][u]) + e0;
        _Float16* dl = reinterpret_cast<_Float16*>(&patch[buf][1][u]) + e0;
        *reinterpret_cast<f16x2*>(dh) = ah; *reinterpret_cast<f16x2*>(dh + 2) = bh;
        *reinterpret_cast<f16x2*>(dl) = al; *reinterpret_cast<f16x2*>(dl + 2) = bl;
      }
    }
  };

  f32x16 acc[NBLK];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) acc[blk] = zero16();
  float4 pv[PJ];
  issue_stage(0);
  issue_stage(1);
  load_patch(0, pv);
  store_patch(0, pv);                              // (compiler waits for the four loads)
  if (CB > 1) load_patch(1, pv);
  // both loops are fully unrolled: the tap index selects a register of `unit`, and the (stage, k-step) position of every
  // (block, tap) pair is a compile-time constant
  const f16x8* lw = nullptr;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int pb = DBUF ? (cb & 1) : 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ks = cb * 9 + t, st = ks / KPS, kk = ks % KPS;
      if (kk == 0) {
        // everything but this wave's pieces of stage st + 1 has landed: stage st, and any patch loads (conservative)
        if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // also publishes the patch written before it
        if (st + 2 < NST) issue_stage(st + 2);
        lw = reinterpret_cast<const f16x8*>(lds + (st % NB) * kStage) + lane;
      } else if (t == 0) {
        __syncthreads();                           // patch block cb is in LDS
      }
      const f16x8 ah = patch[pb][0][unit[t]], al = patch[pb][1][unit[t]];
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk)
        mma3(acc[blk], ah, al, lw[((kk * NBLK + blk) * 2 + 0) * 64], lw[((kk * NBLK + blk) * 2 + 1) * 64]);
    }
    // block cb + 1 (in registers since the start of block cb) goes to LDS - to the other buffer, whose last readers
    // finished before the barrier that opened block cb, or, with one buffer, after a barrier that ends block cb - and
    // block cb + 2 is requested
    if (cb + 1 < CB) {
      if (!DBUF) __syncthreads();
      store_patch(DBUF ? ((cb + 1) & 1) : 0, pv);
      if (cb + 2 < CB) load_patch(cb + 2, pv);
    }
  }
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const float bv = bias[co0 + 32 * blk + i];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long p = pix0 + 8 * (r >> 2) + 4 * h + (r & 3);
      if (p < P) {
        const size_t o = (size_t)p * COUT + co0 + 32 * blk + i;
        float v = fmaf(acc[blk][r], 1.0f / 256.0f, bv);
        if (residual) v += residual[o];
        y[o] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// total4 = n_pixels * C / 4 float4s; C % 4 == 0, so a float4 never straddles two pixels
__global__ void __launch_bounds__(256)
k_bias_relu_nhwc(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ residual, long total4,
                 int c4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const float4 b = reinterpret_cast<const float4*>(bias)[idx % c4];
  float4 v = reinterpret_cast<float4*>(y)[idx];
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  if (residual) {
    const float4 r = reinterpret_cast<const float4*>(residual)[idx];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  reinterpret_cast<float4*>(y)[idx] = v;
}

// =========================================================================================
// k_stem_h2: the ResNet stem in one kernel (GMF_PointDSC/models/resnet.py:198-204: conv1 7x7 stride 2 pad 3, 3 -> 64
// channels, BatchNorm (folded into the weights / bias), ReLU, max-pool 3x3 stride 2 pad 1).
//   input  x [B, 3, H, W] with arbitrary element strides (NCHW as the reference hands it over, or a channels-last view)
//   output y [B, Hp, Wp, 64] NHWC fp32, Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1 (same for W)
// A workgroup produces an 8 x 8 tile of pooled pixels x 64 channels:
//   1. the 39 x 39 x 3 input patch behind the 17 x 17 convolution outputs the tile's pool windows touch is loaded to LDS,
//      pixel-major ([y][x][c], zeros outside the image), coalesced along x;
//   2. the convolution is an implicit GEMM on the f16 MFMA with split-fp16 operands (three products per multiply-add, fp32
//      accumulate; weights as an image of 256 W): M = 289 convolution pixels in 10 tiles of 32 (wave w owns tiles w, w + 4,
//      w + 8), N = 64 channels (2 blocks), K = 7 kernel rows x 24 (21 = 7 taps x 3 channels + 3 zero weights) = 168, padded
//      to 11 k-steps of 16.  A k-step's 8 values of a lane are 8 CONSECUTIVE floats of one patch row (the kernel row's 21
//      values are contiguous in the pixel-major patch), read with four ds_read_b64 and split to fp16 hi / lo in registers;
//      the weight fragments of a k-step are loaded from global memory (L2-resident: 44 KiB) once per wave and k-step and
//      serve its three pixel tiles;
//   3. the convolution tile goes to LDS (over the patch), and the pool + bias + ReLU pass (max commutes with the monotone
//      bias + ReLU) writes 64-channel rows.
// Weight image (packing.stem_image): 16-byte unit ((ks * 2 + blk) * 2 + plane) * 64 + lane, lane (h, j) holding
// W'[32 blk + j][k = 16 ks + 8 h .. + 7], k = 24 ky + 3 kx + c.
// =========================================================================================
constexpr int kStemCS = 68;                         // convolution tile row stride in floats (64 channels + 4)
constexpr int kStemKS = 11;

// kStemPT = pooled tile edge: 8 (a workgroup's waves own up to three 32-pixel convolution tiles each), or - [r5] grids of a few images,
// e.g. the two of one scene pair: 40 workgroups of the 8 x 8 form on 256 CUs - 4: a 9 x 9 convolution tile = three 32-pixel tiles, one per
// wave, four times the workgroups and a third of the chain per wave.  Same products in the same order per pixel: bit-identical.
template <int kStemPT>
__global__ void __launch_bounds__(256, 2)
k_stem_h2(const float* __restrict__ x, long sb, long sc, long sh, long sw, const float* __restrict__ wimg,
          const float* __restrict__ bias, float* __restrict__ y, int H, int W, int Hc, int Wc, int Hp, int Wp) {
  constexpr int kStemCT = 2 * kStemPT + 1;            // convolution tile edge (17 | 9)
  constexpr int kStemIT = 2 * kStemCT + 5;            // input patch edge (39 | 23)
  constexpr int kStemRow = kStemIT * 3 + 3;           // patch row stride in floats (120 | 72: even, so 8-byte reads stay aligned)
  constexpr int kStemRows = kStemIT + 1;              // + one zero row: the padded k-steps (ky = 7) read below the patch
  __shared__ __attribute__((aligned(16))) float lds[kStemCT * kStemCT * kStemCS];      // 19 652 floats >= the 4 800 of the patch (PT = 8)
  static_assert(kStemRows * kStemRow <= kStemCT * kStemCT * kStemCS, "the patch fits under the convolution tile");
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int img = blockIdx.z;
  const int py0 = blockIdx.y * kStemPT, px0 = blockIdx.x * kStemPT;     // first pooled pixel of the tile
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;                       // first convolution pixel (may be -1)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;                       // first input pixel
  // ---- 1. input patch -> LDS --------------------------------------------------------------------------------------
  const float* xi = x + (size_t)img * sb;
  for (int e = threadIdx.x; e < kStemRows * kStemRow; e += 256) {
    const int r = e / kStemRow, q = e - r * kStemRow;
    const int px = q / 3, c = q - 3 * px;
    const int yy = iy0 + r, xx = ix0 + px;
    float v = 0.f;
    if (r < kStemIT && px < kStemIT && yy >= 0 && yy < H && xx >= 0 && xx < W) v = xi[(size_t)c * sc + (size_t)yy * sh + (size_t)xx * sw];
    lds[e] = v;
  }
  __syncthreads();
  // ---- 2. implicit GEMM ---------------------------------------------------------------------------------------------
  constexpr int NPIX = kStemCT * kStemCT;          // 289 | 81
  constexpr int NT = (NPIX + 31) / 32, TS = (NT + 3) / 4;      // 32-pixel tiles (10 | 3), tile slots per wave (3 | 1)
  int aoff[TS];                                    // patch offset (floats) of the window origin of this lane's pixel, per tile
#pragma unroll
  for (int t = 0; t < TS; ++t) {
    const int p = min(32 * (wave + 4 * t) + i, NPIX - 1);
    const int cy = p / kStemCT, cx = p - cy * kStemCT;
    aoff[t] = (2 * cy) * kStemRow + (2 * cx) * 3;
  }
  const int n_tiles = (NT - wave + 3) / 4;         // tiles wave, wave + 4, ... below NT (10 tiles: waves 0, 1 own three; 3 tiles: waves 0 - 2 one)
  f32x16 acc[TS][2];
#pragma unroll
  for (int t = 0; t < TS; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }
  const f16x8* wp = reinterpret_cast<const f16x8*>(wimg) + lane;
  f16x8 wcur[2][2], wnxt[2][2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) { wcur[blk][0] = wp[(blk * 2 + 0) * 64]; wcur[blk][1] = wp[(blk * 2 + 1) * 64]; }
#pragma unroll
  for (int ks = 0; ks < kStemKS; ++ks) {
    if (ks + 1 < kStemKS) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        wnxt[blk][0] = wp[(((ks + 1) * 2 + blk) * 2 + 0) * 64];
        wnxt[blk][1] = wp[(((ks + 1) * 2 + blk) * 2 + 1) * 64];
      }
    }
    const int k0 = 16 * ks + 8 * h, ky = k0 / 24, ko = k0 - 24 * ky;       // per K-half: one kernel row, offset 0 / 8 / 16
    const int koff = ky * kStemRow + ko;
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      if (t < n_tiles) {
        const float2* ap = reinterpret_cast<const float2*>(lds + aoff[t] + koff);
        float a[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float2 v = ap[j]; a[2 * j] = v.x; a[2 * j + 1] = v.y; }
        f16x8 ah, al;
        split8h(a, ah, al);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) mma3(acc[t][blk], ah, al, wcur[blk][0], wcur[blk][1]);
      }
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) { wcur[blk][0] = wnxt[blk][0]; wcur[blk][1] = wnxt[blk][1]; }
  }
  __syncthreads();                                 // every wave is done reading the patch
  // ---- 3. convolution tile -> LDS: D = mfma(A = pixels, B = weights) has the CHANNEL on the lane (32 blk + i) and the
  //         pixels 8 (r >> 2) + 4 h + (r & 3) of the tile in the registers
#pragma unroll
  for (int t = 0; t < TS; ++t) {
    if (t < n_tiles) {
      const int pbase = 32 * (wave + 4 * t);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = pbase + 8 * (r >> 2) + 4 * h + (r & 3);
          if (p < NPIX) lds[p * kStemCS + 32 * blk + i] = acc[t][blk][r] * (1.0f / 256.0f);
        }
    }
  }
  __syncthreads();
  // ---- pool 3x3 stride 2 pad 1 + bias + ReLU ------------------------------------------------------------------------
  for (int e = threadIdx.x; e < kStemPT * kStemPT * 16; e += 256) {
    const int c4 = e & 15, pp = e >> 4;
    const int ty = pp / kStemPT, tx = pp - ty * kStemPT;
    const int py = py0 + ty, px = px0 + tx;
    if (py >= Hp || px >= Wp) continue;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ly = 2 * ty + dy, lx = 2 * tx + dx;          // tile-local convolution pixel
        const int cy = cy0 + ly, cx = cx0 + lx;
        if (cy < 0 || cy >= Hc || cx < 0 || cx >= Wc) continue;
        const float4 v = *reinterpret_cast<const float4*>(lds + (ly * kStemCT + lx) * kStemCS + 4 * c4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    const float4 b = reinterpret_cast<const float4*>(bias)[c4];
    float4 o = make_float4(fmaxf(m.x + b.x, 0.f), fmaxf(m.y + b.y, 0.f), fmaxf(m.z + b.z, 0.f), fmaxf(m.w + b.w, 0.f));
    reinterpret_cast<float4*>(y + (((size_t)img * Hp + py) * Wp + px) * 64)[c4] = o;
  }
}

hipError_t launch_stem_h2(const float* x, long sb, long sc, long sh, long sw, const float* wimg, const float* bias, float* y,
                          int B, int H, int W, hipStream_t s, bool small_grid) {
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  const dim3 grid((Wp + 7) / 8, (Hp + 7) / 8, B);
  if ((long)grid.x * grid.y * grid.z < 128 && small_grid) {
    const dim3 gs((Wp + 3) / 4, (Hp + 3) / 4, B);
    hipLaunchKernelGGL(k_stem_h2<4>, gs, dim3(256), 0, s, x, sb, sc, sh, sw, wimg, bias, y, H, W, Hc, Wc, Hp, Wp);
  } else {
    hipLaunchKernelGGL(k_stem_h2<8>, grid, dim3(256), 0, s, x, sb, sc, sh, sw, wimg, bias, y, H, W, Hc, Wc, Hp, Wp);
  }
  return hipGetLastError();
}

hipError_t launch_bias_relu_nhwc(float* y, const float* bias, const float* residual, long n_pixels, int C, hipStream_t s) {
  const long total4 = n_pixels * (C / 4);
  hipLaunchKernelGGL(k_bias_relu_nhwc, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, y, bias, residual, total4, C / 4);
  return hipGetLastError();
}

// tune.conv_patch: stride-1 3x3 shapes on the LDS-patch kernel when W <= 48 (else the gather kernel, same image): 1 = automatic, 2 = never the three-workgroup form, 0 = gather
hipError_t launch_conv_nhwc_h2(const Tuning& tune, const float* x, const float* wimg, const float* bias, const float* residual, float* y, int B,
                               int H, int W, int cin, int cout, int ks, int stride, int relu, hipStream_t s) {
  const int pad = ks / 2;
  const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
  const long P = (long)B * Ho * Wo;
  const dim3 grid((unsigned)((P + 127) / 128), cout / 64);
  // [r5] small grids (a few images): the K-split kernel, 32 pixels x 32 channels per workgroup (k_conv_small_h2); the switch is where
  // the 128-pixel kernel would leave more than half the chip without a workgroup
  if ((long)grid.x * grid.y < 128 && tune.conv_small) {
    const dim3 gs((unsigned)((P + 31) / 32), cout / 32);
#define GMF_CONV_SMALL(CI, CO, K, S, CBT_, NW_) hipLaunchKernelGGL((k_conv_small_h2<CI, CO, K, S, CBT_, NW_>), gs, dim3(64 * NW_), 0, s, x, wimg, bias, residual, y, B, H, W, Ho, Wo, relu)
    if (cin == 64 && cout == 64 && ks == 3 && stride == 1) GMF_CONV_SMALL(64, 64, 3, 1, true, 4);
    else if (cin == 64 && cout == 128 && ks == 3 && stride == 2) GMF_CONV_SMALL(64, 128, 3, 2, false, 4);
    else if (cin == 128 && cout == 128 && ks == 3 && stride == 1) GMF_CONV_SMALL(128, 128, 3, 1, true, 8);
    else if (cin == 64 && cout == 128 && ks == 1 && stride == 2) GMF_CONV_SMALL(64, 128, 1, 2, false, 4);
    else return hipErrorInvalidValue;
#undef GMF_CONV_SMALL
    return hipGetLastError();
  }
#define GMF_CONV(CI, CO, K, S) hipLaunchKernelGGL((k_conv_nhwc_h2<CI, CO, K, S>), grid, dim3(256), 0, s, x, wimg, bias, residual, y, B, H, W, Ho, Wo, relu)
  // stride-1 3x3 shapes: activations staged through LDS (weight image in (channel block, tap) order); W <= 48
  if (tune.conv_patch && ks == 3 && stride == 1 && W <= 48 && cin == cout && (cin == 64 || cin == 128)) {
    // grids between one round of 512 and one of 768 workgroup slots: the 49.5 KiB form (three workgroups per CU)
    const long nwg = (long)grid.x * grid.y;
    if (cin == 64 && W <= 40 && tune.conv_patch == 1 && nwg > 512 && nwg <= 768)
      hipLaunchKernelGGL((k_conv3x3_patch_h2<64, 64, 3, 40, false, 3>), grid, dim3(256), 0, s, x, wimg, bias, residual, y, B, H, W, relu);
    else if (cin == 64) hipLaunchKernelGGL((k_conv3x3_patch_h2<64, 64, 4, 48, true, 2>), grid, dim3(256), 0, s, x, wimg, bias, residual, y, B, H, W, relu);
    else hipLaunchKernelGGL((k_conv3x3_patch_h2<128, 128, 4, 48, true, 2>), grid, dim3(256), 0, s, x, wimg, bias, residual, y, B, H, W, relu);
    return hipGetLastError();
  }
#define GMF_CONV_CBT(CI, CO) hipLaunchKernelGGL((k_conv_nhwc_h2<CI, CO, 3, 1, true>), grid, dim3(256), 0, s, x, wimg, bias, residual, y, B, H, W, Ho, Wo, relu)
  if (cin == 64 && cout == 64 && ks == 3 && stride == 1) GMF_CONV_CBT(64, 64);
  else if (cin == 64 && cout == 128 && ks == 3 && stride == 2) GMF_CONV(64, 128, 3, 2);
  else if (cin == 128 && cout == 128 && ks == 3 && stride == 1) GMF_CONV_CBT(128, 128);
  else if (cin == 64 && cout == 128 && ks == 1 && stride == 2) GMF_CONV(64, 128, 1, 2);
  else return hipErrorInvalidValue;
#undef GMF_CONV
#undef GMF_CONV_CBT
  return hipGetLastError();
}

}  // namespace gmf
