// Internal C++ launch entry points (one per kernel family).  The public C ABI is include/gmf_hip.h.
#pragma once
#include <hip/hip_runtime.h>

namespace gmf {

// Ragged batches (gmf_encoder_forward_ragged / gmf_pose_head_ragged): pair b owns rows [row0, row0 + n) of the caller's packed
// row-major tensors ([sum n, ...]); S = int(n * ratio) seeds and k = min(k, n - 1) neighbours in the pose head.  The table is
// built on the host from the caller's per-pair sizes and uploaded with the call.  Kernels take a nullable pointer: null =
// the uniform batch [B, N, ...].  Internally every pair keeps a slot of tiles(max n) tiles; a pair's tiles beyond its own are
// neither computed nor read.
// ord [r5]: the attention grid's pair SLOT -> pair map of a ragged batch.  The attention items are dealt to the 8 XCDs in contiguous runs of
// slots (attn_item), and an item costs n^2: with the pairs in the caller's order an XCD's share of the work is whatever its four pairs
// happen to be (32 pairs with N ~ U[4000, 5500]: +-9 % around the mean, and the launch ends with the slowest XCD).  build_pair_table
// deals the pairs to the XCD runs longest first, each to the run with the least work so far.  Results do not depend on it.
struct PairTab { int row0, n, S, k, ord, pad0, pad1, pad2; };

// The "pv_fp8" guard (gmf_set_tuning "pv_fp8" = 1, the default): which of a layer's pairs run the P V cross products of
// the spatial-consistency attention on the fp8 pipe is decided ON THE DEVICE, per pair, from a statistic the previous kernel
// left behind - the largest squared row norm of the layer's input f - against the layer's threshold (gmf_encoder_weights::
// pv_guard).  Every kernel that writes a V image or multiplies one evaluates the same comparison on the same two words, so the
// writer and the reader of a pair's V tiles always agree on their format.
constexpr int kPvStatStride = 32;    // words between two pairs' statistics: one 128-byte line each.  (Packed into one line, the 5 000 device-scope
                                    // atomics of k_front_h2<3> - every wave of every pair within microseconds - serialised at the memory side:
                                    // 30 -> 64 us for that kernel; a line per pair lets the pairs' updates proceed in parallel.)
struct PvGuard {
  const unsigned* stat = nullptr;   // [B][kPvStatStride] float bits of max row |f_l|^2 per pair (word 0 of its line), complete before the layer's first launch; null: unguarded
  const float* thr2 = nullptr;      // the layer's threshold: fp8 cross products while stat <= *thr2
  unsigned* stat_next = nullptr;    // [B] the NEXT layer's statistic: raised (atomic maximum) by whoever produces f_{l+1}; null: nobody asks
};

// compat matrix built once per batch by launch_compat_build (see k_compat_build): [B, tiles, tiles, 1024] floats
struct CompatCache {
  const float* dense;
  const float* tail_wst_h2;   // this layer's fc_message weights as split-fp16 images (epilogue of k_scattn_h2p)
  float* part_o;              // key-split workspace: [max_splits][B * tiles] P32 tile images (or nullptr)
  float* part_ml;             // ... [max_splits][B * tiles][32][2] row maximum, row sum
  int max_splits;
  // when non-null, the attention epilogue applies the NEXT layer's PointCN (4 split-fp16 weight stages + bias) to the block
  // output and stores f_{l+1} instead of feat (k_scattn_h2p / k_scattn_merge only)
  const float* next_wst_h2 = nullptr;
  const float* next_bias = nullptr;
  bool half = false;          // `dense` holds fp16 tiles (2 KiB each) and the attention multiplies one fp16 product: the
                              // throughput numerics mode (Tuning::precision = 1), large grids only (then fmt = 1)
  const PairTab* ptab = nullptr;   // ragged batch: per-pair rows (device), else null
  int min_tiles = 0;               // ragged batch: the smallest pair's 32-row tiles (the key-split plan is made for it)
  int fmt = 0;                // element format of `dense` (k_compat_build): 0 = fp32 (4 KiB per tile); 16-bit, 2 KiB per tile:
                              // 1 = fp16 c (with `half`), 2 = fixed point rint(65535 c)
  unsigned* v_scale = nullptr;   // non-null: the layer's V image carries e4m3 cross planes (store_block_v8) and these are their
                                 // scale words, [B * tiles][64]; k_linear_h2 writes both, k_scattn_h2p<3, *, true> reads them
  // [r4] non-null: no Q' image exists - every attention workgroup projects its own Q' in its prologue from the layer's f image
  // (qf_img), the four Q' weight stages (qw_wst) and bq' (qw_bias, 128 floats); parity kernels of the pipelined form only
  const float* qf_img = nullptr;
  const float* qw_wst = nullptr;
  const float* qw_bias = nullptr;
  PvGuard guard;              // [r5] per pair: e4m3 cross planes or V's low fp16 plane (with v_scale)
};

// Per-handle tuning knobs (gmf_set_tuning).  Every value selects between forms that compute the same result up to
// rounding; there is no process-global state and no environment variable behind any of them.
struct Tuning {
  int scattn_variant = 18;   // 18 = cached, software-pipelined split-fp16 attention; 9 = split-fp16 without the pipelining
                             // (and without the cache when compat_cache = 0); 0 = fp32 MFMA path for every stage
  bool use_cache = true;     // compat cache (built once per batch) for variants 9 / 18
  int key_splits = 0;        // attention key splits: 0 = automatic (small grids, and the tail of large ones), 1 = off, n = forced
  bool tail_split = false;   // large grids: split the last partial round of attention workgroups by keys.  Off by default: a CU
                             // left with one resident workgroup runs it almost twice as fast, so the "half-empty last round" costs
                             // little (32 x 5000: 1.10 ms split vs 1.07 ms whole, measured)
  int ff_split = 0;          // feed-forward hidden splits on small grids: 0 = automatic, 1 = off, 2 / 4 / 8 = forced
  bool front_split = true;   // small grids: one workgroup per output (Q' + f | K | V) of k_front_h2
  bool wide_attn_tile = true;     // 256-wide layer: cross-attention with one workgroup per tile on grids of up to 256 tiles
  bool small_merge_tile = true;   // small grids: the merge step with one workgroup per query tile, its waves splitting the feature blocks
  bool small_fattn_tile = true;   // [r5] small grids: the cross-attention role of the layer's first launch per query tile, context tiles dealt to its waves
  bool small_roles = true;   // small grids: three launches per layer with two kinds of workgroups each (else five or six)
  bool fused_linear = true;  // one kernel per layer for Q'/K/V + Fusion-2 (k_linear_h2) with the next PointCN in the attention epilogue
  int conv_patch = 1;        // stride-1 3x3 convolutions: 1 = LDS patch kernel (automatic form), 2 = never three workgroups per CU, 0 = gather kernel
  bool small_prologue = true;   // [r5] small grids: the prologue's two chains as roles of shared launches
  bool conv_small = true;    // [r5] grids of fewer than 128 workgroups of the 128-pixel convolution kernels (a few images): the K-split kernel
  int nms_binned = 1;        // 1 = grid-binned NMS candidates on large grids, 2 = always, 0 = all pairs
  bool topk_select = true;   // radix select of the S seeds (false = full bitonic sort)
  int mid_grid_roles = 512;  // two-launch form on grids below this many base workgroups (>= 256): the linear kernel runs as two
                             // workgroup roles per row block (Q'/K/V | Fusion-2); 0 = never
  bool q_in_attention = true;   // [r4] large grids: Q' is projected in the attention kernel's prologue, not written by k_linear_h2 (bit-identical)
  int pv_fp8 = 1;            // parity arithmetic of the default path: the two cross products of O += P V on the block-scaled fp8 matrix
                             // pipe (scattn_h2p_body<3, *, 4, true>).  1 = guarded per pair and layer on the device
                             // (PvGuard), 2 = unconditionally, 0 = all three products on the f16 pipe
  int compat_format = 0;     // element format of the compat cache on the cached, pipelined path: 0 = fp32 (default); 2 = 16-bit fixed
                             // point, rint(65535 c): half the attention's c stream and half the build, -4 % per step, absolute
                             // error <= 7.6e-6 on c.  Measured: inside the parity contract on 3DMatch-shape
                             // inputs, 4-8x the reference's own fp32 noise on KITTI-shape inputs - so it is opt-in, not the default
  int precision = 0;         // NOT rounding-equivalent: 0 = parity numerics (fp32-equivalent split-fp16 products, the default);
                             // 1 = throughput numerics: the spatial-consistency attention multiplies plain
                             // fp16 operands (one product, fp32 accumulation) and streams c as fp16 - outside the 1e-4 gate
};

hipError_t launch_compat_build(const float* pts8, float* c_dense, int B, int N, int tiles, float sigma_d, int fmt, hipStream_t s,
                               const PairTab* ptab = nullptr);
hipError_t launch_front(int mode, const float* in, const float* wst, const float* vecs, float* f, float* q, float* k,
                        float* v, int B, int N, int tiles, hipStream_t s);
hipError_t launch_scattn_fp32(const float* q, const float* k, const float* v, const float* pts8, const float* fus,
                              const float* wst, const float* vecs, float* out, int B, int N, int tiles, float sigma_d,
                              hipStream_t s);
hipError_t launch_scattn_h2(const Tuning& tune, const float* q, const float* k, const float* v, const float* pts8,
                            const float* fus, const float* wst, const float* vecs, float* out, int B, int N, int tiles,
                            float sigma_d, hipStream_t s, const CompatCache* cc);
hipError_t launch_scattn_dense(const float* q, const float* k, const float* v, const float* compat, const float* fus,
                               const float* wst, const float* vecs, float* out, int B, int N, int tiles, hipStream_t s);
hipError_t launch_ctx_prep(bool pe, const float* ctx, const float* wst, const float* vecs, float* out, int B, int T,
                           int ttiles, int sets, int wst_stride, int vec_stride, hipStream_t s);
hipError_t launch_fusion_attn(bool pe, const float* x, const float* ctx_img, const float* wst, const float* vecs,
                              float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s);
hipError_t launch_fusion_ff(const float* x1, const float* wst, const float* vecs, float* x2, int B, int tiles, hipStream_t s);
hipError_t launch_head(const float* feat_img, const float* wst, const float* vecs, float* logits, float* feat_n,
                       float* feat_rm, int B, int N, int tiles, hipStream_t s, int* status = nullptr, const PairTab* ptab = nullptr,
                       const unsigned* pv_stat = nullptr, const float* pv_thr2 = nullptr, int n_layers = 0);   // pv_*: the "pv_fp8" guard's
                       // statistics [n_layers][B] and thresholds - a tripped layer sets GMF_STATUS_PV_GUARDED (informational)
hipError_t launch_ctx_prep_w(bool pe, const float* ctx, const float* wst, const float* vecs, float* out, int B, int T,
                             int ttiles, hipStream_t s);
hipError_t launch_fusion_attn_w(bool pe, const float* x, const float* ctx_img, const float* wst, const float* vecs,
                                float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s);
hipError_t launch_fusion_ff_w(const float* x1, const float* wst, const float* vecs, float* x2, int B, int tiles, hipStream_t s);
hipError_t launch_ctx_prep_w_h2(bool pe, const float* ctx, const float* wst_h2, const float* vecs, float* out, int B, int T,
                                int ttiles, hipStream_t s, bool rowmajor = false);
hipError_t launch_fusion_attn_w_h2(bool pe, const float* x, const float* ctx_img, const float* wst_h2, const float* vecs,
                                   float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s, bool tile_form = true);
hipError_t launch_fusion_ff_w_h2(const float* x1, const float* wst_h2, const float* vecs, float* x2, int B, int tiles, hipStream_t s,
                                 float* part = nullptr, int hs = 1, float* out_rm = nullptr, long o_sb = 0, long o_sr = 0,
                                 long o_sk = 0, int n_rows = 0, int* status = nullptr);
int plan_ff_split_w(int base_wgs);
hipError_t launch_front_h2(const Tuning& tune, int mode, const float* in, const float* wst, const float* vecs, float* f,
                           float* q, float* k, float* v, int B, int N, int tiles, hipStream_t s, const PairTab* ptab = nullptr,
                           unsigned* v_scale = nullptr, PvGuard guard = {});   // v_scale: V with e4m3 cross planes (CompatCache::v_scale)
// mode 3: corr_pos -> layer0 -> PointCN -> f only.  launch_linear_h2: all linear stages of one layer from f (k_linear_h2)
hipError_t launch_linear_h2(const Tuning& tune, const float* f, const float* front_wst, const float* front_vec, const float* ctx_img,
                            const float* attn_wst, const float* attn_vec, const float* ff_wst, const float* ff_vec, float* q,
                            float* k, float* v, float* x2, int B, int N, int tiles, int T, int ttiles, hipStream_t s, bool one_product = false,
                            const PairTab* ptab = nullptr, unsigned* v_scale = nullptr, PvGuard guard = {});   // v_scale: CompatCache::v_scale
// small grids: three launches per layer (k_small_front_fattn | k_small_attn_ff | k_scattn_merge)
void plan_attn_split(const Tuning& tune, int W, int tiles, int max_splits, int* n_full, int* ksplits);
int plan_ff_split(const Tuning& tune, int base, int max_parts);
hipError_t launch_small_front_fattn(const float* f, const float* front_wst, const float* front_vec, const float* ctx_img,
                                    const float* attn_wst, const float* attn_vec, float* q, float* k, float* v, float* x1, int B,
                                    int N, int tiles, int T, int ttiles, hipStream_t s, unsigned* v_scale = nullptr, PvGuard guard = {},
                                    bool tile_role = true, const PairTab* ptab = nullptr);   // tile_role: the cross-attention role per query tile (Tuning::small_fattn_tile)
hipError_t launch_small_attn_ff_merge(const float* q, const float* k, const float* v, const float* x1, const float* ff_wst,
                                      const float* ff_vecs, float* ff_part, int ff_hs, const float* tail_vecs, float* out, int B,
                                      int N, int tiles, int ksplits, hipStream_t s, const CompatCache* cc, bool tile_merge = true);
hipError_t launch_ctx_prep_h2(bool pe, const float* ctx, const float* wst, const float* vecs, float* out, int B, int T,
                              int ttiles, int sets, int wst_stride, int vec_stride, hipStream_t s, bool rowmajor = false);
hipError_t launch_fusion_attn_h2(bool pe, const float* x, const float* ctx_img, const float* wst, const float* vecs,
                                 float* x1, int B, int N, int tiles, int T, int ttiles, hipStream_t s, bool rowmajor = false);
// [r5] small grids: the forward's prologue as three launches of two roles each (encoder_h2.hip, k_pro_*)
hipError_t launch_pro_ctx_pts(const float* p_tokens, const float* wst, const float* vecs, float* f1ctx, int B, int T, int ttiles,
                              const float* src, const float* tgt, float* pts8, int N, hipStream_t s, const PairTab* ptab,
                              unsigned* zero_words, int n_zero);
hipError_t launch_pro_fattn_compat(const float* q_tokens, const float* f1ctx, const float* wst, const float* vecs, float* x1t, int B,
                                   int T, int ttiles, const float* pts8, float* c_dense, int N, int tiles, float sigma_d,
                                   hipStream_t s, const PairTab* ptab);
hipError_t launch_pro_ff_front(const Tuning& tune, const float* x1t, const float* wst, const float* vecs, float* imgfeat, int B,
                               int ttiles, float* part, int max_parts, const float* corr_pos, const float* fwst, const float* fvecs,
                               float* f, float* q, float* k, float* v, int N, int tiles, hipStream_t s, const PairTab* ptab,
                               PvGuard guard, int* hs_out);
hipError_t launch_ff_reduce_h2(const float* part, const float* x1, const float* vecs, float* x2, int B, int tiles, int hs, hipStream_t s);
hipError_t launch_fusion_ff_h2(const Tuning& tune, const float* x1, const float* wst, const float* vecs, float* x2, int B,
                               int tiles, hipStream_t s, float* part = nullptr, int max_parts = 0);
int padded_desc_width(int d);
hipError_t launch_nn_match(const float* F0, const float* F1, float* f0_img, float* f1_img, float* norm2, unsigned long long* best, int* idx,
                           float* dist, int N0, int N1, int d, int mode, hipStream_t s);
hipError_t launch_seed_dist(const float* featn_img, const int* seeds, float* dist, int B, int N, int S, hipStream_t s, const PairTab* ptab = nullptr);
hipError_t launch_pack_p32(const float* src, float* dst, int B, int n_rows, int K, long sb, long sr, long sk, hipStream_t s, const PairTab* ptab = nullptr);
// row-major [B, n_rows, 128] -> split-fp16 plane image (the operand image of launch_seed_dist)
hipError_t launch_pack_rows_h2(const float* src, float* dst, int B, int n_rows, hipStream_t s, const PairTab* ptab = nullptr,
                               const float* copy_src = nullptr, float* copy_dst = nullptr, long n_copy = 0);
hipError_t launch_unpack_p32(const float* src, float* dst, int B, int n_rows, int K, long sb, long sr, long sk, hipStream_t s, int* status = nullptr);
hipError_t launch_pack_pts8(const float* src, const float* tgt, float* dst, int B, int N, hipStream_t s, const PairTab* ptab = nullptr,
                            unsigned* zero_words = nullptr, int n_zero = 0);   // zero_words: n_zero words cleared by the same launch (PvGuard statistics)

// validation step (row f-4, forward half): validation_kernels.hip
size_t similarity_image_floats(int B, int N);
int sm_fused_parts_per_pair(int B, int N);
int sm_parts_per_pair(int B, int N);
int classification_parts(int B, int N);
int transformation_slices(int B, int N);
// sigma_dev (everywhere below): non-null = the kernels read sigma from this device address instead (gmf_set_sigma_device)
hipError_t launch_similarity_matrix(const float* feat_n, float* img, float* M, int B, int N, int ldm, float sigma,
                                    hipStream_t s, const float* sigma_dev = nullptr);
hipError_t launch_sm_loss_fused(const float* feat_n, const float* gt, float* img, double* part, double* pair_loss, int B,
                                int N, float sigma, int balanced, float* out, hipStream_t s, const float* sigma_dev = nullptr);
hipError_t launch_sm_loss(const float* M, int ldm, const float* gt, double* part, double* pair_loss, int B, int N,
                          int balanced, float* out, hipStream_t s);
int sm_backward_parts(int B, int N);
hipError_t launch_sm_backward(const float* feat_n, const float* gt, float* img, float* timg, float* consts, double* dsig_part,
                              int B, int N, float sigma, int balanced, float* dF, float* dsigma, hipStream_t s,
                              const float* sigma_dev = nullptr);
hipError_t launch_classification_loss(const float* pred, const float* gt, const float* weight, double* part, int B, int N,
                                      int balanced, float* out, hipStream_t s);
hipError_t launch_transformation_loss(const float* trans, const float* gt_trans, const float* src, const float* tgt,
                                      const float* probs, double* part, int B, int N, float re_thre, float te_thre,
                                      float* out, hipStream_t s);

// training primitives (row f-4): train_kernels.hip
int gemm_ksplits(bool ta, bool tb, const float* A, const float* B, int M, int N, int K, long lda, long ldb, long sA, long sB,
                 int batch);
hipError_t launch_gemm_f32(bool ta, bool tb, const float* A, const float* B, float* C, const float* bias, const float* R, int M, int N,
                           int K, long lda, long ldb, long ldc, long sA, long sB, long sC, int batch, float alpha, float* part,
                           int ksplits, int relu, hipStream_t s);
hipError_t launch_lcpe(bool backward, const float* x, const float* w, const float* bias, float* y, int rows, int L, int C, hipStream_t s);
hipError_t launch_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows, int C,
                         hipStream_t s);
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dx_add,
                         float* dx, long rows, int C, hipStream_t s);
hipError_t launch_softmax(bool backward, const float* a, const float* b, const float* mul, float* out, long rows, int T, float scale,
                          hipStream_t s);
hipError_t launch_geglu(bool backward, const float* hdn, const float* dg, float* out, long rows, int H, hipStream_t s);
int colsum_chunks(long rows);
hipError_t launch_colsum(const float* x, const float* y, const float* mean, const float* rstd, const float* cmean, const float* crstd,
                         int center_x, int shift, int L, long rows, int C, float* part, float* out, hipStream_t s,
                         const float* relu_y = nullptr, bool dual = false, float* out2 = nullptr);
hipError_t launch_bn_finish(const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean, float* run_var, int C,
                            long rows, float eps, float momentum, hipStream_t s);
hipError_t launch_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                           long rows, int C, int relu, hipStream_t s);
hipError_t launch_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* sdy,
                         const float* sdyx, float* dx, long rows, int C, hipStream_t s, const float* relu_y = nullptr);
hipError_t launch_relu_bwd(const float* dy, const float* y, float* out, long total, hipStream_t s);
hipError_t launch_bce_bwd(const float* pred, const float* gt, const float* weight, float* dpred, float* pw_scratch, int balanced,
                          long total, hipStream_t s);
hipError_t launch_sm_dense_bwd(const float* M, long ldm, const float* gt, float* consts, float* dM, int B, int N, int balanced,
                               hipStream_t s);
hipError_t launch_normalize(bool backward, const float* a, const float* dy, float* nrm, float* out, long rows, int C, hipStream_t s);
hipError_t launch_compat_dense(const float* src, const float* tgt, float* out, int B, int N, float sigma_d, hipStream_t s);
hipError_t launch_sim_bwd_G(const float* S, const float* dM, float* G, float* rowdsig, int B, int N, float sigma, hipStream_t s,
                            const float* sigma_dev = nullptr);

// image encoder epilogue (row f-1): image_kernels.hip
hipError_t launch_stem_h2(const float* x, long sb, long sc, long sh, long sw, const float* wimg, const float* bias, float* y,
                          int B, int H, int W, hipStream_t s, bool small_grid = true);   // small_grid: Tuning::conv_small
hipError_t launch_bias_relu_nhwc(float* y, const float* bias, const float* residual, long n_pixels, int C, hipStream_t s);
hipError_t launch_conv_nhwc_h2(const Tuning& tune, const float* x, const float* wimg, const float* bias, const float* residual, float* y, int B,
                               int H, int W, int cin, int cout, int ks, int stride, int relu, hipStream_t s);

}  // namespace gmf
