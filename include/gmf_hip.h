/* gmf_hip.h - C ABI of libgmf_hip.so: the MI355X (gfx950) implementation of GMF's
 * multimodal-fusion hot path.
 *
 * The reference (XiaoshuiHuang/GMF) has NO native plugin ABI: the path sits behind Python
 * nn.Module.forward calls.  This header is therefore the boundary a binding
 * would target; each entry point names the reference interface it replaces (file:line relative to
 * the reference tree).  The Python drop-in modules in gmf_amd/ bind it with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - plain C types only; every data pointer is a DEVICE pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); the library never frees caller memory;
 *   - every call is asynchronous on the caller's stream (gmf_stream_t = hipStream_t passed as
 *     void*; NULL = the default stream) and performs no host synchronisation;
 *   - return value: 0 = GMF_OK, negative = error (see enum); gmf_last_error_string(h) has details;
 *     no exception or abort ever crosses this boundary;
 *   - a handle is bound to one device; distinct handles are independent.  Calls on ONE handle are serialised by an
 *     internal lock and every call reuses the handle's workspace, so a handle serves one logical stream of work: a call on a
 *     different stream than the previous call first waits (on the device) for the previous stream's work; the caller's
 *     current device is restored on return;
 *   - all tensors are fp32.  "P32 image" / "T image" are the tiled layouts described in
 *     gmf_amd/csrc/mfma_core.hpp; gmf_pack_rows_p32 / gmf_unpack_rows_p32 convert from and to
 *     arbitrary strided [B, rows, K] views (row-major [B,N,C] or channel-major [B,C,N]).
 */
#ifndef GMF_HIP_H_
#define GMF_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmf_handle gmf_handle;
typedef void* gmf_stream_t;

enum {
  GMF_OK = 0,
  GMF_ERR_BAD_ARG = -1,
  GMF_ERR_UNSUPPORTED_SHAPE = -2,
  GMF_ERR_HIP = -3,
  GMF_ERR_NO_DEVICE = -4,
  GMF_ERR_OOM = -5,
  GMF_ERR_WORKSPACE = -6   /* the caller-provided workspace (gmf_set_workspace) is too small for this call */
};

#define GMF_ABI_VERSION 5   /* 5: gmf_encoder_weights gained `pv_guard` (a caller that fills the struct itself must be rebuilt; one that uses
                             * gmf_encoder_pack_weights keeps working); "pv_fp8" takes 0 / 1 / 2.  4: + gmf_get_tuning; the pose head /
                             * pick_seeds take any N */

/* ---- lifetime ------------------------------------------------------------------------------ */
int gmf_abi_version(void);
/* Creates a handle on HIP device `device` (fails with GMF_ERR_NO_DEVICE if there is none). */
int gmf_create(int device, gmf_handle** out);
void gmf_destroy(gmf_handle* h);
const char* gmf_last_error_string(gmf_handle* h);
/* Bytes of device workspace currently held by the handle (library-owned or caller-provided). */
long long gmf_workspace_bytes(gmf_handle* h);
/* Workspace ownership.  By default the library owns ONE device block per handle (hipMalloc, grown with a device
 * synchronisation when a larger shape arrives - never grow it inside a stream capture).  gmf_set_workspace hands the
 * library a caller-owned device block instead (e.g. a torch tensor: the memory then belongs to the caller's allocator, and
 * the library frees its own block); a call that needs more than `bytes` returns GMF_ERR_WORKSPACE without launching anything
 * and gmf_workspace_wanted() returns the size that call asked for (with headroom): allocate, set, call again.
 * gmf_set_workspace(h, NULL, 0) returns to the library-owned block.  The block must stay valid until the work of the last
 * call that used it has completed on its stream.  (The reference has no counterpart: its intermediates are torch tensors.) */
int gmf_set_workspace(gmf_handle* h, void* device_ptr, long long bytes);
long long gmf_workspace_wanted(gmf_handle* h);

/* Sticky status word of the handle, kept in host-mapped memory: kernels OR bits into it, the host reads it WITHOUT any
 * device synchronisation (after the caller's own stream / device synchronisation it reflects every finished call).
 *   GMF_STATUS_NONFINITE: an inlier logit, a feature norm (gmf_encoder_forward / gmf_classifier_forward) or an output
 *   element of gmf_fusion_layer_forward was NaN or infinite.  The split-fp16 MFMA operands hold |x| < 65504 (weights are
 *   checked when they are packed; activations cannot be): an activation beyond that range turns into inf / NaN and ends up
 *   here instead of passing silently.  The reference (fp32 throughout) has no such limit.
 * *flags receives the word; clear != 0 resets it. */
#define GMF_STATUS_NONFINITE 1
/* [ABI 5] informational, not an error: under "pv_fp8" = 1 the device-side guard sent at least one (pair, layer) of a finished
 * gmf_encoder_forward(_ragged) to the three-product form of P V (the network's attention scores can exceed the bound of
 * gmf_encoder_weights::pv_guard there).  The results are as valid as any; the bit says why a step took ~3 % longer. */
#define GMF_STATUS_PV_GUARDED 2
int gmf_status_read(gmf_handle* h, int* flags, int clear);

/* [ABI 5] PointDSC's learnable scalar `sigma` (PointDSC.py:164) from DEVICE memory.  While `sigma_dev` is non-NULL every entry point
 * that takes sigma by value - gmf_similarity_matrix, gmf_similarity_backward, gmf_spectral_matching_loss_fused / _backward and, through
 * gmf_pose_params::sigma, gmf_pose_head(_ragged) / gmf_pose_head_backward - ignores that value and its kernels read the float at this
 * address when they run (the same correctly rounded 1 / sigma^2 as the host forms: identical bits).  What it is for: a training step
 * whose optimizer updates sigma on the device can be captured in a HIP graph and replayed without the one host read per step the
 * by-value form needs (libs/trainer.py:131-166; gmf_amd.train.GraphedTrainingStep).  NULL (default) restores the by-value form.  The
 * address must stay valid while set. */
int gmf_set_sigma_device(gmf_handle* h, const float* sigma_dev);

/* Per-handle tuning knobs (state lives in the handle; no process globals, no environment variables).  Every setting
 * except "precision" computes the same result up to rounding - there is no timing-only or wrong-result mode in the
 * library; unknown names and out-of-range values are rejected with GMF_ERR_BAD_ARG.
 *   "precision"         : 0 = parity numerics (default): every contraction is fp32-equivalent (split-fp16 operands, three
 *                         partial products, fp32 accumulation), logits and poses within 1e-4 of the reference.
 *                         1 = throughput numerics (the reduced-precision variant of SURVEY.md section 7 step 8): on large
 *                         grids (>= 256 attention workgroups) the spatial-consistency attention multiplies plain fp16
 *                         operands (ONE product, fp32 accumulation) and streams the compat matrix as fp16; softmax
 *                         statistics, LayerNorm, GELU, the linear stages and the pose head are unchanged.  NOT within the
 *                         1e-4 gate: measured against the parity mode at 32 x 5000: max |d logit| 1.6e-4, identical inlier
 *                         labels, 1.46x the throughput (1.50x at 16 x 10000).
 *                         2 = level 1, and the layer's linear stages (Q'/K/V projections, Fusion-2 cross-attention and
 *                         feed-forward; grids of >= 512 base workgroups) multiply only the high fp16 planes of weights and
 *                         activations too: max |d logit| 4e-2 (mean 2e-3), 99.9 % identical logit signs, 98-100 % identical
 *                         final labels, 1.6x the throughput.
 *   "scattn_variant"    : 18 = split-fp16 MFMA attention (2 planes, 3 products), compat matrix streamed from the per-batch
 *                         cache, tile loop software-pipelined inside each wave (default); 9 = the same arithmetic without
 *                         the pipelining; 0 = every encoder stage on the fp32 MFMA with fp32 images.
 *   "compat_cache"      : 1 = build the compat matrix once per batch (default), 0 = recompute c_ij in the attention kernel
 *                         (what the library does by itself when the cache would exceed 96 GB); 0 implies the
 *                         non-pipelined kernel.
 *   "compat_format"     : element format of the compat cache on the pipelined path: 0 = fp32 (default); 2 = 16-bit fixed point
 *                         rint(65535 c) - half the cache, half its stream, -4 % per step, |dc| <= 7.6e-6.  Opt-in: measured inside
 *                         the 1e-4 gate on 3DMatch-shape inputs, but 4-8x the reference's own fp32 noise on KITTI-shape inputs
 *                         (profiles/r03_compat_formats.txt).
 *   "pv_fp8"            : parity arithmetic of the default path (scattn_variant 18, fused_linear 1): the two CROSS products of the
 *                         attention's O += P V (P_hi V_lo + P_lo V_hi) on the block-scaled fp8 matrix pipe - one
 *                         v_mfma_scale_f32_32x32x64_f8f6f4 per feature block and key tile, e4m3 operands, one scale per (feature,
 *                         tile), the softmax row sum taken over exactly the probabilities the pipe multiplies; P_hi V_hi and all of
 *                         Q'K^T keep the three-product split-fp16 form.  1 (default) = GUARDED [ABI 5]: per pair and layer, on the
 *                         device, from the largest row norm of the layer's input features against gmf_encoder_weights::pv_guard
 *                         (a bound on the attention scores the layer can form): layers whose softmax can collapse onto single keys
 *                         - where the e4m3 rounding of one key's V reaches the output whole - take the three-product form, every
 *                         other layer the fp8 form.  No host synchronisation, no dependence on the other pairs of the batch.
 *                         2 = the fp8 form unconditionally, 0 = all three products of P V on the f16 pipe (the strict form).
 *                         Logits of the two forms differ by <= 3e-5 (mean 2e-6) at 32 x 5000 (profiles/r03_pv_fp8.txt).
 *   "attn_tail_split"   : 1 = large grids: the last partial round of attention workgroups is split by keys, 0 = whole (default).
 *   "small_grid_roles"  : 1 = small grids run three launches per layer with mixed workgroup roles (default), 0 = one per stage.
 *   "attn_key_splits"   : 0 = automatic (small grids only), 1 = off, 2..8 = forced number of key splits.
 *   "ff_hidden_splits"  : 0 = automatic (small grids only), 1 = off, 2 / 4 / 8 = forced.
 *   "front_output_split": 1 = small grids use one workgroup per output of the front kernel (default), 0 = never.
 *   "fused_linear"      : 1 = one kernel per layer for Q'/K/V + Fusion-2 (default), 0 = the three-kernel sequence.
 *   "wide_attn_tile"    : 1 = the cross-attention of the (256, 128) FusionLayer runs one workgroup per 32-row tile on grids of up
 *                         to 256 tiles, its waves splitting the feature blocks and the context tiles (default), 0 = one
 *                         workgroup per four tiles.
 *   "small_merge_tile"  : 1 = the merge step of the small-grid layer runs one workgroup per query tile whose four waves split
 *                         the feature blocks (default; bit-identical results), 0 = one workgroup per four tiles.
 *   "mid_grid_roles"    : W (default 512): on grids of 256 .. W - 1 base workgroups that kernel runs as two workgroup roles per
 *                         row block (Q'/K/V | Fusion-2; bit-identical results); 0 = never.
 *   "conv_lds_patch"    : 1 = stride-1 3x3 convolutions stage activations through LDS (default), 2 = same without the
 *                         three-workgroup form, 0 = gather form.
 *   "small_fattn_tile"  : [ABI 5] 1 = on small grids (B = 1, the reference's evaluation mode) the cross-attention role of a layer's first
 *                         launch runs one workgroup per query tile whose four waves deal the context tiles among themselves (default;
 *                         fusion_layer.py:84-94), 0 = one workgroup per four query tiles, each wave walking all context tiles.
 *                         Same products, another order of the softmax's partial sums.
 *   "conv_small_grid"   : [ABI 5] 1 = grids of fewer than 128 workgroups of those kernels - a few images, e.g. the two of one scene pair - run
 *                         the K-split kernel: 32 pixels x 32 channels per workgroup, its four waves a quarter of the k range each
 *                         (default), and the stem kernel 4 x 4 instead of 8 x 8 pooled pixels per workgroup (bit-identical);
 *                         0 = the 128-pixel kernels at every size.  Same products, another accumulation order.
 *   "small_prologue_roles": [ABI 5] 1 = on small grids the forward's prologue - two independent chains of few-workgroup kernels, image
 *                         side (Fusion-1 context / cross-attention / feed-forward, fusion_layer.py:172-201) and point side (key points,
 *                         compat cache PointDSC.py:216-221, layer 0 + first PointCN) - runs as three launches that carry one link of
 *                         each chain as two workgroup roles (default; the kernels' own bodies: bit-identical results), 0 = six kernels.
 *   "nms_binned"        : 1 = grid-binned NMS candidates on large grids (default), 2 = always, 0 = all pairs.
 *   "topk_select"       : 1 = radix select of the S seeds (default), 0 = full bitonic sort.
 *   "q_in_attention"    : [ABI 4] 1 = on large grids every attention workgroup projects its own Q' in its prologue (default; PointDSC.py:56),
 *                         0 = the linear kernel writes a Q' image.  Bit-identical results. */
int gmf_set_tuning(gmf_handle* h, const char* name, int value);
/* [ABI 4] The current value of a knob, so that a caller that changes one for a single call can put back what it found
 * (the Python PointDSC module does this for its module-local numerics mode).  Both calls take the handle's lock. */
int gmf_get_tuning(gmf_handle* h, const char* name, int* value);

/* In-situ timing of the dominant kernel (the spatial-consistency attention): while enabled, every
 * k_scattn launch made through gmf_encoder_forward / gmf_nonlocal_block_forward is bracketed by HIP events
 * recorded on the caller's stream.  gmf_profile_read synchronises those events (host sync - call it outside
 * timed regions), returns their summed duration and count, and resets the list.  Used by bench.py. */
int gmf_profile_enable(gmf_handle* h, int on);
int gmf_profile_read(gmf_handle* h, double* scattn_ms_total, int* scattn_launches);

/* ---- layout conversion ----------------------------------------------------------------------
 * src element (b, r, k) lives at src[b*sb + r*sr + k*sk]; dst is the P32 image
 * [B, ceil(n_rows/32), 32*K] (rows >= n_rows are zero).  K must be a multiple of 8.
 * Replaces the .permute()/.view() glue around the reference modules (PointDSC.py:70,130-135,223). */
int gmf_pack_rows_p32(gmf_handle* h, const float* src, long long sb, long long sr, long long sk, int B, int n_rows,
                      int K, float* dst, gmf_stream_t stream);
int gmf_unpack_rows_p32(gmf_handle* h, const float* src_img, int B, int n_rows, int K, float* dst, long long sb,
                        long long sr, long long sk, gmf_stream_t stream);
/* src_keypts, tgt_keypts [B,N,3] -> pts8 [B, Npad, 8] = (sx,sy,sz,0,tx,ty,tz,0). */
int gmf_pack_pts8(gmf_handle* h, const float* src, const float* tgt, int B, int N, float* dst, gmf_stream_t stream);

/* ---- encoder stages (operate on images; weights are packed blobs, layouts in gmf_amd/packing.py) */
/* layer0 (if first) + PointCN_i + projection_{q,k,v} of NonLocalBlock i.
 * Replaces GMF_PointDSC/models/PointDSC.py:88,104-109 (NonLocalNet) and :56-58 (NonLocalBlock).
 * in: corr_pos [B,N,6] row-major if first, else feat P32 image.  Outputs f,q,k: P32; v: T image. */
int gmf_front_forward(gmf_handle* h, int first, const float* in, const float* wst, const float* vecs, float* f,
                      float* q, float* k, float* v, int B, int N, gmf_stream_t stream);
/* Spatial-consistency self-attention + fc_message + (message + fusion2_out).
 * Replaces PointDSC.py:216-221 (compat matrix, recomputed in-kernel from pts8, never stored),
 * :60-65 and :73. */
int gmf_scattn_forward(gmf_handle* h, const float* q, const float* k, const float* v, const float* pts8,
                       const float* fusion2_out, const float* wst, const float* vecs, float* out, int B, int N,
                       float sigma_d, gmf_stream_t stream);
/* Same, but with the caller's dense compatibility matrix `attention` [B,N,N] (row-major) as
 * NonLocalBlock.forward receives it (PointDSC.py:40-45,62) instead of recomputing it from key points. */
int gmf_scattn_forward_dense(gmf_handle* h, const float* q, const float* k, const float* v, const float* attention,
                             const float* fusion2_out, const float* wst, const float* vecs, float* out, int B, int N,
                             gmf_stream_t stream);
/* Context side of FusionLayer: [LCPE] + LayerNorm_context + to_kv, for `sets` weight sets at once.
 * Replaces GMF_PointDSC/models/fusion_layer.py:124-126,46-49,86-87 (DGR twin: model/perceiver_io.py:126-128,89-91).
 * ctx: P32 image [B, Tt, 32*128]; out: [sets, B, Tt, 4096] (K image | V image per 32-token tile). */
int gmf_fusion_ctx_prepare(gmf_handle* h, int pe, const float* ctx, const float* wst, const float* vecs, float* out,
                           int B, int T, int sets, int wst_stride, int vec_stride, gmf_stream_t stream);
/* Query side up to the first residual: [LCPE] + LayerNorm + to_q + softmax(QK^T)V + to_out + x.
 * Replaces fusion_layer.py:119-121,44,84-94,190. */
int gmf_fusion_attn_forward(gmf_handle* h, int pe, const float* x, const float* ctx_img, const float* wst,
                            const float* vecs, float* x1, int B, int N, int T, gmf_stream_t stream);
/* LayerNorm + Linear(128,1024) + GEGLU + Linear(512,128) + residual.  Replaces fusion_layer.py:54-69,191. */
int gmf_fusion_ff_forward(gmf_handle* h, const float* x1, const float* wst, const float* vecs, float* x2, int B,
                          int N, gmf_stream_t stream);
/* classification head + F.normalize.  Replaces PointDSC.py:175-181,229,241.
 * Outputs are row-major: logits [B,N], feat_n [B,N,128], feat [B,N,128] (feat may be NULL). */
int gmf_classifier_forward(gmf_handle* h, const float* feat_img, const float* wst, const float* vecs, float* logits,
                           float* feat_n, float* feat, int B, int N, gmf_stream_t stream);

/* ---- whole encoder ---------------------------------------------------------------------------
 * Packed weights of PointDSC's NonLocalNet (minus the ResNet image encoder, which is upstream of the
 * hot path) + classifier.  All pointers are device pointers; *_stride are in floats between layers. */
typedef struct gmf_encoder_weights {
  int num_layers;
  const float* f1_ctx_wst;  const float* f1_ctx_vec;      /* Fusion-1 context side (pe = 0)  */
  const float* f1_attn_wst; const float* f1_attn_vec;     /* Fusion-1 query side             */
  const float* f1_ff_wst;   const float* f1_ff_vec;
  const float* ctx_wst;     const float* ctx_vec;   int ctx_wst_stride,   ctx_vec_stride;   /* Fusion-2, per layer */
  const float* attn_wst;    const float* attn_vec;  int attn_wst_stride,  attn_vec_stride;
  const float* ff_wst;      const float* ff_vec;    int ff_wst_stride,    ff_vec_stride;
  const float* front_wst;   const float* front_vec; int front_wst_stride, front_vec_stride;
  const float* tail_wst;    const float* tail_vec;  int tail_wst_stride,  tail_vec_stride;  /* fc_message */
  const float* head_wst;    const float* head_vec;
  float sigma_d;
  /* optional split-fp16 (fp16x2) images of every dense weight of the linear stages, same sizes and strides as the
   * fp32 blobs.  When all are non-NULL (tail_wst_h2 included) and the attention variant is 9 / 18 (the default),
   * [layer0]+PointCN+QKV, the context prepare, the cross-attention, the feed-forward and the attention run on the f16
   * MFMA with split-fp16 operands (fp32-equivalent accuracy); otherwise every stage runs on the fp32 MFMA. */
  const float* front_wst_h2; const float* ctx_wst_h2; const float* attn_wst_h2; const float* ff_wst_h2;
  const float* f1_ctx_wst_h2; const float* f1_attn_wst_h2; const float* f1_ff_wst_h2;
  /* optional split-fp16 image of the fc_message weights (same size and stride as tail_wst): the epilogue of the cached,
   * software-pipelined attention kernel then runs on the f16 MFMA too. */
  const float* tail_wst_h2;
  /* [ABI 5] optional, [num_layers]: the "pv_fp8" guard.  Entry l is the largest SQUARED row norm of layer l's input features
   * f = ReLU(PointCN_l(.)) up to which that layer's attention may run its P V cross products on the fp8 pipe (written by
   * gmf_encoder_pack_weights from the spectral norms of projection_q / projection_k, PointDSC.py:23-25,56-64); NULL = no thresholds:
   * the guarded default ("pv_fp8" = 1) then runs the three-product form (as "pv_fp8" = 0), never the unguarded one. */
  const float* pv_guard;
} gmf_encoder_weights;

/* ---- weight packing ---------------------------------------------------------------------------
 * The reference keeps its weights as a PyTorch state_dict (588 non-image tensors under the names of SURVEY.md section 8b) and
 * has no packed form; the kernels stream blobs (P32 images, BatchNorm folded, softmax scales folded, split-fp16 planes).
 * These entry points build the blobs from the state_dict tensors themselves, so that a host in any language can go from a
 * checkpoint to gmf_encoder_forward: list the tensors by their reference names, pack once, run.
 * Tensors are contiguous fp32; shapes as torch stores them (conv1x1 weights [out, in, 1], depthwise taps [C, 1, 3]). */
typedef struct gmf_tensor {
  const char* name;          /* state_dict key, e.g. "encoder.blocks.NonLocal_layer_3.projection_q.weight" */
  const float* data;         /* host pointer (default) or device pointer (GMF_PACK_DEVICE_TENSORS) */
  int ndim;                  /* 0 .. 4 */
  long long shape[4];
} gmf_tensor;
#define GMF_PACK_DEVICE_TENSORS 1     /* `data` pointers are device pointers (copied to the host first) */
#define GMF_PACK_STANDALONE_BLOCK 2   /* NonLocalBlock on its own (PointDSC.py:40-74): no PointCN in front (identity) */
#define GMF_PACK_HOST_BLOCK 4         /* [ABI 4] keep the packed block in host memory although h is given (errors then still land in
                                       * gmf_last_error_string(h)); gmf_packed_*_place moves it to caller-owned device memory */
typedef struct gmf_packed_encoder gmf_packed_encoder;
/* Packs NonLocalNet (Fusion-1 if present, layer0, num_layers x {PointCN, NonLocalBlock with its Fusion-2}) + classifier
 * (PointDSC.py:77-181) into ONE library-owned device block on h's device and fills a gmf_encoder_weights that points into it.
 * Tensors the list does not need are ignored (the image encoder, num_batches_tracked); a missing one is GMF_ERR_BAD_ARG with
 * its name in gmf_last_error_string.  h may be NULL: the blobs then stay in host memory (inspection / tests; not for kernels).
 * A weight outside the fp16 range of the split operands (|256 w| > 65504, e.g. a BatchNorm with a tiny running_var) leaves the
 * *_h2 pointers NULL: every stage then runs on the fp32 MFMA; gmf_packed_encoder_info reports it. */
int gmf_encoder_pack_weights(gmf_handle* h, const gmf_tensor* tensors, int n_tensors, int num_layers, int flags,
                             gmf_packed_encoder** out);
const struct gmf_encoder_weights* gmf_packed_encoder_weights(const gmf_packed_encoder* p);
/* sigma (PointDSC.py:164, for gmf_pose_params), sigma_d (= sigma_spat, :165), whether the split-fp16 images were built, and
 * the largest |value| met while splitting (inf: a weight was not finite).  Any output pointer may be NULL. */
int gmf_packed_encoder_info(const gmf_packed_encoder* p, float* sigma, float* sigma_d, int* split_fp16, float* max_abs_scaled);
void gmf_packed_encoder_free(gmf_packed_encoder* p);
/* [ABI 4] The packed block in CALLER-OWNED device memory (a host that has its own allocator - PyTorch's caching allocator,
 * an arena - should not get a hipMalloc per pack and a device-wide hipFree per release).  Pack with h = NULL (the blobs stay in host
 * memory), ask for the size, then place: the block is copied to `device_dst` (256-byte aligned, >= gmf_packed_encoder_bytes, on h's
 * device) asynchronously on `stream`, and the object's gmf_encoder_weights is re-based to point into it.  The library never frees
 * `device_dst`; it must outlive every call that uses the weights.  May be called again to move the weights. */
long long gmf_packed_encoder_bytes(const gmf_packed_encoder* p);
int gmf_packed_encoder_place(gmf_handle* h, gmf_packed_encoder* p, void* device_dst, long long bytes, gmf_stream_t stream);

/* One FusionLayer / PerceiverIO with depth = 0 (fusion_layer.py:131-201, perceiver_io.py:139-221; widths (128, 64) and
 * (256, 128)): the arguments of gmf_fusion_layer_forward.  `prefix` is prepended to the module's own key names
 * ("cross_attend_blocks.0.fn.to_q.weight", "cpe.proj_q.weight", ...); pe = whether the layer has a `cpe`. */
typedef struct gmf_fusion_weights {
  int latent_dim, d_head, pe, split_fp16;
  float max_abs_scaled;
  const float *ctx_wst, *ctx_vec, *attn_wst, *attn_vec, *ff_wst, *ff_vec;
  const float *ctx_wst_h2, *attn_wst_h2, *ff_wst_h2;      /* NULL when a weight is outside the fp16 range */
} gmf_fusion_weights;
typedef struct gmf_packed_fusion gmf_packed_fusion;
int gmf_fusion_pack_weights(gmf_handle* h, const gmf_tensor* tensors, int n_tensors, const char* prefix, int pe, int flags,
                            gmf_packed_fusion** out);
const gmf_fusion_weights* gmf_packed_fusion_weights(const gmf_packed_fusion* p);
void gmf_packed_fusion_free(gmf_packed_fusion* p);
long long gmf_packed_fusion_bytes(const gmf_packed_fusion* p);                          /* [ABI 4] as gmf_packed_encoder_bytes / _place */
int gmf_packed_fusion_place(gmf_handle* h, gmf_packed_fusion* p, void* device_dst, long long bytes, gmf_stream_t stream);

/* PointDSC.forward up to the logits (PointDSC.py:216-241) with image TOKENS as input:
 * corr_pos [B,N,6], src/tgt_keypts [B,N,3], p_tokens/q_tokens [B,T,128] (row-major)
 * -> logits [B,N], feat_n [B,N,128] (unit rows), feat [B,N,128] (may be NULL). */
int gmf_encoder_forward(gmf_handle* h, const gmf_encoder_weights* w, const float* corr_pos, const float* src_keypts,
                        const float* tgt_keypts, const float* p_tokens, const float* q_tokens, int B, int N, int T,
                        float* logits, float* feat_n, float* feat, gmf_stream_t stream);

/* The same for a RAGGED batch: B pairs, each with its OWN number of correspondences, in one launch.  The reference feeds its
 * evaluation loop one pair at a time, each with its own N (evaluation/test_3DMatch.py:24-119, PointDSC.py:279,504 assert
 * B == 1); its collate function clips a training batch to the smallest N (datasets/dataloader.py:6-23).  Here the row-major
 * tensors are PACKED: pair b owns rows [n_0 + .. + n_{b-1}, + n_b) of corr_pos [sum n, 6], src/tgt_keypts [sum n, 3], logits
 * [sum n], feat_n / feat [sum n, 128]; the tokens stay [B, T, 128].  n_points: HOST array [B] (the caller knows its tensor
 * shapes; the library sizes its grids from it).  The result equals B calls with B = 1, pair by pair, whatever the order of the pairs
 * (the library deals them to the chip's eight dies by work; small batches run the same small-grid kernels as uniform ones).  Default path only (split-
 * fp16 weight images, num_layers >= 2, "fused_linear" = 1, "scattn_variant" = 18): otherwise GMF_ERR_UNSUPPORTED_SHAPE. */
int gmf_encoder_forward_ragged(gmf_handle* h, const gmf_encoder_weights* w, const float* corr_pos, const float* src_keypts,
                               const float* tgt_keypts, const float* p_tokens, const float* q_tokens, const int* n_points, int B,
                               int T, float* logits, float* feat_n, float* feat, gmf_stream_t stream);

/* One NonLocalBlock on P32 images (PointDSC.py:40-74): feat_img -> out_img given the fused image tokens
 * (P32 image [B,Tt,..]) and EITHER pts8 (compat recomputed in-kernel) OR the dense `attention` [B,N,N]
 * (exactly one of the two non-NULL).  layer selects the weight set inside `w`. */
int gmf_nonlocal_block_forward(gmf_handle* h, const gmf_encoder_weights* w, int layer, int apply_pointcn,
                               const float* feat_img, const float* pts8, const float* attention,
                               const float* image_feat_img, float* out_img, int B, int N, int T, gmf_stream_t stream);

/* FusionLayer.forward / PerceiverIO.forward with depth=0 (fusion_layer.py:172-201; DGR twin
 * model/perceiver_io.py:187-221): data [B,T,128] (context, row-major), queries [B,N,latent_dim] (any strides)
 * -> out [B,N,latent_dim].  (latent_dim, d_head) = (128, 64) is GMF-PointDSC's Fusion-1/2 and DGR's
 * image_fusion (resunet_new.py:618-626); (256, 128) is the DGR bottleneck instance (resunet_new.py:516-525)
 * whose to_out maps the head back to the 256-wide query.  Weight blobs: gmf_amd/packing.py.
 * ff_wst_h2 (may be NULL): the feed-forward weights as split-fp16 images (packing.p32_h2, same size as ff_wst); when
 * given, the GEGLU feed-forward - 96 % of this layer's FLOPs - runs on the f16 MFMA with fp32-equivalent accuracy.
 * ctx_wst_h2 / attn_wst_h2 (both or neither): the context and attention weights as
 * split-fp16 images (packing.p32_h2s, same sizes) - context preparation and cross-attention on the f16 MFMA as well. */
int gmf_fusion_layer_forward(gmf_handle* h, int pe, int latent_dim, int d_head, const float* ctx_wst, const float* ctx_vec,
                             const float* attn_wst, const float* attn_vec, const float* ff_wst, const float* ff_vec,
                             const float* data, const float* queries, long long q_sb, long long q_sr, long long q_sk,
                             float* out, long long o_sb, long long o_sr, long long o_sk, int B, int N, int T,
                             gmf_stream_t stream, const float* ff_wst_h2, const float* ctx_wst_h2, const float* attn_wst_h2);

/* ---- pose head ------------------------------------------------------------------------------- */
typedef struct gmf_pose_params {
  int num_seeds;        /* S = int(N * ratio)                        PointDSC.py:244 */
  int k;                /* neighbours per seed (<= 64, <= N-1)        PointDSC.py:324 */
  int num_iterations;   /* power iterations                          PointDSC.py:437 */
  int use_nms;          /* 1: pick_seeds (test mode) ; 0: plain top-S (train mode)  PointDSC.py:243-246 */
  int refine_iters;     /* 20 in test mode, 0 to skip post_refinement PointDSC.py:256-257,505-508 */
  float sigma;          /* learned feature bandwidth                 PointDSC.py:164 */
  float sigma_d;        /* spatial bandwidth                         PointDSC.py:165 */
  float inlier_threshold;
  float nms_radius;
  float refine_threshold; /* 0.10 if inlier_threshold == 0.10 else 1.2 PointDSC.py:505-508 */
} gmf_pose_params;

/* seeds -> kNN -> seed compatibility + power iteration -> weighted Kabsch -> hypothesis scoring ->
 * argmax -> labels -> post refinement, per pair, all on device.
 * Replaces PointDSC.py:243-257 (pick_seeds :268-286, cal_seed_trans :303-427, cal_leading_eigenvector
 * :429-448, post_refinement :493-528) and models/common.py:10-75.
 * seeds_in may be NULL (seeds are computed from logits) or caller-provided [B,S].
 * Outputs: final_trans [B,16], final_labels [B,N], and optionally (may be NULL) seeds_out [B,S] int32,
 * knn_out [B,S,k] int32, seed_trans [B,S,16], fitness [B,S]. */
int gmf_pose_head(gmf_handle* h, const gmf_pose_params* p, const float* feat_n, const float* src_keypts,
                  const float* tgt_keypts, const float* logits, const int* seeds_in, int B, int N, float* final_trans,
                  float* final_labels, int* seeds_out, int* knn_out, float* seed_trans, float* fitness,
                  gmf_stream_t stream);

/* gmf_pose_head for a ragged batch (packing as gmf_encoder_forward_ragged; n_points: HOST array [B]): pair b uses
 * S_b = int(n_b * ratio) seeds (PointDSC.py:244; p->num_seeds is ignored) and is solved exactly as a B = 1 call - the power
 * iteration's allclose exit (PointDSC.py:444) is resolved per pair.  Every pair needs more than p->k correspondences.
 * final_trans [B,16]; final_labels [sum n]; the optional per-seed outputs are [B, S_max, ...] with S_max = max S_b, the slots
 * behind a pair's own seeds zero. */
int gmf_pose_head_ragged(gmf_handle* h, const gmf_pose_params* p, double ratio, const float* feat_n, const float* src_keypts,
                         const float* tgt_keypts, const float* logits, const int* n_points, int B, float* final_trans,
                         float* final_labels, int* seeds_out, int* knn_out, float* seed_trans, float* fitness,
                         gmf_stream_t stream);

/* pick_seeds / plain top-S only (PointDSC.py:268-286 / :246). */
int gmf_pick_seeds(gmf_handle* h, const float* src_keypts, const float* scores, int B, int N, float nms_radius,
                   int use_nms, int num_seeds, int* seeds_out, gmf_stream_t stream);

/* knn(x, k, ignore_self=True, normalized=True) restricted to the rows listed in `rows` [B,S]
 * (models/common.py:53-75 followed by the gather at PointDSC.py:327-329): feat_n [B,N,128] unit rows
 * -> knn_out [B,S,k] int32, nearest first, the row itself (rank 0) dropped.  Any N for k <= 63 (the rows' distances come from the
 * matrix pipe, in slices of at most 4 GiB; workspace: the handle's arena); k > 63 keeps the in-LDS form (N <= 38400). */
int gmf_knn_rows(gmf_handle* h, const float* feat_n, const int* rows, int B, int N, int S, int k, int* knn_out,
                 gmf_stream_t stream);

/* [ABI 5] The selection step alone: the k smallest entries behind rank 0 of every given DISTANCE row, under the order (distance,
 * index) - `topk(k + 1, largest=False)[1][..., 1:]` of models/common.py:71-74 for rows the caller has formed itself (gmf_amd.knn with
 * `normalized=False` or a feature width other than 128: distances xx_i - 2 x_i.x_j + xx_j from gmf_gemm_f32, common.py:64-69).
 * dist [B, S, ld] with ld = 32 * ceil(N / 32) floats per row (columns N .. ld - 1 are not read) -> knn_out [B, S, k] int32,
 * nearest first.  0 < k <= 63, k <= N - 1; any N. */
int gmf_knn_from_distances(gmf_handle* h, const float* dist, int B, int N, int S, int k, int* knn_out, gmf_stream_t stream);

/* Descriptor-space nearest neighbour: for every row of F0 [N0,d] the closest row of F1 [N1,d] (row-major, d <= 128),
 * ties to the lower index, fused distance GEMM + row argmin (no N0 x N1 matrix).
 *   mode 0: PointDSC matching (datasets/ThreeDMatch.py:164-166, demo_registration.py:101-103), unit descriptors:
 *           dist = sqrt(2 - 2 <a,b> + 1e-6);
 *   mode 1: DGR find_knn_gpu with nn_max_n > 1 (core/knn.py:50-64): dist = sqrt(||a-b||^2 + 1e-7) (pdist 'L2');
 *   mode 2: DGR find_knn_gpu with nn_max_n <= 1 (core/knn.py:66-70): dist = ||a-b||^2 (pdist 'SquareL2').
 * idx_out [N0] int32, dist_out [N0]. */
int gmf_nn_match(gmf_handle* h, const float* F0, const float* F1, int N0, int N1, int d, int mode, int* idx_out,
                 float* dist_out, gmf_stream_t stream);

/* rigid_transform_3d(A, B, weights, weight_threshold) (models/common.py:10-50):
 * A,B [n,k,3], weights [n,k] or NULL -> T [n,4,4].  The 3x3 SVD runs on the device. */
int gmf_procrustes_batched(gmf_handle* h, const float* A, const float* B, const float* weights, int n, int k,
                           float weight_threshold, float* T44, gmf_stream_t stream);

/* post_refinement (PointDSC.py:493-528) for B pairs independently: T_in/out [B,16]. */
int gmf_post_refinement(gmf_handle* h, const float* T_in, const float* src_keypts, const float* tgt_keypts, int B, int N,
                        float refine_threshold, int iters, float* T_out, gmf_stream_t stream);

/* DGR weighted_procrustes(X, Y, w, eps) (GMF_DeepGlobalRegistration/.../core/registration.py:91-113),
 * batched over B pairs with ragged sizes: offsets [B+1] (device int32) delimit rows of X,Y [sum N,3], w [sum N].
 * R [B,9], t [B,3].  fp64 SVD on device, no host transfer. */
int gmf_weighted_procrustes(gmf_handle* h, const float* X, const float* Y, const float* w, const int* offsets, int B,
                            float eps, float* R, float* t, gmf_stream_t stream);

/* DGR GlobalRegistration(points, trans_points, weights, max_iter, ..., max_break_count, break_threshold_ratio, ...,
 * quantization_size) (GMF_DeepGlobalRegistration/.../core/registration.py:135-194; loss core/loss.py:42-61;
 * caller core/deep_global_registration.py:334-340): weighted-Procrustes initialisation + robust Adam refinement of a 6-D
 * rotation and a translation, batched over B ragged pairs like gmf_weighted_procrustes (offsets [B+1] device int32;
 * X,Y [sum N,3]; w [sum N] or NULL for the unweighted form).  eps = the loss's eps (float32 machine epsilon in the
 * reference).  R [B,9], t [B,3], stats [B,3] = {iterations, loss, break_count}.  max_points: the largest per-pair point
 * count if the host knows it (sizes the workgroup: 16 register-resident points per thread), 0 otherwise.  One persistent
 * kernel, no host sync. */
int gmf_global_registration(gmf_handle* h, const float* X, const float* Y, const float* w, const int* offsets, int B,
                            float eps, float quantization_size, int max_iter, int max_break_count,
                            double break_threshold_ratio, float* R, float* t, float* stats, int max_points,
                            gmf_stream_t stream);

/* ---- validation step: the non-test forward's similarity matrix and the metrics of libs/trainer.py:194-262 ---------- */

/* M = clamp(1 - (1 - Fn Fn^T) / sigma^2, 0, 1) with a zero diagonal (GMF_PointDSC/models/PointDSC.py:231-234):
 * feat_n [B,N,128] unit rows -> M [B,N,N] with row stride ldm >= N floats (element (b,i,j) at M[(b*N + i)*ldm + j];
 * the padding columns are not written).  Split-fp16 MFMA product, fp32 accumulate; writing M is the bound, and a row
 * stride that is a multiple of 32 floats (whole 128-byte lines per row piece) nearly doubles the store rate. */
int gmf_similarity_matrix(gmf_handle* h, const float* feat_n, int B, int N, float sigma, float* M, int ldm,
                          gmf_stream_t stream);

/* SpectralMatchingLoss.forward(M, gt_labels) (GMF_PointDSC/libs/loss.py:116-140): M [B,N,N] with row stride ldm,
 * gt_labels [B,N] float 0/1 -> loss_out [1] (device).  balanced != 0 is the reference's default form. */
int gmf_spectral_matching_loss(gmf_handle* h, const float* M, int ldm, const float* gt_labels, int B, int N, int balanced,
                               float* loss_out, gmf_stream_t stream);

/* The same loss straight from the features: gmf_similarity_matrix followed by gmf_spectral_matching_loss without M
 * ever reaching memory (upper triangle of tiles only, reduced on the fly). */
int gmf_spectral_matching_loss_fused(gmf_handle* h, const float* feat_n, const float* gt_labels, int B, int N, float sigma,
                                     int balanced, float* loss_out, gmf_stream_t stream);

/* Backward of gmf_spectral_matching_loss_fused (first slice of the training path, libs/trainer.py:158): the gradient of
 * SpectralMatchingLoss(M(feat_n, sigma), gt_labels) (PointDSC.py:231-234, libs/loss.py:116-140) for an upstream gradient
 * of 1, with respect to feat_n -> d_feat_n [B,N,128] and to the bandwidth sigma -> d_sigma [1] (both device, fp32).  M and
 * dL/dM are never written: S = Fn Fn^T tile by tile, dL/dFn = 2 G Fn with G = dL/dM * [0 <= u <= 1] / sigma^2 (clamp's
 * gradient mask), split-fp16 MFMA products with fp32 accumulation, dsigma summed in fp64 in a fixed order. */
int gmf_spectral_matching_backward(gmf_handle* h, const float* feat_n, const float* gt_labels, int B, int N, float sigma,
                                   int balanced, float* d_feat_n, float* d_sigma, gmf_stream_t stream);

/* ClassificationLoss.forward(pred, gt, weight) (GMF_PointDSC/libs/loss.py:67-113): pred [B,N] logits, gt [B,N] float
 * 0/1, weight [B,N] or NULL -> out [6] (device) = loss, precision, recall, f1 (pair 0, as loss.py:99-101), mean logit of
 * the inliers, mean logit of the outliers. */
int gmf_classification_loss(gmf_handle* h, const float* pred, const float* gt, const float* weight, int B, int N,
                            int balanced, float* out6, gmf_stream_t stream);

/* TransformationLoss.forward(trans, gt_trans, src_keypts, tgt_keypts, probs) (GMF_PointDSC/libs/loss.py:12-64):
 * trans, gt_trans [B,4,4]; src_keypts, tgt_keypts [B,N,3]; probs [B,N] -> out [5] (device) = loss, recall (%), RE (deg),
 * TE (cm), RMSE.  As the reference, pair i's warped source points are compared with the target points of every pair of
 * the batch (loss.py:47-48,61 broadcast [N,3] against [bs,N,3]). */
int gmf_transformation_loss(gmf_handle* h, const float* trans, const float* gt_trans, const float* src_keypts,
                            const float* tgt_keypts, const float* probs, int B, int N, float re_thre, float te_thre,
                            float* out5, gmf_stream_t stream);

/* ---- training primitives (row f-4) -----------------------------------------------------------------------------------
 * What the training forward + backward of the encoder are made of (FusionLayer / PerceiverIO fusion_layer.py:32-128,172-201;
 * PointCN, NonLocalBlock and classifier PointDSC.py:10-74,104-109,175-181 with TRAIN-mode BatchNorm; the two losses the
 * reference trains with by default, libs/loss.py:67-140, config_3DMatch.py:50-52), on plain row-major fp32 tensors;
 * gmf_amd/train.py composes them into torch.autograd.Functions.  Contractions run on the fp32 MFMA (exact
 * fp32 products; gradients of magnitude 1e-8 need no operand scaling), every cross-row sum is taken in a fixed order. */

/* C[b] = alpha * op(A[b]) op(B[b]) (+ bias[col]) (+ residual[b]) (-> ReLU if relu != 0) for b < batch; op(X) = X^T when trans_x != 0.  op(A) is
 * M x K, op(B) is K x N; lda / ldb / ldc are row strides and stride_* batch strides, in floats (residual shares C's layout).
 * Few output tiles with a long contraction (weight gradients: K = every row of the batch) are split over K into partials
 * that are added in index order. */
int gmf_gemm_f32(gmf_handle* h, int trans_a, int trans_b, const float* A, const float* B, float* C, const float* bias,
                 const float* residual, int M, int N, int K, long long lda, long long ldb, long long ldc, long long stride_a,
                 long long stride_b, long long stride_c, int batch, float alpha, int relu, gmf_stream_t stream);
/* LCPE (fusion_layer.py:118-128) on x [rows, C] = sequences of L rows: forward y = x + bias + depthwise conv3(x) with taps
 * w [C,1,3]; backward (x := dy, y := dx, bias unused) the transposed taps: dx[l] = dy[l] (1 + w1) + w0 dy[l+1] + w2 dy[l-1]. */
int gmf_lcpe(gmf_handle* h, int backward, const float* x, const float* w, const float* bias, float* y, int rows, int L, int C,
             gmf_stream_t stream);
/* nn.LayerNorm over the last dim (eps 1e-5): y, and mean / rstd [rows] for the backward.  Backward: dx = the input gradient
 * (+ dx_add when non-NULL: the residual branch's gradient). */
int gmf_layernorm_forward(gmf_handle* h, const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          long long rows, int C, gmf_stream_t stream);
int gmf_layernorm_backward(gmf_handle* h, const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* dx_add, float* dx, long long rows, int C, gmf_stream_t stream);
/* forward: out = softmax(mul * scale * a) per row of length T (mul: optional element-wise multiplier of the logits, the
 * compat matrix of the spatial-consistency attention, PointDSC.py:60-62); backward: a = P, b = dP -> out = dS =
 * scale * mul * P * (dP - <dP, P>). */
int gmf_softmax_rows(gmf_handle* h, int backward, const float* a, const float* b, const float* mul, float* out, long long rows, int T,
                     float scale, gmf_stream_t stream);
/* GEGLU (fusion_layer.py:54-57): hdn [rows, 2H] -> out [rows, H] = x * gelu_erf(gates); backward: dg [rows, H] -> out = dhdn [rows, 2H]. */
int gmf_geglu(gmf_handle* h, int backward, const float* hdn, const float* dg, float* out, long long rows, int H,
              gmf_stream_t stream);
/* out[c] = sum over rows r of x'[r][c] * y'[r + shift][c] with y' = 1 (y NULL), y, (y - mean[r]) * rstd[r] (LayerNorm's xhat)
 * or (y - cmean[c]) * crstd[c] (BatchNorm's; crstd NULL = 1), x' = x or x - cmean[c] (center_x); rows r + shift outside r's
 * sequence of L rows contribute 0.  Bias, LayerNorm / BatchNorm parameter and LCPE-tap gradients, BatchNorm statistics.
 * relu_y (optional): x is first masked by the saved output of a ReLU (x where relu_y > 0, else 0).  dual != 0 (needs y): out
 * holds 2C values, [sum x' y' | sum x'] - a normalisation's dgamma and dbeta, or an LCPE tap and its bias, in one pass. */
int gmf_colsum(gmf_handle* h, const float* x, const float* y, const float* mean, const float* rstd, const float* cmean,
               const float* crstd, int center_x, int shift, int L, long long rows, int C, const float* relu_y, int dual, float* out,
               gmf_stream_t stream);
/* nn.BatchNorm1d in TRAINING mode over the rows of x [rows, C] (= BatchNorm1d on [B, C, N]): batch statistics (mean / rstd [C]
 * returned for the backward), running statistics updated as torch does (momentum, unbiased variance; may be NULL), optional
 * fused ReLU.  Backward: dx, dgamma, dbeta; y_relu (the saved output) masks dy when the forward applied the ReLU, else NULL. */
int gmf_batchnorm_train_forward(gmf_handle* h, const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                float* rstd, float* running_mean, float* running_var, long long rows, int C, float eps,
                                float momentum, int relu, gmf_stream_t stream);
int gmf_batchnorm_train_backward(gmf_handle* h, const float* dy, const float* x, const float* y_relu, const float* mean,
                                 const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, long long rows, int C,
                                 gmf_stream_t stream);
/* F.normalize(x, p=2, dim=-1) (PointDSC.py:229) on rows [rows, C]: forward a = x -> out = y, nrm [rows] = max(||x||, 1e-12);
 * backward a = y (the saved output), dy, nrm -> out = dx = (dy - y <dy, y>) / nrm. */
int gmf_normalize_rows(gmf_handle* h, int backward, const float* a, const float* dy, float* nrm, float* out, long long rows, int C,
                       gmf_stream_t stream);
/* out = y > 0 ? dy : 0 (ReLU backward from the saved output). */
int gmf_relu_backward(gmf_handle* h, const float* dy, const float* y, float* out, long long total, gmf_stream_t stream);
/* d loss / d pred of ClassificationLoss (libs/loss.py:67-93) for an upstream gradient of 1: mean BCE-with-logits, balanced
 * (pos_weight = num_neg / num_pos over the batch) or not, or weighted per element (weight non-NULL). */
int gmf_classification_backward(gmf_handle* h, const float* pred, const float* gt, const float* weight, int B, int N, int balanced,
                                float* d_pred, gmf_stream_t stream);
/* dL/dM [B,N,N] (dense) of SpectralMatchingLoss(M, gt_labels) (libs/loss.py:116-140), M with row stride ldm. */
int gmf_spectral_matching_dense_backward(gmf_handle* h, const float* M, int ldm, const float* gt_labels, int B, int N, int balanced,
                                         float* dM, gmf_stream_t stream);
/* Backward of gmf_similarity_matrix for an arbitrary upstream dM [B,N,N] (dense): d_feat_n [B,N,128] and d_sigma [1]
 * (PointDSC.py:231-234: matmul, clamp with its gradient mask, zero diagonal).  S and G are materialised in the workspace
 * (training sizes: N ~ 1000); the loss-specific fused form without them is gmf_spectral_matching_backward. */
int gmf_similarity_backward(gmf_handle* h, const float* feat_n, const float* dM, int B, int N, float sigma, float* d_feat_n,
                            float* d_sigma, gmf_stream_t stream);
/* Dense compat matrix [B,N,N] row-major, c_ij = clamp(1 - (|p_i-p_j| - |q_i-q_j|)^2 / sigma_d^2, min=0) (PointDSC.py:216-221;
 * the trainable path materialises it as the reference does; the inference path streams it in kernel order instead). */
int gmf_compat_dense(gmf_handle* h, const float* src_keypts, const float* tgt_keypts, int B, int N, float sigma_d, float* out,
                     gmf_stream_t stream);
/* d loss / d trans [B,4,4] of TransformationLoss (libs/loss.py:35-64) for an upstream gradient of 1:
 * loss = (1/B) sum_i [any(probs_i > 0)] mean_{b,n} |R_i p_in + t_i - q_bn|^2 (the reference's broadcast over the batch). */
int gmf_transformation_loss_backward(gmf_handle* h, const float* trans, const float* src_keypts, const float* tgt_keypts,
                                     const float* probs, int B, int N, float* d_trans, gmf_stream_t stream);
/* Backward of gmf_pose_head in train mode (use_nms = 0, refine_iters = 0; PointDSC.py:246,252,330-425): final_trans is the
 * hypothesis of the best seed of each pair, so d_final_trans [B,4,4] flows through that seed's weighted Kabsch solve
 * (closed-form derivative of the 3x3 SVD), its weights, the power iterations in reverse and the feature compatibility
 * matrix to the k neighbour rows of feat_n and to sigma.  knn_idx [B,S,k] and fitness [B,S] are gmf_pose_head's outputs for
 * the same inputs.  d_feat_n [B,N,128] is written whole (zero outside the neighbour rows), d_sigma [B] per pair. */
int gmf_pose_head_backward(gmf_handle* h, const gmf_pose_params* p, const float* feat_n, const float* src_keypts,
                           const float* tgt_keypts, const int* knn_idx, const float* fitness, const float* d_final_trans, int B,
                           int N, float* d_feat_n, float* d_sigma, gmf_stream_t stream);
/* Backward of gmf_weighted_procrustes with respect to the weights (DGR trains its inlier network through this solve,
 * core/trainer.py:594-614; X and Y carry no gradient there): d_R [B,3,3], d_t [B,3] -> d_w (ragged like w). */
int gmf_weighted_procrustes_backward(gmf_handle* h, const float* X, const float* Y, const float* w, const int* offsets, int B,
                                     float eps, const float* d_R, const float* d_t, float* d_w, gmf_stream_t stream);

/* ---- image encoder epilogue (ResNet BasicBlock, GMF_PointDSC/models/resnet.py:59-75) ------------------------------- */

/* y = max(y + bias[c] (+ residual), 0) in place on an NHWC fp32 tensor of n_pixels x C (C % 4 == 0): the folded
 * BatchNorm bias, the residual add and the ReLU that follow each MIOpen convolution, in one pass.  residual may be NULL. */
int gmf_bias_relu_nhwc(gmf_handle* h, float* y, const float* bias, const float* residual, long long n_pixels, int C,
                       gmf_stream_t stream);

/* The ResNet stem in one kernel (GMF_PointDSC/models/resnet.py:198-204): conv1 7x7 stride 2 pad 3 (3 -> 64 channels) with
 * the BatchNorm folded in, ReLU and max-pool 3x3 stride 2 pad 1.  x [B,3,H,W] fp32 with element strides (sb, sc, sh, sw) -
 * NCHW as the reference passes it, or any view; y [B,Hp,Wp,64] NHWC fp32 with Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W).
 * wimg: the folded weights as a split-fp16 image of 256 W (gmf_amd/packing.py: stem_image), bias [64].  Implicit GEMM on the
 * f16 MFMA with split-fp16 operands, fp32 accumulate; the pooled tile is written once. */
int gmf_stem_forward(gmf_handle* h, const float* x, long long sb, long long sc, long long sh, long long sw, const float* wimg,
                     const float* bias, float* y, int B, int H, int W, gmf_stream_t stream);

/* Convolution of the ResNet-34 layer1 / layer2 shapes (GMF_PointDSC/models/resnet.py:36-75, 144-170) as an implicit GEMM on
 * the f16 MFMA with split-fp16 operands: x [B,H,W,cin] NHWC fp32 -> y [B,Ho,Wo,cout] NHWC fp32 (pad = ksize / 2),
 *   y = conv(x, W) + bias (+ residual [B,Ho,Wo,cout]) -> ReLU if relu != 0.
 * wimg: the BatchNorm-folded weights as a split-fp16 image of 256 W (gmf_amd/packing.py: conv_image(W, stride); k order
 * (channel block, tap) for the stride-1 3x3 shapes, (tap, channel block) for the others).  Supported:
 * (cin, cout, ksize, stride) = (64,64,3,1), (64,128,3,2), (128,128,3,1), (64,128,1,2). */
int gmf_conv_nhwc(gmf_handle* h, const float* x, const float* wimg, const float* bias, const float* residual, float* y,
                  int B, int H, int W, int cin, int cout, int ksize, int stride, int relu, gmf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GMF_HIP_H_ */
