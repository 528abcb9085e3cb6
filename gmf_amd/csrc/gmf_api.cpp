// C ABI of libgmf_hip.so (see include/gmf_hip.h).  Thin host layer: argument checks, the
// library-owned workspace arena, and the launch sequences.  No torch types, no host syncs.
#include "../../include/gmf_hip.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "api_common.hpp"
#include "launchers_pose.hpp"

extern "C" {

int gmf_abi_version(void) { return GMF_ABI_VERSION; }

int gmf_create(int device, gmf_handle** out) {
  if (!out) return GMF_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device < 0 || device >= n) return GMF_ERR_NO_DEVICE;
  gmf_handle* h = new (std::nothrow) gmf_handle();
  if (!h) return GMF_ERR_OOM;
  h->device = device;
  // the sticky status word: host memory mapped into the device's address space (read by the host without a synchronisation)
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(device);
  void* hp = nullptr;
  if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
      h->status_host = static_cast<int*>(hp);
      h->status_dev = static_cast<int*>(dp);
      *h->status_host = 0;
    } else {
      (void)hipHostFree(hp);
    }
  }
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (!h->status_dev) { delete h; return GMF_ERR_OOM; }
  *out = h;
  return GMF_OK;
}

int gmf_set_workspace(gmf_handle* h, void* device_ptr, long long bytes) {
  GMF_REQUIRE(h, GMF_ERR_BAD_ARG, "set_workspace: null handle");
  GMF_REQUIRE((device_ptr == nullptr) == (bytes == 0) && bytes >= 0, GMF_ERR_BAD_ARG, "set_workspace: pass a pointer with its size, or NULL and 0");
  GMF_REQUIRE(((uintptr_t)device_ptr & 255) == 0, GMF_ERR_BAD_ARG, "set_workspace: the block must be 256-byte aligned");
  SetDevice sd(h);
  if (!h->arena_external && h->arena) {          // the library's own block goes (hipFree synchronises the device)
    hipError_t e = hipFree(h->arena);
    h->arena = nullptr;
    h->arena_bytes = 0;
    if (e != hipSuccess) return hip_fail(h, e, "hipFree(workspace)");
  }
  h->arena = device_ptr;
  h->arena_bytes = (size_t)bytes;
  h->arena_external = device_ptr != nullptr;
  h->arena_used = 0;
  return GMF_OK;
}

long long gmf_workspace_wanted(gmf_handle* h) { return h ? (long long)h->arena_wanted : 0; }

int gmf_status_read(gmf_handle* h, int* flags, int clear) {
  GMF_REQUIRE(h && flags, GMF_ERR_BAD_ARG, "status_read: null pointer");
  int* w = h->status_host;
  *flags = clear ? __atomic_exchange_n(w, 0, __ATOMIC_RELAXED) : __atomic_load_n(w, __ATOMIC_RELAXED);
  return GMF_OK;
}

int gmf_set_sigma_device(gmf_handle* h, const float* sigma_dev) {
  GMF_REQUIRE(h, GMF_ERR_BAD_ARG, "set_sigma_device: null handle");
  std::lock_guard<std::mutex> lock(h->mu);
  h->sigma_dev = sigma_dev;
  return GMF_OK;
}

int gmf_set_tuning(gmf_handle* h, const char* name, int value) {
  GMF_REQUIRE(h && name, GMF_ERR_BAD_ARG, "set_tuning: null pointer");
  std::lock_guard<std::mutex> lock(h->mu);        // (forwards read h->tune under the same lock)
  gmf::Tuning& t = h->tune;
  if (std::strcmp(name, "scattn_variant") == 0) {
    GMF_REQUIRE(value == 0 || value == 9 || value == 18, GMF_ERR_BAD_ARG,
                "set_tuning: scattn_variant must be 18 (cached, pipelined split-fp16; default), 9 (split-fp16, not pipelined) or 0 (fp32 MFMA)");
    t.scattn_variant = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "front_output_split") == 0) {  // 1 = small grids use one workgroup per output of k_front_h2 (default), 0 = never
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: front_output_split must be 0 or 1");
    t.front_split = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "ff_hidden_splits") == 0) {    // 0 = automatic (small grids only), 1 = off, 2 / 4 / 8 = forced
    GMF_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4 || value == 8, GMF_ERR_BAD_ARG,
                "set_tuning: ff_hidden_splits must be 0, 1, 2, 4 or 8");
    t.ff_split = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "attn_key_splits") == 0) {     // 0 = automatic (small grids only), 1 = off, n = force n splits
    GMF_REQUIRE(value >= 0 && value <= 8, GMF_ERR_BAD_ARG, "set_tuning: attn_key_splits out of range (0..8)");
    t.key_splits = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "attn_tail_split") == 0) {     // 1 = split the last partial round of a large attention grid by keys (default 0)
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: attn_tail_split must be 0 or 1");
    t.tail_split = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "small_grid_roles") == 0) {    // 1 = small grids: three launches per layer with mixed workgroup roles (default), 0 = one kernel per stage
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: small_grid_roles must be 0 or 1");
    t.small_roles = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "fused_linear") == 0) {        // 1 = two launches per layer (k_linear_h2 + attention with the next PointCN; default), 0 = four
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: fused_linear must be 0 or 1");
    t.fused_linear = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "compat_cache") == 0) {
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: compat_cache must be 0 or 1");
    t.use_cache = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "conv_lds_patch") == 0) {      // 1 = stride-1 3x3 convolutions stage their activations through LDS (default), 2 = same without the three-workgroup form, 0 = gather form
    GMF_REQUIRE(value >= 0 && value <= 2, GMF_ERR_BAD_ARG, "set_tuning: conv_lds_patch must be 0, 1 or 2");
    t.conv_patch = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "small_fattn_tile") == 0) {    // [ABI 5] 1 = small grids: the cross-attention role per query tile (default), 0 = per four tiles
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: small_fattn_tile must be 0 or 1");
    t.small_fattn_tile = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "small_prologue_roles") == 0) { // [ABI 5] 1 = small grids: the prologue's image-side and point-side chains share three launches (default), 0 = six kernels
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: small_prologue_roles must be 0 or 1");
    t.small_prologue = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "conv_small_grid") == 0) {     // [ABI 5] 1 = grids of a few images run the K-split convolution kernel (default), 0 = the 128-pixel kernels at every size
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: conv_small_grid must be 0 or 1");
    t.conv_small = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "nms_binned") == 0) {          // 1 = grid-binned NMS candidates on large grids (default), 2 = always, 0 = all pairs
    GMF_REQUIRE(value >= 0 && value <= 2, GMF_ERR_BAD_ARG, "set_tuning: nms_binned must be 0, 1 or 2");
    t.nms_binned = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "topk_select") == 0) {         // 1 = radix select of the S seeds (default), 0 = full bitonic sort
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: topk_select must be 0 or 1");
    t.topk_select = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "wide_attn_tile") == 0) {      // 256-wide layer: cross-attention per tile (1, default; grids of up to 256 tiles) or per four tiles (0)
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: wide_attn_tile must be 0 or 1");
    t.wide_attn_tile = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "small_merge_tile") == 0) {    // small grids: merge step per query tile (1, default) or per four (0)
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: small_merge_tile must be 0 or 1");
    t.small_merge_tile = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "mid_grid_roles") == 0) {      // two-launch form below this many base workgroups: linear kernel as two roles (0 = never)
    GMF_REQUIRE(value >= 0 && value <= 4096, GMF_ERR_BAD_ARG, "set_tuning: mid_grid_roles out of range (0..4096)");
    t.mid_grid_roles = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "q_in_attention") == 0) {      // 1 = large grids: every attention workgroup projects its own Q' (default), 0 = k_linear_h2 writes a Q' image
    GMF_REQUIRE(value == 0 || value == 1, GMF_ERR_BAD_ARG, "set_tuning: q_in_attention must be 0 or 1");
    t.q_in_attention = value != 0;
    return GMF_OK;
  }
  if (std::strcmp(name, "pv_fp8") == 0) {       // large grids: the cross products of O += P V on the block-scaled fp8 pipe (default 1)
    GMF_REQUIRE(value >= 0 && value <= 2, GMF_ERR_BAD_ARG, "set_tuning: pv_fp8 must be 0 (three f16 products), 1 (guarded, default) or 2 (always)");
    t.pv_fp8 = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "compat_format") == 0) {       // element format of the compat cache: 0 = fp32 (default), 2 = 16-bit fixed point (opt-in)
    GMF_REQUIRE(value == 0 || value == 2, GMF_ERR_BAD_ARG, "set_tuning: compat_format must be 0 (fp32) or 2 (16-bit fixed point of c)");
    t.compat_format = value;
    return GMF_OK;
  }
  if (std::strcmp(name, "precision") == 0) {           // 0 = parity numerics (default), 1 / 2 = throughput numerics (NOT within 1e-4)
    GMF_REQUIRE(value >= 0 && value <= 2, GMF_ERR_BAD_ARG, "set_tuning: precision must be 0 (parity), 1 or 2 (throughput numerics)");
    t.precision = value;
    return GMF_OK;
  }
  return fail(h, GMF_ERR_BAD_ARG, std::string("gmf: set_tuning: unknown knob ") + name);
}

// The current value of a knob (the counterpart of gmf_set_tuning: a caller that changes a knob for one call can put back what it found).
int gmf_get_tuning(gmf_handle* h, const char* name, int* value) {
  GMF_REQUIRE(h && name && value, GMF_ERR_BAD_ARG, "get_tuning: null pointer");
  std::lock_guard<std::mutex> lock(h->mu);
  const gmf::Tuning& t = h->tune;
  const struct { const char* name; int v; } tab[] = {
      {"scattn_variant", t.scattn_variant}, {"front_output_split", t.front_split ? 1 : 0}, {"ff_hidden_splits", t.ff_split},
      {"attn_key_splits", t.key_splits}, {"attn_tail_split", t.tail_split ? 1 : 0}, {"small_grid_roles", t.small_roles ? 1 : 0},
      {"fused_linear", t.fused_linear ? 1 : 0}, {"compat_cache", t.use_cache ? 1 : 0}, {"conv_lds_patch", t.conv_patch}, {"conv_small_grid", t.conv_small ? 1 : 0}, {"small_prologue_roles", t.small_prologue ? 1 : 0}, {"small_fattn_tile", t.small_fattn_tile ? 1 : 0},
      {"nms_binned", t.nms_binned}, {"topk_select", t.topk_select ? 1 : 0}, {"wide_attn_tile", t.wide_attn_tile ? 1 : 0},
      {"small_merge_tile", t.small_merge_tile ? 1 : 0}, {"mid_grid_roles", t.mid_grid_roles}, {"pv_fp8", t.pv_fp8}, {"q_in_attention", t.q_in_attention ? 1 : 0},
      {"compat_format", t.compat_format}, {"precision", t.precision}};
  for (const auto& e : tab) {
    if (std::strcmp(name, e.name) == 0) { *value = e.v; return GMF_OK; }
  }
  return fail(h, GMF_ERR_BAD_ARG, std::string("gmf: get_tuning: unknown knob ") + name);
}

int gmf_profile_enable(gmf_handle* h, int on) {
  GMF_REQUIRE(h, GMF_ERR_BAD_ARG, "profile_enable: null handle");
  h->profile = (on != 0);
  h->prof_used = 0;
  return GMF_OK;
}

int gmf_profile_read(gmf_handle* h, double* scattn_ms_total, int* scattn_launches) {
  GMF_REQUIRE(h && scattn_ms_total && scattn_launches, GMF_ERR_BAD_ARG, "profile_read: null pointer");
  SetDevice sd(h);
  double total = 0.0;
  for (size_t i = 0; i < h->prof_used; ++i) {
    GMF_HIP(hipEventSynchronize(h->prof_events[i].second));
    float ms = 0.f;
    GMF_HIP(hipEventElapsedTime(&ms, h->prof_events[i].first, h->prof_events[i].second));
    total += ms;
  }
  *scattn_ms_total = total;
  *scattn_launches = (int)h->prof_used;
  h->prof_used = 0;
  return GMF_OK;
}

void gmf_destroy(gmf_handle* h) {
  if (!h) return;
  for (auto& e : h->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(h->device);
  if (h->arena && !h->arena_external) (void)hipFree(h->arena);
  if (h->xs_event) (void)hipEventDestroy(h->xs_event);
  for (auto& sl : h->ptab_ring) {
    if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (sl.host) (void)hipHostFree(sl.host);
  }
  if (h->status_host) (void)hipHostFree(h->status_host);
  if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
  delete h;
}

const char* gmf_last_error_string(gmf_handle* h) { return h ? h->err.c_str() : "gmf: null handle"; }

long long gmf_workspace_bytes(gmf_handle* h) { return h ? (long long)h->arena_bytes : 0; }

// ---------------------------------------------------------------------------------------------
int gmf_pack_rows_p32(gmf_handle* h, const float* src, long long sb, long long sr, long long sk, int B, int n_rows,
                      int K, float* dst, gmf_stream_t stream) {
  GMF_REQUIRE(h && src && dst, GMF_ERR_BAD_ARG, "pack_rows_p32: null pointer");
  GMF_REQUIRE(B > 0 && n_rows > 0 && K > 0 && K % 8 == 0, GMF_ERR_UNSUPPORTED_SHAPE, "pack_rows_p32: K must be a positive multiple of 8");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_pack_p32(src, dst, B, n_rows, K, sb, sr, sk, S(stream)));
  return GMF_OK;
}

int gmf_unpack_rows_p32(gmf_handle* h, const float* src_img, int B, int n_rows, int K, float* dst, long long sb,
                        long long sr, long long sk, gmf_stream_t stream) {
  GMF_REQUIRE(h && src_img && dst, GMF_ERR_BAD_ARG, "unpack_rows_p32: null pointer");
  GMF_REQUIRE(B > 0 && n_rows > 0 && K > 0 && K % 8 == 0, GMF_ERR_UNSUPPORTED_SHAPE, "unpack_rows_p32: K must be a positive multiple of 8");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_unpack_p32(src_img, dst, B, n_rows, K, sb, sr, sk, S(stream)));
  return GMF_OK;
}

int gmf_pack_pts8(gmf_handle* h, const float* src, const float* tgt, int B, int N, float* dst, gmf_stream_t stream) {
  GMF_REQUIRE(h && src && tgt && dst, GMF_ERR_BAD_ARG, "pack_pts8: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "pack_pts8: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_pack_pts8(src, tgt, dst, B, N, S(stream)));
  return GMF_OK;
}

// ---------------------------------------------------------------------------------------------
int gmf_front_forward(gmf_handle* h, int first, const float* in, const float* wst, const float* vecs, float* f,
                      float* q, float* k, float* v, int B, int N, gmf_stream_t stream) {
  GMF_REQUIRE(h && in && wst && vecs && f && q && k && v, GMF_ERR_BAD_ARG, "front_forward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "front_forward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_front(first ? 1 : 0, in, wst, vecs, f, q, k, v, B, N, tiles_of(N), S(stream)));
  return GMF_OK;
}

int gmf_scattn_forward(gmf_handle* h, const float* q, const float* k, const float* v, const float* pts8,
                       const float* fusion2_out, const float* wst, const float* vecs, float* out, int B, int N,
                       float sigma_d, gmf_stream_t stream) {
  GMF_REQUIRE(h && q && k && v && pts8 && fusion2_out && wst && vecs && out, GMF_ERR_BAD_ARG, "scattn_forward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "scattn_forward: empty input");
  GMF_REQUIRE(sigma_d > 0.f, GMF_ERR_BAD_ARG, "scattn_forward: sigma_d must be positive");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_scattn_fp32(q, k, v, pts8, fusion2_out, wst, vecs, out, B, N, tiles_of(N), sigma_d, S(stream)));
  return GMF_OK;
}

int gmf_scattn_forward_dense(gmf_handle* h, const float* q, const float* k, const float* v, const float* attention,
                             const float* fusion2_out, const float* wst, const float* vecs, float* out, int B, int N,
                             gmf_stream_t stream) {
  GMF_REQUIRE(h && q && k && v && attention && fusion2_out && wst && vecs && out, GMF_ERR_BAD_ARG, "scattn_forward_dense: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "scattn_forward_dense: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_scattn_dense(q, k, v, attention, fusion2_out, wst, vecs, out, B, N, tiles_of(N), S(stream)));
  return GMF_OK;
}

int gmf_fusion_ctx_prepare(gmf_handle* h, int pe, const float* ctx, const float* wst, const float* vecs, float* out,
                           int B, int T, int sets, int wst_stride, int vec_stride, gmf_stream_t stream) {
  GMF_REQUIRE(h && ctx && wst && vecs && out, GMF_ERR_BAD_ARG, "fusion_ctx_prepare: null pointer");
  GMF_REQUIRE(B > 0 && T > 0 && sets > 0, GMF_ERR_UNSUPPORTED_SHAPE, "fusion_ctx_prepare: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_ctx_prep(pe != 0, ctx, wst, vecs, out, B, T, tiles_of(T), sets, wst_stride, vec_stride, S(stream)));
  return GMF_OK;
}

int gmf_fusion_attn_forward(gmf_handle* h, int pe, const float* x, const float* ctx_img, const float* wst,
                            const float* vecs, float* x1, int B, int N, int T, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && ctx_img && wst && vecs && x1, GMF_ERR_BAD_ARG, "fusion_attn_forward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "fusion_attn_forward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_fusion_attn(pe != 0, x, ctx_img, wst, vecs, x1, B, N, tiles_of(N), T, tiles_of(T), S(stream)));
  return GMF_OK;
}

int gmf_fusion_ff_forward(gmf_handle* h, const float* x1, const float* wst, const float* vecs, float* x2, int B,
                          int N, gmf_stream_t stream) {
  GMF_REQUIRE(h && x1 && wst && vecs && x2, GMF_ERR_BAD_ARG, "fusion_ff_forward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "fusion_ff_forward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_fusion_ff(x1, wst, vecs, x2, B, tiles_of(N), S(stream)));
  return GMF_OK;
}

int gmf_classifier_forward(gmf_handle* h, const float* feat_img, const float* wst, const float* vecs, float* logits,
                           float* feat_n, float* feat, int B, int N, gmf_stream_t stream) {
  GMF_REQUIRE(h && feat_img && wst && vecs && logits && feat_n, GMF_ERR_BAD_ARG, "classifier_forward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "classifier_forward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_head(feat_img, wst, vecs, logits, feat_n, feat, B, N, tiles_of(N), S(stream), h->status_dev));
  return GMF_OK;
}

// ---------------------------------------------------------------------------------------------
static int check_weights(gmf_handle* h, const gmf_encoder_weights* w) {
  GMF_REQUIRE(w, GMF_ERR_BAD_ARG, "encoder weights: null");
  GMF_REQUIRE(w->num_layers >= 0 && w->num_layers <= 64, GMF_ERR_BAD_ARG, "encoder weights: bad num_layers");
  GMF_REQUIRE(w->sigma_d > 0.f, GMF_ERR_BAD_ARG, "encoder weights: sigma_d must be positive");
  return GMF_OK;
}

// split-fp16 path: needs every split-fp16 weight image; the dense-`attention` drop-in and scattn_variant 0 run on fp32 images
static bool use_h2(const gmf_handle* h, const gmf_encoder_weights* w, bool dense) {
  return !dense && h->tune.scattn_variant >= 9 && w->front_wst_h2 && w->ctx_wst_h2 && w->attn_wst_h2 && w->ff_wst_h2 && w->tail_wst_h2;
}

static int run_scattn(gmf_handle* h, const gmf_encoder_weights* w, int l, const float* q, const float* k, const float* v,
                      const float* pts8, const float* x2, float* out, int B, int N, hipStream_t st, const float* dense_compat,
                      const gmf::CompatCache* cc);

// Runs Fusion-2 + the spatial-consistency block of layer `l` given f,q,k,v.
static int run_block_tail(gmf_handle* h, const gmf_encoder_weights* w, int l, const float* f, const float* q,
                          const float* k, const float* v, const float* pts8, const float* ctx_l, float* x1, float* x2,
                          float* out, int B, int N, int T, hipStream_t st, const float* dense_compat = nullptr,
                          const gmf::CompatCache* cc = nullptr) {
  const int tiles = tiles_of(N), tt = tiles_of(T);
  const bool h2 = use_h2(h, w, dense_compat != nullptr);
  if (h2) {
    GMF_HIP(gmf::launch_fusion_attn_h2(true, f, ctx_l, w->attn_wst_h2 + (size_t)l * w->attn_wst_stride,
                                       w->attn_vec + (size_t)l * w->attn_vec_stride, x1, B, N, tiles, T, tt, st));
    GMF_HIP(gmf::launch_fusion_ff_h2(h->tune, x1, w->ff_wst_h2 + (size_t)l * w->ff_wst_stride,
                                     w->ff_vec + (size_t)l * w->ff_vec_stride, x2, B, tiles, st,
                                     cc ? cc->part_o : nullptr, cc ? cc->max_splits : 0));   // the attention's partial buffer is free here
  } else {
    GMF_HIP(gmf::launch_fusion_attn(true, f, ctx_l, w->attn_wst + (size_t)l * w->attn_wst_stride,
                                    w->attn_vec + (size_t)l * w->attn_vec_stride, x1, B, N, tiles, T, tt, st));
    GMF_HIP(gmf::launch_fusion_ff(x1, w->ff_wst + (size_t)l * w->ff_wst_stride, w->ff_vec + (size_t)l * w->ff_vec_stride,
                                  x2, B, tiles, st));
  }
  return run_scattn(h, w, l, q, k, v, pts8, x2, out, B, N, st, dense_compat, cc);
}

// in-situ profiling: an event pair around the attention launch(es) of a layer, recorded on the caller's stream
static int prof_begin(gmf_handle* h, hipStream_t st, hipEvent_t* ev0, hipEvent_t* ev1) {
  *ev0 = *ev1 = nullptr;
  if (!h->profile) return GMF_OK;
  if (h->prof_used == h->prof_events.size()) {
    hipEvent_t a, b;
    GMF_HIP(hipEventCreate(&a));
    GMF_HIP(hipEventCreate(&b));
    h->prof_events.emplace_back(a, b);
  }
  *ev0 = h->prof_events[h->prof_used].first;
  *ev1 = h->prof_events[h->prof_used].second;
  ++h->prof_used;
  GMF_HIP(hipEventRecord(*ev0, st));
  return GMF_OK;
}

// The attention launch of layer `l` (bracketed by the in-situ profiling events when enabled).
static int run_scattn(gmf_handle* h, const gmf_encoder_weights* w, int l, const float* q, const float* k, const float* v,
                      const float* pts8, const float* x2, float* out, int B, int N, hipStream_t st, const float* dense_compat,
                      const gmf::CompatCache* cc) {
  const int tiles = tiles_of(N);
  const bool h2 = use_h2(h, w, dense_compat != nullptr);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (int rc = prof_begin(h, st, &ev0, &ev1)) return rc;
  if (dense_compat)
    GMF_HIP(gmf::launch_scattn_dense(q, k, v, dense_compat, x2, w->tail_wst + (size_t)l * w->tail_wst_stride,
                                     w->tail_vec + (size_t)l * w->tail_vec_stride, out, B, N, tiles, st));
  else if (h2)
    GMF_HIP(gmf::launch_scattn_h2(h->tune, q, k, v, pts8, x2, w->tail_wst + (size_t)l * w->tail_wst_stride,
                                  w->tail_vec + (size_t)l * w->tail_vec_stride, out, B, N, tiles, w->sigma_d, st, cc));
  else
    GMF_HIP(gmf::launch_scattn_fp32(q, k, v, pts8, x2, w->tail_wst + (size_t)l * w->tail_wst_stride,
                                    w->tail_vec + (size_t)l * w->tail_vec_stride, out, B, N, tiles, w->sigma_d, st));
  if (ev1) GMF_HIP(hipEventRecord(ev1, st));
  return GMF_OK;
}

// Host table of a ragged batch -> the next slot of the handle's pinned ring; returns max n, sum n.
// ratio < 0: the encoder's table (S, k unused).  The caller uploads it with upload_pair_table().
static int build_pair_table(gmf_handle* h, const int* n_points, int B, double ratio, int k, int* n_max, long long* n_sum, int* s_max) {
  for (int b = 0; b < B; ++b) {
    GMF_REQUIRE(n_points[b] > 0, GMF_ERR_UNSUPPORTED_SHAPE, "ragged batch: every pair needs at least one correspondence");
  }
  h->ptab_cur = h->ptab_next;
  h->ptab_next = (h->ptab_next + 1) % gmf_handle::kPtabSlots;
  gmf_handle::PtabSlot& slot = h->ptab_ring[h->ptab_cur];
  if (slot.pending) {                         // (eight uploads ago: long done unless the host runs far ahead of the device)
    GMF_HIP(hipEventSynchronize(slot.ev));
    slot.pending = false;
  }
  if (slot.cap < (size_t)B) {
    if (slot.host) GMF_HIP(hipHostFree(slot.host));
    slot.host = nullptr;
    slot.cap = 0;
    const size_t cap = ((size_t)B + 63) / 64 * 64;
    GMF_HIP(hipHostMalloc(reinterpret_cast<void**>(&slot.host), cap * sizeof(gmf::PairTab), hipHostMallocDefault));
    slot.cap = cap;
  }
  if (!slot.ev) GMF_HIP(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
  gmf::PairTab* tab = slot.host;
  long long row = 0;
  int nm = 0, sm = 0;
  for (int b = 0; b < B; ++b) {
    const int n = n_points[b];
    GMF_REQUIRE(row + n <= 0x7fffffffLL, GMF_ERR_UNSUPPORTED_SHAPE, "ragged batch: more than 2^31 rows");
    const int Sb = ratio >= 0.0 ? (int)((double)n * ratio) : 0;     // S = int(N * ratio)   PointDSC.py:244
    tab[b] = gmf::PairTab{(int)row, n, Sb, k, b, 0, 0, 0};
    row += n;
    nm = n > nm ? n : nm;
    sm = Sb > sm ? Sb : sm;
  }
  {
    // slot -> pair: the pairs longest first, each into the XCD run (B / 8 consecutive slots; the first B % 8 runs hold one more) with
    // the least n^2 so far that still has a free slot; within a run longest first (PairTab::ord, launchers.hpp)
    std::vector<int> by_len(B);
    for (int b = 0; b < B; ++b) by_len[b] = b;
    std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int c) { return n_points[a] > n_points[c]; });
    const int runs = B < 8 ? 1 : 8;
    std::vector<std::vector<int>> run(runs);
    std::vector<double> load(runs, 0.0);
    std::vector<int> cap(runs);
    for (int r = 0; r < runs; ++r) cap[r] = B / runs + (r < B % runs ? 1 : 0);
    for (int b : by_len) {
      int best = -1;
      for (int r = 0; r < runs; ++r)
        if ((int)run[r].size() < cap[r] && (best < 0 || load[r] < load[best])) best = r;
      run[best].push_back(b);
      load[best] += (double)n_points[b] * (double)n_points[b];
    }
    int slot_i = 0;
    for (int r = 0; r < runs; ++r)
      for (int b : run[r]) tab[slot_i++].ord = b;
  }
  *n_max = nm;
  *n_sum = row;
  if (s_max) *s_max = sm;
  return GMF_OK;
}

// the current slot -> device, asynchronously on the call's stream; the slot's event marks the copy
static int upload_pair_table(gmf_handle* h, gmf::PairTab* dtab, int B, hipStream_t st) {
  gmf_handle::PtabSlot& slot = h->ptab_ring[h->ptab_cur];
  GMF_HIP(hipMemcpyAsync(dtab, slot.host, (size_t)B * sizeof(gmf::PairTab), hipMemcpyHostToDevice, st));
  GMF_HIP(hipEventRecord(slot.ev, st));
  slot.pending = true;
  return GMF_OK;
}

static int encoder_forward_impl(gmf_handle* h, const gmf_encoder_weights* w, const float* corr_pos, const float* src_keypts,
                                const float* tgt_keypts, const float* p_tokens, const float* q_tokens, int B, int N, int T,
                                float* logits, float* feat_n, float* feat, gmf_stream_t stream, const int* n_points);

int gmf_encoder_forward(gmf_handle* h, const gmf_encoder_weights* w, const float* corr_pos, const float* src_keypts,
                        const float* tgt_keypts, const float* p_tokens, const float* q_tokens, int B, int N, int T,
                        float* logits, float* feat_n, float* feat, gmf_stream_t stream) {
  GMF_REQUIRE(h, GMF_ERR_BAD_ARG, "encoder_forward: null handle");
  GMF_REQUIRE(B > 0 && N > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "encoder_forward: empty input");
  return encoder_forward_impl(h, w, corr_pos, src_keypts, tgt_keypts, p_tokens, q_tokens, B, N, T, logits, feat_n, feat, stream, nullptr);
}

int gmf_encoder_forward_ragged(gmf_handle* h, const gmf_encoder_weights* w, const float* corr_pos, const float* src_keypts,
                               const float* tgt_keypts, const float* p_tokens, const float* q_tokens, const int* n_points, int B,
                               int T, float* logits, float* feat_n, float* feat, gmf_stream_t stream) {
  GMF_REQUIRE(h, GMF_ERR_BAD_ARG, "encoder_forward_ragged: null handle");
  GMF_REQUIRE(n_points, GMF_ERR_BAD_ARG, "encoder_forward_ragged: null n_points");
  GMF_REQUIRE(B > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "encoder_forward_ragged: empty input");
  return encoder_forward_impl(h, w, corr_pos, src_keypts, tgt_keypts, p_tokens, q_tokens, B, 0, T, logits, feat_n, feat, stream, n_points);
}

static int encoder_forward_impl(gmf_handle* h, const gmf_encoder_weights* w, const float* corr_pos, const float* src_keypts,
                                const float* tgt_keypts, const float* p_tokens, const float* q_tokens, int B, int N, int T,
                                float* logits, float* feat_n, float* feat, gmf_stream_t stream, const int* n_points) {
  if (int rc = check_weights(h, w)) return rc;
  GMF_REQUIRE(corr_pos && src_keypts && tgt_keypts && p_tokens && q_tokens && logits && feat_n, GMF_ERR_BAD_ARG,
              "encoder_forward: null pointer");
  SetDevice sd(h, stream);
  hipStream_t st = S(stream);
  // ragged batch: per-pair sizes from the host; every image keeps a slot of tiles(max n) tiles per pair
  const bool ragged = n_points != nullptr;
  if (ragged) {
    long long n_sum = 0;
    if (int rc = build_pair_table(h, n_points, B, -1.0, 0, &N, &n_sum, nullptr)) return rc;
  }
  const int L = w->num_layers;
  const int tiles = tiles_of(N), tt = tiles_of(T);
  const size_t act = (size_t)B * tiles * kTileFloats;
  const size_t tok = (size_t)B * tt * kTileFloats;
  // compat cache (built once per batch, streamed by all L attention launches): 4 KiB per pair of 32-row tiles
  const size_t n_tt = (size_t)B * tiles * tiles;
  const bool h2 = use_h2(h, w, false);
  // (2 KiB per tile pair in the 16-bit formats of the pipelined kernel, 4 KiB as fp32)
  const size_t c_tile_floats = (h->tune.scattn_variant == 18 && h->tune.compat_format != 0) ? 512 : 1024;
  const bool want_cache = h2 && (L > 1) && h->tune.use_cache && n_tt * c_tile_floats * 4 <= ((size_t)96 << 30);
  const size_t cache_need = want_cache ? arena_need(n_tt * c_tile_floats, 4) : 0;
  // Two launches per layer (the default path): see below.  Checked BEFORE the workspace is reserved and anything is launched
  const bool fuse = h2 && L > 0 && want_cache && h->tune.fused_linear && h->tune.scattn_variant == 18;
  GMF_REQUIRE(fuse || !ragged, GMF_ERR_UNSUPPORTED_SHAPE,
              "encoder_forward_ragged: ragged batches run on the default path only (split-fp16 weight images, at least two layers, the "
              "compat cache, \"fused_linear\" = 1, \"scattn_variant\" = 18)");
  // key-split attention for small grids (fewer than 256 workgroups of 128 queries): partial-result workspace
  // (small grids: up to 8 splits of every query block; large grids: the last partial round of workgroups is split in 2..4)
  const int kMaxSplits = ((tiles + 3) / 4) * B < 384 ? 8 : 4;
  const bool want_split = want_cache && tiles >= 8;
  const size_t split_need = want_split ? arena_need((size_t)kMaxSplits * act, 4) + arena_need((size_t)kMaxSplits * B * tiles * 64, 4) +
                                             (kMaxSplits == 8 ? arena_need((size_t)8 * act, 4) : 0) : 0;
  const size_t need = 8 * arena_need(act, 4) + arena_need((size_t)B * tiles * 32 * 8, 4) +
                      5 * arena_need(tok, 4) + arena_need((size_t)(L > 0 ? L : 1) * tok, 4) + cache_need + split_need +
                      arena_need((size_t)B, sizeof(gmf::PairTab)) + arena_need((size_t)B * tiles * 128, 4) +
                      arena_need((size_t)(L + 1) * B * gmf::kPvStatStride, 4);
  if (int rc = arena_reserve(h, need)) return rc;
  const gmf::PairTab* ptab = nullptr;
  if (ragged) {
    gmf::PairTab* dtab = arena_take<gmf::PairTab>(h, (size_t)B);
    if (int rc = upload_pair_table(h, dtab, B, st)) return rc;
    ptab = dtab;
  }
  float* featA = arena_take<float>(h, act);
  float* featB = arena_take<float>(h, act);
  float* f = arena_take<float>(h, act);
  float* q = arena_take<float>(h, act);
  float* k = arena_take<float>(h, act);
  float* v = arena_take<float>(h, act);
  float* x1 = arena_take<float>(h, act);
  float* x2 = arena_take<float>(h, act);
  float* pts8 = arena_take<float>(h, (size_t)B * tiles * 32 * 8);
  float* pimg = arena_take<float>(h, tok);
  float* qimg = arena_take<float>(h, tok);
  float* f1ctx = arena_take<float>(h, tok);
  float* x1t = arena_take<float>(h, tok);
  float* imgfeat = arena_take<float>(h, tok);
  float* ctxall = arena_take<float>(h, (size_t)(L > 0 ? L : 1) * tok);
  unsigned* v_scale = arena_take<unsigned>(h, (size_t)B * tiles * 128);    // scale words of the V and K images' e4m3 planes ("pv_fp8"): per tile [V: 64 | K: 64]
  unsigned* fstat = arena_take<unsigned>(h, (size_t)(L + 1) * B * gmf::kPvStatStride);   // "pv_fp8" guard: [layer][pair] max row |f_l|^2 (float bits), one 128-byte line per pair
  gmf::CompatCache cc{nullptr, nullptr, nullptr, nullptr, 0};
  float* c_dense = nullptr;
  if (want_cache) {
    c_dense = arena_take<float>(h, n_tt * c_tile_floats);
    cc.dense = c_dense;
  }
  float* ff_part = nullptr;
  if (want_split && kMaxSplits == 8) ff_part = arena_take<float>(h, (size_t)8 * act);   // small grids: feed-forward partials beside the attention's
  if (want_split) {
    cc.part_o = arena_take<float>(h, (size_t)kMaxSplits * act);
    cc.part_ml = arena_take<float>(h, (size_t)kMaxSplits * B * tiles * 64);
    cc.max_splits = kMaxSplits;
  }

  // throughput numerics ("precision" = 1, 2): on the two-launch path of large grids the attention multiplies one fp16
  // product and streams the compat matrix as fp16 (level 2: the layer's linear stages multiply one product as well); every
  // other path keeps the parity numerics
  cc.half = h->tune.precision >= 1 && h2 && L > 0 && want_cache && h->tune.fused_linear && h->tune.scattn_variant == 18 &&
            ((tiles + 3) / 4) * B >= 256 && !ragged;
  // the cache's element format: fp16 c in the throughput mode; else the handle's "compat_format" wherever the pipelined
  // kernel (variant 18) is the cache's only reader
  cc.fmt = cc.half ? 1 : (want_cache && h->tune.scattn_variant == 18) ? h->tune.compat_format : 0;
  cc.ptab = ptab;
  if (ragged) {
    int nmin = n_points[0];
    for (int b = 1; b < B; ++b) nmin = n_points[b] < nmin ? n_points[b] : nmin;
    cc.min_tiles = tiles_of(nmin);
  }
  // (the key-point packing also clears the "pv_fp8" guard's statistics - a superset of the forwards that read them)
  const bool may_guard = fuse && h->tune.pv_fp8 == 1 && w->pv_guard;
  unsigned* const zero_words = may_guard ? fstat : nullptr;
  const int n_zero = may_guard ? (L + 1) * B * gmf::kPvStatStride : 0;
  // [r5] small grids: the prologue is two independent chains of few-workgroup kernels - image side (Fusion-1 context, cross-attention,
  // feed-forward) and point side (key points, compat cache, layer 0 + first PointCN).  Three launches carry one link of each
  // (k_pro_*, encoder_h2.hip): the point side runs under the image side instead of behind it.  Same bodies: bit-identical.
  const bool pro = h->tune.small_prologue && fuse && w->f1_ctx_wst_h2 && w->f1_attn_wst_h2 && w->f1_ff_wst_h2 && cc.fmt == 0 &&
                   ((tiles + 3) / 4) * B < 256;
  if (pro) {
    GMF_HIP(gmf::launch_pro_ctx_pts(p_tokens, w->f1_ctx_wst_h2, w->f1_ctx_vec, f1ctx, B, T, tt, src_keypts, tgt_keypts, pts8, N, st, ptab,
                                    zero_words, n_zero));
    GMF_HIP(gmf::launch_pro_fattn_compat(q_tokens, f1ctx, w->f1_attn_wst_h2, w->f1_attn_vec, x1t, B, T, tt, pts8, c_dense, N, tiles,
                                         w->sigma_d, st, ptab));
    gmf::PvGuard g0;
    if (may_guard) g0.stat_next = fstat;         // (small grids never run the throughput numerics: may_guard is `guarded` below)
    int hs = 1;
    GMF_HIP(gmf::launch_pro_ff_front(h->tune, x1t, w->f1_ff_wst_h2, w->f1_ff_vec, imgfeat, B, tt, tt <= tiles ? cc.part_o : nullptr,
                                     tt <= tiles ? cc.max_splits : 0, corr_pos, w->front_wst_h2, w->front_vec, f, q, k, v, N, tiles, st,
                                     ptab, g0, &hs));
    if (hs > 1) GMF_HIP(gmf::launch_ff_reduce_h2(cc.part_o, x1t, w->f1_ff_vec, imgfeat, B, tt, hs, st));
    GMF_HIP(gmf::launch_ctx_prep_h2(true, imgfeat, w->ctx_wst_h2, w->ctx_vec, ctxall, B, T, tt, L, w->ctx_wst_stride, w->ctx_vec_stride, st));
  } else {
    // Fusion-1: image_feat = FusionLayer(p_tok (context), queries = q_tok), pe = False (PointDSC.py:137)
    if (h2 && w->f1_ctx_wst_h2 && w->f1_attn_wst_h2 && w->f1_ff_wst_h2) {
      // (the two token tensors are read row-major: no packing launches in front of these few-workgroup kernels)
      GMF_HIP(gmf::launch_ctx_prep_h2(false, p_tokens, w->f1_ctx_wst_h2, w->f1_ctx_vec, f1ctx, B, T, tt, 1, 0, 0, st, true));
      GMF_HIP(gmf::launch_fusion_attn_h2(false, q_tokens, f1ctx, w->f1_attn_wst_h2, w->f1_attn_vec, x1t, B, T, tt, T, tt, st, true));
      GMF_HIP(gmf::launch_fusion_ff_h2(h->tune, x1t, w->f1_ff_wst_h2, w->f1_ff_vec, imgfeat, B, tt, st,
                                       tt <= tiles ? cc.part_o : nullptr, tt <= tiles ? cc.max_splits : 0));
    } else {
      GMF_HIP(gmf::launch_pack_p32(p_tokens, pimg, B, T, kC, (long)T * kC, kC, 1, st));
      GMF_HIP(gmf::launch_pack_p32(q_tokens, qimg, B, T, kC, (long)T * kC, kC, 1, st));
      GMF_HIP(gmf::launch_ctx_prep(false, pimg, w->f1_ctx_wst, w->f1_ctx_vec, f1ctx, B, T, tt, 1, 0, 0, st));
      GMF_HIP(gmf::launch_fusion_attn(false, qimg, f1ctx, w->f1_attn_wst, w->f1_attn_vec, x1t, B, T, tt, T, tt, st));
      GMF_HIP(gmf::launch_fusion_ff(x1t, w->f1_ff_wst, w->f1_ff_vec, imgfeat, B, tt, st));
    }
    // context side of all L Fusion-2 layers in one launch
    if (L > 0) {
      if (h2) GMF_HIP(gmf::launch_ctx_prep_h2(true, imgfeat, w->ctx_wst_h2, w->ctx_vec, ctxall, B, T, tt, L, w->ctx_wst_stride,
                                              w->ctx_vec_stride, st));
      else GMF_HIP(gmf::launch_ctx_prep(true, imgfeat, w->ctx_wst, w->ctx_vec, ctxall, B, T, tt, L, w->ctx_wst_stride,
                                        w->ctx_vec_stride, st));
    }
    GMF_HIP(gmf::launch_pack_pts8(src_keypts, tgt_keypts, pts8, B, N, st, ptab, zero_words, n_zero));
    if (want_cache) GMF_HIP(gmf::launch_compat_build(pts8, c_dense, B, N, tiles, w->sigma_d, cc.fmt, st, ptab));
  }

  float* cur = featA;
  float* nxt = featB;
  // Two launches per layer (default on the split-fp16 path with the cached, pipelined attention kernel): the layer's PointCN
  // runs in the PREVIOUS layer's attention epilogue (layer 0: a small f-only kernel), so a layer is
  //   k_linear_h2 : f -> Q', K, V (split-fp16 images) and x2 = Fusion-2(f)        (small grids: the three split-capable kernels)
  //   k_scattn_h2p: Q', K, V, c, x2 -> f_{l+1} = ReLU(PointCN_{l+1}(fc_message(attention) + x2))   (last layer: the features)
  if (fuse) {
    const int Wg = ((tiles + 3) / 4) * B;
    // small grids: when both the attention and the feed-forward are split anyway, their workgroups share launches
    // ([r5] ragged batches too: the split is planned on the smallest pair's tiles, the role kernels read the pair table)
    int small_nf = 1, small_ks = 1;
    gmf::plan_attn_split(h->tune, Wg, ragged ? cc.min_tiles : tiles, cc.part_o ? cc.max_splits : 0, &small_nf, &small_ks);
    const int ff_hs = cc.part_o ? gmf::plan_ff_split(h->tune, Wg, cc.max_splits) : 1;
    const bool small3_ok = Wg < 256 && h->tune.small_roles && small_nf == 0 && small_ks > 1 && ff_hs > 1 && ff_part;
    // below 256 row blocks key / hidden / output splits fill the chip better; a ragged batch takes either the three-launch form of
    // small grids or the two-launch form (the one-kernel-per-stage form in between has no pair table)
    const bool one_kernel = Wg >= 256 || (ragged && !small3_ok);
    const bool small3 = !one_kernel && small3_ok;
    // parity arithmetic: V with e4m3 cross planes for the pv_fp8 form of the attention body (scattn_h2p_body<3, *, 4, true>), which every
    // attention kernel of this path instantiates - large grids, split tails and the small-grid role kernels alike
    // (a weights block without thresholds - filled in by hand, pv_guard = NULL - gets the three-product form under the guarded default,
    // never the unguarded one)
    cc.v_scale = ((h->tune.pv_fp8 == 2 || (h->tune.pv_fp8 == 1 && w->pv_guard)) && !cc.half) ? v_scale : nullptr;
    // [r5] "pv_fp8" = 1: guarded per pair and layer on the device (PvGuard).  The statistics start at zero (k_pack_pts8); f_0's is raised by the
    // front kernel, f_{l+1}'s by the attention epilogue / merge kernels of layer l - always before the kernels that read it
    const bool guarded = cc.v_scale && h->tune.pv_fp8 == 1 && w->pv_guard;       // (implies may_guard: launch_pack_pts8 cleared fstat)
    if (!pro) {                                   // (small grids: the prologue's third launch ran it)
      gmf::PvGuard g0;
      if (guarded) g0.stat_next = fstat;
      GMF_HIP(gmf::launch_front_h2(h->tune, 3, corr_pos, w->front_wst_h2, w->front_vec, f, q, k, v, B, N, tiles, st, ptab, nullptr, g0));
    }
    for (int l = 0; l < L; ++l) {
      cc.guard = gmf::PvGuard{};
      if (guarded) cc.guard = gmf::PvGuard{fstat + (size_t)l * B * gmf::kPvStatStride, w->pv_guard + l, fstat + (size_t)(l + 1) * B * gmf::kPvStatStride};
      const float* fw = w->front_wst_h2 + (size_t)l * w->front_wst_stride;
      const float* fv = w->front_vec + (size_t)l * w->front_vec_stride;
      const float* aw = w->attn_wst_h2 + (size_t)l * w->attn_wst_stride;
      const float* av = w->attn_vec + (size_t)l * w->attn_vec_stride;
      const float* ffw = w->ff_wst_h2 + (size_t)l * w->ff_wst_stride;
      const float* ffv = w->ff_vec + (size_t)l * w->ff_vec_stride;
      const float* ctx_l = ctxall + (size_t)l * tok;
      if (one_kernel) {
        // [r4] Q' in the attention kernel's prologue (parity arithmetic of the pipelined kernel; not with the linear kernel as two
        // roles, whose Q'/K/V role writes the image): k_linear_h2 then projects K and V only
        const int Wl = ((tiles + 3) / 4) * B;
        const bool roles = h->tune.mid_grid_roles > 0 && Wl < h->tune.mid_grid_roles;     // ([r5] ragged batches too: the role kernel reads the pair table)
        const bool qproj = h->tune.q_in_attention && !cc.half && !roles;
        cc.qf_img = qproj ? f : nullptr;
        cc.qw_wst = qproj ? fw + 4 * kTileFloats : nullptr;
        cc.qw_bias = qproj ? fv + kC : nullptr;
        GMF_HIP(gmf::launch_linear_h2(h->tune, f, fw, fv, ctx_l, aw, av, ffw, ffv, qproj ? nullptr : q, k, v, x2, B, N, tiles, T, tt, st,
                                      cc.half && h->tune.precision == 2, ptab, cc.v_scale, cc.guard));
      } else if (small3) {
        // three launches: {Q' | K | V | cross-attention} -> {key-split attention | hidden-split feed-forward} -> merge
        GMF_HIP(gmf::launch_small_front_fattn(f, fw, fv, ctx_l, aw, av, q, k, v, x1, B, N, tiles, T, tt, st, cc.v_scale, cc.guard,
                                              h->tune.small_fattn_tile, ptab));
        const bool last3 = (l + 1 == L);
        cc.tail_wst_h2 = w->tail_wst_h2 + (size_t)l * w->tail_wst_stride;
        cc.next_wst_h2 = last3 ? nullptr : w->front_wst_h2 + (size_t)(l + 1) * w->front_wst_stride;
        cc.next_bias = last3 ? nullptr : w->front_vec + (size_t)(l + 1) * w->front_vec_stride;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (int rc = prof_begin(h, st, &e0, &e1)) return rc;
        GMF_HIP(gmf::launch_small_attn_ff_merge(q, k, v, x1, ffw, ffv, ff_part, ff_hs, w->tail_vec + (size_t)l * w->tail_vec_stride,
                                                last3 ? cur : f, B, N, tiles, small_ks, st, &cc, h->tune.small_merge_tile));
        if (e1) GMF_HIP(hipEventRecord(e1, st));
        continue;
      } else {
        // (projecting Q'/K/V on a side stream beside the Fusion-2 kernels, forked and joined with events, was measured at
        // B = 1: 1.62 vs 1.58 ms at N = 5000, 1.08 vs 1.00 ms at N = 1000 - the event round trips cost more than the overlap gives)
        GMF_HIP(gmf::launch_front_h2(h->tune, 2, f, fw, fv, f, q, k, v, B, N, tiles, st, nullptr, cc.v_scale, cc.guard));
        GMF_HIP(gmf::launch_fusion_attn_h2(true, f, ctx_l, aw, av, x1, B, N, tiles, T, tt, st));
        GMF_HIP(gmf::launch_fusion_ff_h2(h->tune, x1, ffw, ffv, x2, B, tiles, st, cc.part_o, cc.max_splits));
      }
      const bool last = (l + 1 == L);
      cc.tail_wst_h2 = w->tail_wst_h2 + (size_t)l * w->tail_wst_stride;
      cc.next_wst_h2 = last ? nullptr : w->front_wst_h2 + (size_t)(l + 1) * w->front_wst_stride;
      cc.next_bias = last ? nullptr : w->front_vec + (size_t)(l + 1) * w->front_vec_stride;
      if (int rc = run_scattn(h, w, l, q, k, v, pts8, x2, last ? cur : f, B, N, st, nullptr, &cc)) return rc;
    }
    GMF_HIP(gmf::launch_head(cur, w->head_wst, w->head_vec, logits, feat_n, feat, B, N, tiles, st, h->status_dev, ptab,
                             guarded ? fstat : nullptr, guarded ? w->pv_guard : nullptr, L));
    return GMF_OK;
  }
  if (L == 0) {
    // degenerate: features are layer0(corr_pos) only
    GMF_HIP(gmf::launch_front(1, corr_pos, w->front_wst, w->front_vec, cur, q, k, v, B, N, tiles, st));
  }
  for (int l = 0; l < L; ++l) {
    const float* in = (l == 0) ? corr_pos : cur;
    if (h2) GMF_HIP(gmf::launch_front_h2(h->tune, l == 0 ? 1 : 0, in, w->front_wst_h2 + (size_t)l * w->front_wst_stride,
                                         w->front_vec + (size_t)l * w->front_vec_stride, f, q, k, v, B, N, tiles, st));
    else GMF_HIP(gmf::launch_front(l == 0 ? 1 : 0, in, w->front_wst + (size_t)l * w->front_wst_stride,
                                   w->front_vec + (size_t)l * w->front_vec_stride, f, q, k, v, B, N, tiles, st));
    cc.tail_wst_h2 = w->tail_wst_h2 ? w->tail_wst_h2 + (size_t)l * w->tail_wst_stride : nullptr;
    if (int rc = run_block_tail(h, w, l, f, q, k, v, pts8, ctxall + (size_t)l * tok, x1, x2, nxt, B, N, T, st, nullptr,
                                &cc)) return rc;
    float* t = cur; cur = nxt; nxt = t;
  }
  GMF_HIP(gmf::launch_head(cur, w->head_wst, w->head_vec, logits, feat_n, feat, B, N, tiles, st, h->status_dev));
  return GMF_OK;
}

int gmf_nonlocal_block_forward(gmf_handle* h, const gmf_encoder_weights* w, int layer, int apply_pointcn,
                               const float* feat_img, const float* pts8, const float* attention,
                               const float* image_feat_img, float* out_img, int B, int N, int T, gmf_stream_t stream) {
  GMF_REQUIRE(h, GMF_ERR_BAD_ARG, "nonlocal_block_forward: null handle");
  if (int rc = check_weights(h, w)) return rc;
  GMF_REQUIRE(feat_img && image_feat_img && out_img, GMF_ERR_BAD_ARG, "nonlocal_block_forward: null pointer");
  GMF_REQUIRE((pts8 != nullptr) != (attention != nullptr), GMF_ERR_BAD_ARG, "nonlocal_block_forward: pass exactly one of pts8 / attention");
  GMF_REQUIRE(layer >= 0 && layer < w->num_layers, GMF_ERR_BAD_ARG, "nonlocal_block_forward: layer out of range");
  GMF_REQUIRE(B > 0 && N > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "nonlocal_block_forward: empty input");
  GMF_REQUIRE(apply_pointcn == 0 || apply_pointcn == 1, GMF_ERR_BAD_ARG, "nonlocal_block_forward: bad flag");
  SetDevice sd(h, stream);
  hipStream_t st = S(stream);
  const int tiles = tiles_of(N), tt = tiles_of(T);
  const size_t act = (size_t)B * tiles * kTileFloats;
  const size_t tok = (size_t)B * tt * kTileFloats;
  if (int rc = arena_reserve(h, 6 * arena_need(act, 4) + arena_need(tok, 4))) return rc;
  float* f = arena_take<float>(h, act);
  float* q = arena_take<float>(h, act);
  float* k = arena_take<float>(h, act);
  float* v = arena_take<float>(h, act);
  float* x1 = arena_take<float>(h, act);
  float* x2 = arena_take<float>(h, act);
  float* ctx = arena_take<float>(h, tok);
  // apply_pointcn = 0: the caller's feat is already the block input (NonLocalBlock.forward, PointDSC.py:40-45)
  if (use_h2(h, w, attention != nullptr)) {   // (the dense-compat kernel consumes fp32 images: fp32 path)
    GMF_HIP(gmf::launch_front_h2(h->tune, apply_pointcn ? 0 : 2, feat_img, w->front_wst_h2 + (size_t)layer * w->front_wst_stride,
                                 w->front_vec + (size_t)layer * w->front_vec_stride, f, q, k, v, B, N, tiles, st));
    GMF_HIP(gmf::launch_ctx_prep_h2(true, image_feat_img, w->ctx_wst_h2 + (size_t)layer * w->ctx_wst_stride,
                                    w->ctx_vec + (size_t)layer * w->ctx_vec_stride, ctx, B, T, tt, 1, 0, 0, st));
  } else {
    GMF_HIP(gmf::launch_front(apply_pointcn ? 0 : 2, feat_img, w->front_wst + (size_t)layer * w->front_wst_stride,
                              w->front_vec + (size_t)layer * w->front_vec_stride, f, q, k, v, B, N, tiles, st));
    GMF_HIP(gmf::launch_ctx_prep(true, image_feat_img, w->ctx_wst + (size_t)layer * w->ctx_wst_stride,
                                 w->ctx_vec + (size_t)layer * w->ctx_vec_stride, ctx, B, T, tt, 1, 0, 0, st));
  }
  gmf::CompatCache cc{nullptr, w->tail_wst_h2 ? w->tail_wst_h2 + (size_t)layer * w->tail_wst_stride : nullptr, nullptr, nullptr, 0};
  return run_block_tail(h, w, layer, f, q, k, v, pts8, ctx, x1, x2, out_img, B, N, T, st, attention, &cc);
}

int gmf_fusion_layer_forward(gmf_handle* h, int pe, int latent_dim, int d_head, const float* ctx_wst, const float* ctx_vec,
                             const float* attn_wst, const float* attn_vec, const float* ff_wst, const float* ff_vec,
                             const float* data, const float* queries, long long q_sb, long long q_sr, long long q_sk,
                             float* out, long long o_sb, long long o_sr, long long o_sk, int B, int N, int T,
                             gmf_stream_t stream, const float* ff_wst_h2, const float* ctx_wst_h2, const float* attn_wst_h2) {
  GMF_REQUIRE(h && ctx_wst && ctx_vec && attn_wst && attn_vec && ff_wst && ff_vec && data && queries && out,
              GMF_ERR_BAD_ARG, "fusion_layer_forward: null pointer");
  GMF_REQUIRE((ctx_wst_h2 == nullptr) == (attn_wst_h2 == nullptr), GMF_ERR_BAD_ARG,
              "fusion_layer_forward: ctx_wst_h2 and attn_wst_h2 go together (the attention reads the context image the "
              "context kernel of the same kind writes)");
  GMF_REQUIRE(B > 0 && N > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "fusion_layer_forward: empty input");
  const bool narrow = (latent_dim == 128 && d_head == 64), wide = (latent_dim == 256 && d_head == 128);
  GMF_REQUIRE(narrow || wide, GMF_ERR_UNSUPPORTED_SHAPE,
              "fusion_layer_forward: kernels exist for (latent_dim, d_head) = (128, 64) and (256, 128), context dim 128");
  SetDevice sd(h, stream);
  hipStream_t st = S(stream);
  const int tiles = tiles_of(N), tt = tiles_of(T);
  const size_t act = (size_t)B * tiles * 32 * latent_dim;
  const size_t tok = (size_t)B * tt * kTileFloats;
  const size_t ctxsz = wide ? 2 * tok : tok;
  // small grids of the wide layer: the feed-forward's hidden chunks are split over hs workgroups per row block
  const int ff_hs_w = (wide && ff_wst_h2 && h->tune.ff_split != 1) ?
                          (h->tune.ff_split > 1 ? h->tune.ff_split : gmf::plan_ff_split_w(((tiles + 3) / 4) * B)) : 1;
  if (int rc = arena_reserve(h, 3 * arena_need(act, 4) + arena_need(tok, 4) + arena_need(ctxsz, 4) +
                                    (ff_hs_w > 1 ? arena_need((size_t)ff_hs_w * act, 4) : 0))) return rc;
  float* xin = arena_take<float>(h, act);
  float* x1 = arena_take<float>(h, act);
  float* x2 = arena_take<float>(h, act);
  float* cimg = arena_take<float>(h, tok);
  float* ctx = arena_take<float>(h, ctxsz);
  float* ff_part_w = ff_hs_w > 1 ? arena_take<float>(h, (size_t)ff_hs_w * act) : nullptr;
  // the split-fp16 kernels of the 256-wide layer read the context tokens row-major and (hidden-split feed-forward) write the
  // output through its strides: on the small grids this layer usually runs on, a packing pass is a launch like any other
  const bool wide_direct = wide && attn_wst_h2 != nullptr;
  if (!wide_direct) GMF_HIP(gmf::launch_pack_p32(data, cimg, B, T, kC, (long)T * kC, kC, 1, st));
  GMF_HIP(gmf::launch_pack_p32(queries, xin, B, N, latent_dim, q_sb, q_sr, q_sk, st));
  if (narrow) {
    if (attn_wst_h2) {
      GMF_HIP(gmf::launch_ctx_prep_h2(pe != 0, cimg, ctx_wst_h2, ctx_vec, ctx, B, T, tt, 1, 0, 0, st));
      GMF_HIP(gmf::launch_fusion_attn_h2(pe != 0, xin, ctx, attn_wst_h2, attn_vec, x1, B, N, tiles, T, tt, st));
    } else {
      GMF_HIP(gmf::launch_ctx_prep(pe != 0, cimg, ctx_wst, ctx_vec, ctx, B, T, tt, 1, 0, 0, st));
      GMF_HIP(gmf::launch_fusion_attn(pe != 0, xin, ctx, attn_wst, attn_vec, x1, B, N, tiles, T, tt, st));
    }
    if (ff_wst_h2) GMF_HIP(gmf::launch_fusion_ff_h2(h->tune, x1, ff_wst_h2, ff_vec, x2, B, tiles, st));
    else GMF_HIP(gmf::launch_fusion_ff(x1, ff_wst, ff_vec, x2, B, tiles, st));
  } else {
    if (attn_wst_h2) {
      GMF_HIP(gmf::launch_ctx_prep_w_h2(pe != 0, data, ctx_wst_h2, ctx_vec, ctx, B, T, tt, st, true));
      GMF_HIP(gmf::launch_fusion_attn_w_h2(pe != 0, xin, ctx, attn_wst_h2, attn_vec, x1, B, N, tiles, T, tt, st, h->tune.wide_attn_tile));
    } else {
      GMF_HIP(gmf::launch_ctx_prep_w(pe != 0, cimg, ctx_wst, ctx_vec, ctx, B, T, tt, st));
      GMF_HIP(gmf::launch_fusion_attn_w(pe != 0, xin, ctx, attn_wst, attn_vec, x1, B, N, tiles, T, tt, st));
    }
    if (ff_wst_h2 && ff_hs_w > 1) {                 // the partials' reduction writes the caller's tensor
      GMF_HIP(gmf::launch_fusion_ff_w_h2(x1, ff_wst_h2, ff_vec, x2, B, tiles, st, ff_part_w, ff_hs_w, out, o_sb, o_sr, o_sk, N, h->status_dev));
      return GMF_OK;
    }
    if (ff_wst_h2) GMF_HIP(gmf::launch_fusion_ff_w_h2(x1, ff_wst_h2, ff_vec, x2, B, tiles, st, ff_part_w, ff_hs_w));
    else GMF_HIP(gmf::launch_fusion_ff_w(x1, ff_wst, ff_vec, x2, B, tiles, st));
  }
  GMF_HIP(gmf::launch_unpack_p32(x2, out, B, N, latent_dim, o_sb, o_sr, o_sk, st, h->status_dev));
  return GMF_OK;
}

// ---------------------------------------------------------------------------------------------
int gmf_pick_seeds(gmf_handle* h, const float* src_keypts, const float* scores, int B, int N, float nms_radius,
                   int use_nms, int num_seeds, int* seeds_out, gmf_stream_t stream) {
  GMF_REQUIRE(h && src_keypts && scores && seeds_out, GMF_ERR_BAD_ARG, "pick_seeds: null pointer");
  GMF_REQUIRE(B > 0 && N > 0 && num_seeds > 0 && num_seeds <= N, GMF_ERR_UNSUPPORTED_SHAPE, "pick_seeds: need 0 < num_seeds <= N");
  GMF_REQUIRE((size_t)num_seeds * 8 <= 156 * 1024, GMF_ERR_UNSUPPORTED_SHAPE, "pick_seeds: more than 19968 seeds (the winners' list lives in the LDS)");
  SetDevice sd(h, stream);
  hipStream_t st = S(stream);
  const float* keys = scores;
  if (use_nms) {
    const size_t n_scr = gmf::nms_scratch_floats(B, N);
    if (int rc = arena_reserve(h, arena_need((size_t)B * N, 4) + arena_need(n_scr, 4))) return rc;
    float* kbuf = arena_take<float>(h, (size_t)B * N);
    float* scr = arena_take<float>(h, n_scr);
    GMF_HIP(gmf::launch_nms_keys(h->tune, src_keypts, scores, kbuf, B, N, nms_radius, st, scr));
    keys = kbuf;
  }
  GMF_HIP(gmf::launch_sort_topk(h->tune, keys, seeds_out, B, N, num_seeds, st));
  return GMF_OK;
}

static int pose_head_impl(gmf_handle* h, const gmf_pose_params* p, const float* feat_n, const float* src_keypts,
                          const float* tgt_keypts, const float* logits, const int* seeds_in, int B, int N, float* final_trans,
                          float* final_labels, int* seeds_out, int* knn_out, float* seed_trans, float* fitness,
                          gmf_stream_t stream, const int* n_points, double ratio);

int gmf_pose_head(gmf_handle* h, const gmf_pose_params* p, const float* feat_n, const float* src_keypts,
                  const float* tgt_keypts, const float* logits, const int* seeds_in, int B, int N, float* final_trans,
                  float* final_labels, int* seeds_out, int* knn_out, float* seed_trans, float* fitness,
                  gmf_stream_t stream) {
  GMF_REQUIRE(h && p, GMF_ERR_BAD_ARG, "pose_head: null pointer");
  GMF_REQUIRE(B > 0 && N > 1, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head: need N > 1");
  return pose_head_impl(h, p, feat_n, src_keypts, tgt_keypts, logits, seeds_in, B, N, final_trans, final_labels, seeds_out, knn_out,
                        seed_trans, fitness, stream, nullptr, 0.0);
}

int gmf_pose_head_ragged(gmf_handle* h, const gmf_pose_params* p, double ratio, const float* feat_n, const float* src_keypts,
                         const float* tgt_keypts, const float* logits, const int* n_points, int B, float* final_trans,
                         float* final_labels, int* seeds_out, int* knn_out, float* seed_trans, float* fitness,
                         gmf_stream_t stream) {
  GMF_REQUIRE(h && p && n_points, GMF_ERR_BAD_ARG, "pose_head_ragged: null pointer");
  GMF_REQUIRE(B > 0, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head_ragged: empty batch");
  GMF_REQUIRE(ratio > 0.0 && ratio <= 1.0, GMF_ERR_BAD_ARG, "pose_head_ragged: ratio must be in (0, 1]");
  GMF_REQUIRE(logits, GMF_ERR_BAD_ARG, "pose_head_ragged: the seeds come from the logits (no caller-provided seeds)");
  return pose_head_impl(h, p, feat_n, src_keypts, tgt_keypts, logits, nullptr, B, 0, final_trans, final_labels, seeds_out, knn_out,
                        seed_trans, fitness, stream, n_points, ratio);
}

static int pose_head_impl(gmf_handle* h, const gmf_pose_params* p, const float* feat_n, const float* src_keypts,
                          const float* tgt_keypts, const float* logits, const int* seeds_in, int B, int N, float* final_trans,
                          float* final_labels, int* seeds_out, int* knn_out, float* seed_trans, float* fitness,
                          gmf_stream_t stream, const int* n_points, double ratio) {
  GMF_REQUIRE(feat_n && src_keypts && tgt_keypts && final_trans && final_labels, GMF_ERR_BAD_ARG, "pose_head: null pointer");
  GMF_REQUIRE(seeds_in || logits, GMF_ERR_BAD_ARG, "pose_head: need logits or seeds_in");
  const bool ragged = n_points != nullptr;
  int Sn = p->num_seeds;
  const int k = p->k, iters = p->num_iterations;
  long long n_sum = (long long)B * N;
  SetDevice sd(h, stream);
  if (ragged) {
    // per-pair sizes from the host: S = int(n * ratio) seeds each; N, Sn below are the largest pair's (grids, buffer strides)
    if (int rc = build_pair_table(h, n_points, B, ratio, k, &N, &n_sum, &Sn)) return rc;
    for (int b = 0; b < B; ++b) {
      GMF_REQUIRE(n_points[b] > k, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head_ragged: every pair needs more than k correspondences (k = min(k, N - 1) "
                                                               "per pair is not supported in a ragged batch: run such a pair on its own)");
      GMF_REQUIRE((int)((double)n_points[b] * ratio) >= 1, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head_ragged: a pair would have no seed (int(n * ratio) = 0)");
    }
  }
  GMF_REQUIRE(B > 0 && N > 1, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head: need N > 1");
  GMF_REQUIRE(Sn > 0 && Sn <= N, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head: need 0 < num_seeds <= N");
  GMF_REQUIRE(k > 0 && k <= 64 && k <= N - 1, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head: need 0 < k <= min(64, N-1)");
  GMF_REQUIRE(iters > 0 && iters <= 64, GMF_ERR_BAD_ARG, "pose_head: bad num_iterations");
  // [r4] no limit on N (the reference has none: PointDSC.py:268-286, common.py:53-75; evaluation/test_3DMatch.py:143 feeds num_node = 'all'):
  // above 16 384 rows the seed selection reads its keys from global memory and the kNN selection streams the distance rows
  GMF_REQUIRE(seeds_in || (size_t)Sn * 8 <= 156 * 1024, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head: more than 19968 seeds per pair (the winners' list lives in the LDS)");
  GMF_REQUIRE((p->sigma > 0.f || h->sigma_dev) && p->sigma_d > 0.f, GMF_ERR_BAD_ARG, "pose_head: sigma, sigma_d must be positive");
  hipStream_t st = S(stream);
  const size_t BS = (size_t)B * Sn;
  const size_t need = arena_need((size_t)B * N, 4) + arena_need(BS, 4) + arena_need(BS * k, 4) +
                      arena_need(BS * iters * k, 4) + arena_need(BS * iters, 1) + arena_need(BS * 16, 4) +
                      arena_need(BS, 4) + arena_need(BS, 4) + arena_need(B, 4) + arena_need(BS * 15, 8) + arena_need(1, 4) +
                      arena_need((size_t)B * tiles_of(N) * kTileFloats, 4) + arena_need(BS * tiles_of(N) * 32, 4) +
                      arena_need((size_t)B, sizeof(gmf::PairTab));
  if (int rc = arena_reserve(h, need)) return rc;
  float* fimg = arena_take<float>(h, (size_t)B * tiles_of(N) * kTileFloats);
  float* dmat = arena_take<float>(h, BS * tiles_of(N) * 32);       // distance rows of the seeds, padded to whole tiles
  float* keys = arena_take<float>(h, (size_t)B * N);
  int* seeds = arena_take<int>(h, BS);
  int* knn = arena_take<int>(h, BS * k);
  float* snaps = arena_take<float>(h, BS * iters * k);
  unsigned char* conv = arena_take<unsigned char>(h, BS * iters);
  float* sT = arena_take<float>(h, BS * 16);
  int* counts = arena_take<int>(h, BS);
  float* fit = arena_take<float>(h, BS);
  int* best = arena_take<int>(h, B);
  double* hsum = arena_take<double>(h, BS * 15);
  int* stop_it = arena_take<int>(h, 1);
  const gmf::PairTab* ptab = nullptr;
  if (ragged) {
    gmf::PairTab* dtab = arena_take<gmf::PairTab>(h, (size_t)B);
    if (int rc = upload_pair_table(h, dtab, B, st)) return rc;
    ptab = dtab;
    // per-seed outputs are [B, S_max, ...] slots: the slots behind a pair's own seeds are defined (zero), not left-over memory
    if (seeds_out) GMF_HIP(hipMemsetAsync(seeds_out, 0, BS * sizeof(int), st));
    if (knn_out) GMF_HIP(hipMemsetAsync(knn_out, 0, BS * k * sizeof(int), st));
    if (seed_trans) GMF_HIP(hipMemsetAsync(seed_trans, 0, BS * 16 * sizeof(float), st));
    if (fitness) GMF_HIP(hipMemsetAsync(fitness, 0, BS * sizeof(float), st));
  }
  if (seeds_out) seeds = seeds_out;
  if (knn_out) knn = knn_out;
  if (seed_trans) sT = seed_trans;
  if (fitness) fit = fitness;

  const int* seeds_use = seeds_in;
  // the unit features as the split-fp16 image of k_seed_dist; [r5] the same launch hands the NMS its keys as a copy of the scores
  // (the all-pairs form with candidate splits starts from one: a hipMemcpyAsync of its own was 4 us + a kernel boundary at B = 1)
  const bool preset = !seeds_in && p->use_nms;
  GMF_HIP(gmf::launch_pack_rows_h2(feat_n, fimg, B, N, st, ptab, preset ? logits : nullptr, preset ? keys : nullptr,
                                   preset ? (ptab ? (long)n_sum : (long)B * N) : 0));
  if (!seeds_in) {
    const float* kk = logits;
    if (p->use_nms) {
      // dmat is free until k_seed_dist: it doubles as the grid-binning scratch of the NMS when it is large enough
      float* scr = (BS * N >= gmf::nms_scratch_floats(B, N)) ? dmat : nullptr;
      GMF_HIP(gmf::launch_nms_keys(h->tune, src_keypts, logits, keys, B, N, p->nms_radius, st, scr, ptab, (long)n_sum, true));
      kk = keys;
    }
    GMF_HIP(gmf::launch_sort_topk(h->tune, kk, seeds, B, N, Sn, st, ptab));
    seeds_use = seeds;
  } else if (seeds_out) {
    GMF_HIP(hipMemcpyAsync(seeds_out, seeds_in, BS * sizeof(int), hipMemcpyDeviceToDevice, st));
  }
  // feature-space distances of the seed rows by MFMA (k_seed_dist), then per-seed top-(k+1) selection
  // (the fused form - distances computed twice and never written - was built in round 4, is exact, and is not faster:
  // tools/ubench/archive/seed_knn_fused_r04.hip)
  GMF_HIP(gmf::launch_seed_dist(fimg, seeds_use, dmat, B, N, Sn, st, ptab));
  GMF_HIP(gmf::launch_knn_seeds(feat_n, seeds_use, dmat, knn, B, N, Sn, k, st, ptab));
  GMF_HIP(gmf::launch_seed_power(feat_n, src_keypts, tgt_keypts, knn, snaps, conv, hsum, B, N, Sn, k, iters, p->sigma, p->sigma_d, st, ptab, h->sigma_dev));
  GMF_HIP(gmf::launch_seed_kabsch(src_keypts, tgt_keypts, knn, snaps, conv, sT, B, N, Sn, k, iters, hsum, stop_it, st, ptab));
  GMF_HIP(gmf::launch_score_hyp(src_keypts, tgt_keypts, sT, counts, B, N, Sn, p->inlier_threshold, st, ptab));
  GMF_HIP(gmf::launch_finalize_pose(src_keypts, tgt_keypts, sT, counts, fit, final_trans, final_labels, best, B, N, Sn,
                                    p->inlier_threshold, p->refine_threshold, p->refine_iters, st, ptab));
  return GMF_OK;
}

int gmf_knn_from_distances(gmf_handle* h, const float* dist, int B, int N, int Sn, int k, int* knn_out, gmf_stream_t stream) {
  GMF_REQUIRE(h && dist && knn_out, GMF_ERR_BAD_ARG, "knn_from_distances: null pointer");
  GMF_REQUIRE(B > 0 && N > 1 && Sn > 0 && k > 0 && k <= 63 && k <= N - 1, GMF_ERR_UNSUPPORTED_SHAPE,
              "knn_from_distances: need 0 < k <= min(63, N - 1)");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_knn_seeds(nullptr, nullptr, dist, knn_out, B, N, Sn, k, S(stream), nullptr));
  return GMF_OK;
}

int gmf_knn_rows(gmf_handle* h, const float* feat_n, const int* rows, int B, int N, int Sn, int k, int* knn_out,
                 gmf_stream_t stream) {
  GMF_REQUIRE(h && feat_n && rows && knn_out, GMF_ERR_BAD_ARG, "knn_rows: null pointer");
  GMF_REQUIRE(B > 0 && N > 1 && Sn > 0 && k > 0 && k <= N - 1, GMF_ERR_UNSUPPORTED_SHAPE, "knn_rows: need 0 < k <= N-1");
  SetDevice sd(h, stream);
  hipStream_t st = S(stream);
  if (k > 63) {            // (the selection kernels behind the distance rows rank up to 64 candidates: larger k keeps the in-LDS kNN)
    GMF_REQUIRE((size_t)N * 4 <= 150 * 1024, GMF_ERR_UNSUPPORTED_SHAPE, "knn_rows: k > 63 needs N <= 38400 (the in-LDS kNN)");
    GMF_HIP(gmf::launch_knn_seeds(feat_n, rows, nullptr, knn_out, B, N, Sn, k, st));
    return GMF_OK;
  }
  // [r4] the pose head's own path (common.py:53-75 is what both compute): the rows' distances to every row of their pair by
  // MFMA (k_seed_dist on the split-fp16 feature image) and a threshold selection per row - 8 x 5000 x 5000: 10.7 ms -> below 1 ms
  // against one workgroup per row forming its distance row with vector instructions.  The distance rows are Sn x N floats per
  // pair: above 4 GiB the rows go through in slices, pair by pair.
  const int tiles = tiles_of(N);
  const size_t ld = (size_t)tiles * 32;
  const size_t cap = (size_t)1 << 30;                           // floats
  const bool whole = (size_t)B * Sn * ld <= cap;
  const int slice = whole ? Sn : (int)std::max<size_t>(1, std::min<size_t>((size_t)Sn, cap / ld));
  if (int rc = arena_reserve(h, arena_need((size_t)B * tiles * kTileFloats, 4) + arena_need((size_t)(whole ? B : 1) * slice * ld, 4))) return rc;
  float* fimg = arena_take<float>(h, (size_t)B * tiles * kTileFloats);
  float* dmat = arena_take<float>(h, (size_t)(whole ? B : 1) * slice * ld);
  GMF_HIP(gmf::launch_pack_rows_h2(feat_n, fimg, B, N, st, nullptr));
  if (whole) {
    GMF_HIP(gmf::launch_seed_dist(fimg, rows, dmat, B, N, Sn, st, nullptr));
    GMF_HIP(gmf::launch_knn_seeds(feat_n, rows, dmat, knn_out, B, N, Sn, k, st, nullptr));
    return GMF_OK;
  }
  for (int b = 0; b < B; ++b)
    for (int r0 = 0; r0 < Sn; r0 += slice) {
      const int n = std::min(slice, Sn - r0);
      const int* rb = rows + (size_t)b * Sn + r0;
      GMF_HIP(gmf::launch_seed_dist(fimg + (size_t)b * tiles * kTileFloats, rb, dmat, 1, N, n, st, nullptr));
      GMF_HIP(gmf::launch_knn_seeds(feat_n + (size_t)b * N * 128, rb, dmat, knn_out + ((size_t)b * Sn + r0) * k, 1, N, n, k, st, nullptr));
    }
  return GMF_OK;
}

int gmf_nn_match(gmf_handle* h, const float* F0, const float* F1, int N0, int N1, int d, int mode, int* idx_out,
                 float* dist_out, gmf_stream_t stream) {
  GMF_REQUIRE(h && F0 && F1 && idx_out && dist_out, GMF_ERR_BAD_ARG, "nn_match: null pointer");
  GMF_REQUIRE(N0 > 0 && N1 > 0 && d > 0, GMF_ERR_UNSUPPORTED_SHAPE, "nn_match: empty input");
  GMF_REQUIRE(mode >= 0 && mode <= 2, GMF_ERR_BAD_ARG, "nn_match: mode must be 0 (PointDSC), 1 (DGR L2) or 2 (DGR SquareL2)");
  const int K = gmf::padded_desc_width(d);
  GMF_REQUIRE(K > 0, GMF_ERR_UNSUPPORTED_SHAPE, "nn_match: descriptor width above 128 is not supported");
  SetDevice sd(h, stream);
  const size_t n0 = (size_t)tiles_of(N0) * 32 * K, n1 = (size_t)tiles_of(N1) * 32 * K + 4096;
  if (int rc = arena_reserve(h, arena_need(n0, 4) + arena_need(n1, 4) + arena_need((size_t)tiles_of(N1) * 32, 4) + arena_need((size_t)N0, 8))) return rc;
  float* i0 = arena_take<float>(h, n0);
  float* i1 = arena_take<float>(h, n1);      // + one stage of slack: the last stage may be read past the final tile
  float* nb = arena_take<float>(h, (size_t)tiles_of(N1) * 32);       // whole tiles: the padding holds +inf
  unsigned long long* best = arena_take<unsigned long long>(h, (size_t)N0);     // (score, index) of every row's winner: one atomic minimum per key split
  GMF_HIP(gmf::launch_nn_match(F0, F1, i0, i1, nb, best, idx_out, dist_out, N0, N1, d, mode, S(stream)));
  return GMF_OK;
}

int gmf_procrustes_batched(gmf_handle* h, const float* A, const float* Bp, const float* weights, int n, int k,
                           float weight_threshold, float* T44, gmf_stream_t stream) {
  GMF_REQUIRE(h && A && Bp && T44, GMF_ERR_BAD_ARG, "procrustes_batched: null pointer");
  GMF_REQUIRE(n > 0 && k > 0, GMF_ERR_UNSUPPORTED_SHAPE, "procrustes_batched: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_rigid_transform(A, Bp, weights, T44, n, k, weight_threshold, S(stream)));
  return GMF_OK;
}

int gmf_post_refinement(gmf_handle* h, const float* T_in, const float* src_keypts, const float* tgt_keypts, int B, int N,
                        float refine_threshold, int iters, float* T_out, gmf_stream_t stream) {
  GMF_REQUIRE(h && T_in && src_keypts && tgt_keypts && T_out, GMF_ERR_BAD_ARG, "post_refinement: null pointer");
  GMF_REQUIRE(B > 0 && N > 0 && iters >= 0, GMF_ERR_UNSUPPORTED_SHAPE, "post_refinement: empty input");
  GMF_REQUIRE(refine_threshold > 0.f, GMF_ERR_BAD_ARG, "post_refinement: threshold must be positive");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_post_refine(T_in, src_keypts, tgt_keypts, T_out, B, N, refine_threshold, iters, S(stream)));
  return GMF_OK;
}

int gmf_weighted_procrustes(gmf_handle* h, const float* X, const float* Y, const float* w, const int* offsets, int B,
                            float eps, float* R, float* t, gmf_stream_t stream) {
  GMF_REQUIRE(h && X && Y && w && offsets && R && t, GMF_ERR_BAD_ARG, "weighted_procrustes: null pointer");
  GMF_REQUIRE(B > 0, GMF_ERR_UNSUPPORTED_SHAPE, "weighted_procrustes: empty batch");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_weighted_procrustes(X, Y, w, offsets, B, eps, R, t, S(stream)));
  return GMF_OK;
}

int gmf_global_registration(gmf_handle* h, const float* X, const float* Y, const float* w, const int* offsets, int B,
                            float eps, float quantization_size, int max_iter, int max_break_count,
                            double break_threshold_ratio, float* R, float* t, float* stats, int max_points,
                            gmf_stream_t stream) {
  GMF_REQUIRE(h && X && Y && offsets && R && t && stats, GMF_ERR_BAD_ARG, "global_registration: null pointer");
  GMF_REQUIRE(B > 0, GMF_ERR_UNSUPPORTED_SHAPE, "global_registration: empty batch");
  GMF_REQUIRE(quantization_size > 0.f && max_iter >= 0 && max_break_count >= 1, GMF_ERR_BAD_ARG,
              "global_registration: quantization_size must be > 0, max_iter >= 0, max_break_count >= 1");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_global_registration(X, Y, w, offsets, B, eps, quantization_size, max_iter, max_break_count,
                                          break_threshold_ratio, R, t, stats, max_points > 0 ? max_points : (1 << 30), S(stream)));
  return GMF_OK;
}

int gmf_similarity_matrix(gmf_handle* h, const float* feat_n, int B, int N, float sigma, float* M, int ldm,
                          gmf_stream_t stream) {
  GMF_REQUIRE(h && feat_n && M, GMF_ERR_BAD_ARG, "similarity_matrix: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "similarity_matrix: empty input");
  GMF_REQUIRE(ldm >= N, GMF_ERR_BAD_ARG, "similarity_matrix: ldm (row stride of M in floats) must be >= N");
  GMF_REQUIRE(sigma != 0.f || h->sigma_dev, GMF_ERR_BAD_ARG, "similarity_matrix: sigma must be non-zero");
  SetDevice sd(h, stream);
  const size_t n_img = gmf::similarity_image_floats(B, N);
  if (int rc = arena_reserve(h, arena_need(n_img, 4))) return rc;
  float* img = arena_take<float>(h, n_img);
  GMF_HIP(gmf::launch_similarity_matrix(feat_n, img, M, B, N, ldm, sigma, S(stream), h->sigma_dev));
  return GMF_OK;
}

int gmf_spectral_matching_loss(gmf_handle* h, const float* M, int ldm, const float* gt_labels, int B, int N, int balanced,
                               float* loss_out, gmf_stream_t stream) {
  GMF_REQUIRE(h && M && gt_labels && loss_out, GMF_ERR_BAD_ARG, "spectral_matching_loss: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "spectral_matching_loss: empty input");
  GMF_REQUIRE(ldm >= N, GMF_ERR_BAD_ARG, "spectral_matching_loss: ldm (row stride of M in floats) must be >= N");
  SetDevice sd(h, stream);
  const size_t n_part = (size_t)2 * B * gmf::sm_parts_per_pair(B, N);
  if (int rc = arena_reserve(h, arena_need(n_part, 8) + arena_need((size_t)B, 8))) return rc;
  double* part = arena_take<double>(h, n_part);
  double* pair_loss = arena_take<double>(h, (size_t)B);
  GMF_HIP(gmf::launch_sm_loss(M, ldm, gt_labels, part, pair_loss, B, N, balanced, loss_out, S(stream)));
  return GMF_OK;
}

int gmf_spectral_matching_loss_fused(gmf_handle* h, const float* feat_n, const float* gt_labels, int B, int N, float sigma,
                                     int balanced, float* loss_out, gmf_stream_t stream) {
  GMF_REQUIRE(h && feat_n && gt_labels && loss_out, GMF_ERR_BAD_ARG, "spectral_matching_loss_fused: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "spectral_matching_loss_fused: empty input");
  GMF_REQUIRE(sigma != 0.f || h->sigma_dev, GMF_ERR_BAD_ARG, "spectral_matching_loss_fused: sigma must be non-zero");
  SetDevice sd(h, stream);
  const size_t n_img = gmf::similarity_image_floats(B, N);
  const size_t n_part = (size_t)2 * B * gmf::sm_fused_parts_per_pair(B, N);
  if (int rc = arena_reserve(h, arena_need(n_img, 4) + arena_need(n_part, 8) + arena_need((size_t)B, 8))) return rc;
  float* img = arena_take<float>(h, n_img);
  double* part = arena_take<double>(h, n_part);
  double* pair_loss = arena_take<double>(h, (size_t)B);
  GMF_HIP(gmf::launch_sm_loss_fused(feat_n, gt_labels, img, part, pair_loss, B, N, sigma, balanced, loss_out, S(stream), h->sigma_dev));
  return GMF_OK;
}

int gmf_spectral_matching_backward(gmf_handle* h, const float* feat_n, const float* gt_labels, int B, int N, float sigma,
                                   int balanced, float* d_feat_n, float* d_sigma, gmf_stream_t stream) {
  GMF_REQUIRE(h && feat_n && gt_labels && d_feat_n && d_sigma, GMF_ERR_BAD_ARG, "spectral_matching_backward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "spectral_matching_backward: empty input");
  GMF_REQUIRE(sigma != 0.f || h->sigma_dev, GMF_ERR_BAD_ARG, "spectral_matching_backward: sigma must be non-zero");
  SetDevice sd(h, stream);
  const size_t n_img = gmf::similarity_image_floats(B, N);
  const size_t n_part = (size_t)gmf::sm_backward_parts(B, N);
  if (int rc = arena_reserve(h, 2 * arena_need(n_img, 4) + arena_need((size_t)4 * B, 4) + arena_need(n_part, 8))) return rc;
  float* img = arena_take<float>(h, n_img);
  float* timg = arena_take<float>(h, n_img);
  float* consts = arena_take<float>(h, (size_t)4 * B);
  double* part = arena_take<double>(h, n_part);
  GMF_HIP(gmf::launch_sm_backward(feat_n, gt_labels, img, timg, consts, part, B, N, sigma, balanced, d_feat_n, d_sigma, S(stream), h->sigma_dev));
  return GMF_OK;
}

int gmf_classification_loss(gmf_handle* h, const float* pred, const float* gt, const float* weight, int B, int N,
                            int balanced, float* out6, gmf_stream_t stream) {
  GMF_REQUIRE(h && pred && gt && out6, GMF_ERR_BAD_ARG, "classification_loss: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "classification_loss: empty input");
  SetDevice sd(h, stream);
  const size_t n_part = (size_t)9 * gmf::classification_parts(B, N);
  if (int rc = arena_reserve(h, arena_need(n_part, 8))) return rc;
  double* part = arena_take<double>(h, n_part);
  GMF_HIP(gmf::launch_classification_loss(pred, gt, weight, part, B, N, balanced, out6, S(stream)));
  return GMF_OK;
}

int gmf_transformation_loss(gmf_handle* h, const float* trans, const float* gt_trans, const float* src_keypts,
                            const float* tgt_keypts, const float* probs, int B, int N, float re_thre, float te_thre,
                            float* out5, gmf_stream_t stream) {
  GMF_REQUIRE(h && trans && gt_trans && src_keypts && tgt_keypts && probs && out5, GMF_ERR_BAD_ARG,
              "transformation_loss: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "transformation_loss: empty input");
  SetDevice sd(h, stream);
  const size_t n_part = (size_t)3 * B * gmf::transformation_slices(B, N);
  if (int rc = arena_reserve(h, arena_need(n_part, 8))) return rc;
  double* part = arena_take<double>(h, n_part);
  GMF_HIP(gmf::launch_transformation_loss(trans, gt_trans, src_keypts, tgt_keypts, probs, part, B, N, re_thre, te_thre,
                                          out5, S(stream)));
  return GMF_OK;
}

int gmf_bias_relu_nhwc(gmf_handle* h, float* y, const float* bias, const float* residual, long long n_pixels, int C,
                       gmf_stream_t stream) {
  GMF_REQUIRE(h && y && bias, GMF_ERR_BAD_ARG, "bias_relu_nhwc: null pointer");
  GMF_REQUIRE(n_pixels > 0 && C > 0 && C % 4 == 0, GMF_ERR_UNSUPPORTED_SHAPE, "bias_relu_nhwc: need n_pixels > 0 and C a positive multiple of 4");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_bias_relu_nhwc(y, bias, residual, (long)n_pixels, C, S(stream)));
  return GMF_OK;
}

int gmf_stem_forward(gmf_handle* h, const float* x, long long sb, long long sc, long long sh, long long sw, const float* wimg,
                     const float* bias, float* y, int B, int H, int W, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && wimg && bias && y, GMF_ERR_BAD_ARG, "stem_forward: null pointer");
  GMF_REQUIRE(B > 0 && H > 0 && W > 0, GMF_ERR_UNSUPPORTED_SHAPE, "stem_forward: empty input");
  GMF_REQUIRE(B <= 65535, GMF_ERR_UNSUPPORTED_SHAPE, "stem_forward: at most 65535 images per call");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_stem_h2(x, (long)sb, (long)sc, (long)sh, (long)sw, wimg, bias, y, B, H, W, S(stream), h->tune.conv_small));
  return GMF_OK;
}

int gmf_conv_nhwc(gmf_handle* h, const float* x, const float* wimg, const float* bias, const float* residual, float* y,
                  int B, int H, int W, int cin, int cout, int ksize, int stride, int relu, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && wimg && bias && y, GMF_ERR_BAD_ARG, "conv_nhwc: null pointer");
  GMF_REQUIRE(B > 0 && H > 0 && W > 0, GMF_ERR_UNSUPPORTED_SHAPE, "conv_nhwc: empty input");
  const bool known = (cin == 64 && cout == 64 && ksize == 3 && stride == 1) || (cin == 64 && cout == 128 && ksize == 3 && stride == 2) ||
                     (cin == 128 && cout == 128 && ksize == 3 && stride == 1) || (cin == 64 && cout == 128 && ksize == 1 && stride == 2);
  GMF_REQUIRE(known, GMF_ERR_UNSUPPORTED_SHAPE,
              "conv_nhwc: supported are the ResNet-34 layer1 / layer2 shapes (64->64 3x3 s1, 64->128 3x3 s2, 128->128 3x3 s1, 64->128 1x1 s2)");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_conv_nhwc_h2(h->tune, x, wimg, bias, residual, y, B, H, W, cin, cout, ksize, stride, relu, S(stream)));
  return GMF_OK;
}

}  // extern "C"
