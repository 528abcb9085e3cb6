// Internal C++ launch entry points of the pose-head kernels.  Public C ABI: include/gmf_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include "launchers.hpp"

namespace gmf {

hipError_t launch_nms_keys(const Tuning& tune, const float* src, const float* scores, float* keys, int B, int N, float R, hipStream_t s,
                           float* scratch = nullptr,      // scratch: nms_scratch_floats(B, N) floats, or null = all-pairs form
                           const PairTab* ptab = nullptr, long total_rows = 0,    // ragged batch: the per-pair table and sum n
                           bool keys_preset = false);     // [r5] keys already hold a copy of the scores (k_pack_rows_h2's copy role)
size_t nms_scratch_floats(int B, int N);
hipError_t launch_sort_topk(const Tuning& tune, const float* keys, int* out_idx, int B, int N, int S, hipStream_t s, const PairTab* ptab = nullptr);
hipError_t launch_knn_seeds(const float* feat_n, const int* seeds, const float* dist_in, int* knn_idx, int B, int N, int S,
                            int k, hipStream_t s, const PairTab* ptab = nullptr);
// hsum non-null ([B,S,15] doubles): the centroids and H of the weighted Kabsch problem of the LAST power iterate are summed
// in the same kernel; launch_seed_kabsch given the same buffer then only runs the SVD (and redoes the sums of pairs whose
// iteration stopped earlier)
hipError_t launch_seed_power(const float* feat_n /* row-major [B, N, 128] */, const float* src, const float* tgt, const int* knn_idx, float* snaps,
                             unsigned char* conv, double* hsum, int B, int N, int S, int k, int iters, float sigma,
                             float sigma_d, hipStream_t s, const PairTab* ptab = nullptr, const float* sigma_dev = nullptr);
hipError_t launch_seed_kabsch(const float* src, const float* tgt, const int* knn_idx, const float* snaps,
                              const unsigned char* conv, float* seed_T, int B, int N, int S, int k, int iters,
                              const double* hsum, int* stop_scratch, hipStream_t s, const PairTab* ptab = nullptr);   // stop_scratch: 1 int (batches: the
                                                                                       // batch-wide stop iteration lands there)
hipError_t launch_stop_iteration(const unsigned char* conv, int B, int S, int iters, int* stop_out, hipStream_t s);
hipError_t launch_score_hyp(const float* src, const float* tgt, const float* seed_T, int* counts, int B, int N, int S,
                            float tau, hipStream_t s, const PairTab* ptab = nullptr);
hipError_t launch_finalize_pose(const float* src, const float* tgt, const float* seed_T, const int* counts, float* fitness,
                                float* final_T, float* labels, int* best, int B, int N, int S, float tau, float refine_thr,
                                int refine_iters, hipStream_t s, const PairTab* ptab = nullptr);
hipError_t launch_post_refine(const float* T_in, const float* src, const float* tgt, float* T_out, int B, int N, float thr,
                              int iters, hipStream_t s);
hipError_t launch_rigid_transform(const float* A, const float* Bp, const float* w, float* T, int n, int k,
                                  float weight_threshold, hipStream_t s);
hipError_t launch_weighted_procrustes(const float* X, const float* Y, const float* w, const int* offsets, int B, float eps,
                                      float* R, float* t, hipStream_t s);
hipError_t launch_global_registration(const float* X, const float* Y, const float* w, const int* offsets, int B, float eps,
                                      float qsize, int max_iter, int max_break, double ratio, float* R, float* t,
                                      float* stats, int max_n, hipStream_t s);

// backward of the pose head (pose_backward.hip)
hipError_t launch_tl_backward(const float* trans, const float* src, const float* tgt, const float* probs, float* g_trans,
                              int B, int N, hipStream_t s);
hipError_t launch_pose_best_backward(const float* feat_n, const float* src, const float* tgt, const int* knn_idx,
                                     const float* fitness, const float* snaps, const unsigned char* conv, const float* g_T,
                                     float* g_feat, float* g_sigma, int B, int N, int S, int k, int iters, float sigma,
                                     float sigma_d, const int* stop_batch, hipStream_t s, const float* sigma_dev = nullptr);
hipError_t launch_wp_backward(const float* X, const float* Y, const float* w, const int* offsets, int B, float eps,
                              const float* g_R, const float* g_t, float* g_w, hipStream_t s);

}  // namespace gmf
