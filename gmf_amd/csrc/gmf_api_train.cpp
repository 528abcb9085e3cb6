// C ABI, training primitives (include/gmf_hip.h, section "training primitives"): argument checks and the split-K / column-sum
// workspaces around the kernels of train_kernels.hip.  Kept apart from gmf_api.cpp so that the inference launch sequences
// (and the source hash bench.py ties its profile to) do not move when the training path grows.
#include "api_common.hpp"
#include "launchers_pose.hpp"

extern "C" {

int gmf_gemm_f32(gmf_handle* h, int trans_a, int trans_b, const float* A, const float* B, float* C, const float* bias,
                 const float* residual, int M, int N, int K, long long lda, long long ldb, long long ldc, long long stride_a,
                 long long stride_b, long long stride_c, int batch, float alpha, int relu, gmf_stream_t stream) {
  GMF_REQUIRE(h && A && B && C, GMF_ERR_BAD_ARG, "gemm_f32: null pointer");
  GMF_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, GMF_ERR_UNSUPPORTED_SHAPE, "gemm_f32: empty problem");
  // leading dimensions: a row of op(X) as it is stored must fit its stride (a smaller one reads or writes out of bounds on the device)
  GMF_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, GMF_ERR_BAD_ARG,
              "gemm_f32: need lda >= (trans_a ? M : K), ldb >= (trans_b ? K : N), ldc >= N (row strides in floats)");
  GMF_REQUIRE(batch == 1 || (stride_a >= 0 && stride_b >= 0 && stride_c >= N), GMF_ERR_BAD_ARG,
              "gemm_f32: batched outputs overlap (stride_c < N) or a batch stride is negative");
  // grid limits: blockIdx.y = ceil(M / 64) at most, blockIdx.z = batch * ksplits (ksplits <= 128)
  GMF_REQUIRE((M + 63) / 64 <= 65535, GMF_ERR_UNSUPPORTED_SHAPE, "gemm_f32: M too large for one launch (ceil(M / 64) > 65535)");
  SetDevice sd(h, stream);
  const int ksplits = gmf::gemm_ksplits(trans_a != 0, trans_b != 0, A, B, M, N, K, (long)lda, (long)ldb, (long)stride_a, (long)stride_b, batch);
  GMF_REQUIRE((long long)batch * ksplits <= 65535, GMF_ERR_UNSUPPORTED_SHAPE, "gemm_f32: batch * k-splits exceeds 65535 workgroups in z: split the batch");
  float* part = nullptr;
  if (ksplits > 1) {
    if (int rc = arena_reserve(h, arena_need((size_t)batch * ksplits * M * N, 4))) return rc;
    part = arena_take<float>(h, (size_t)batch * ksplits * M * N);
  }
  GMF_HIP(gmf::launch_gemm_f32(trans_a != 0, trans_b != 0, A, B, C, bias, residual, M, N, K, (long)lda, (long)ldb, (long)ldc,
                               (long)stride_a, (long)stride_b, (long)stride_c, batch, alpha, part, ksplits, relu ? 1 : 0, S(stream)));
  return GMF_OK;
}

int gmf_lcpe(gmf_handle* h, int backward, const float* x, const float* w, const float* bias, float* y, int rows, int L, int C,
             gmf_stream_t stream) {
  GMF_REQUIRE(h && x && w && y && (backward || bias), GMF_ERR_BAD_ARG, "lcpe: null pointer");
  GMF_REQUIRE(rows > 0 && L > 0 && C > 0 && rows % L == 0, GMF_ERR_UNSUPPORTED_SHAPE, "lcpe: rows must be a positive multiple of L");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_lcpe(backward != 0, x, w, bias, y, rows, L, C, S(stream)));
  return GMF_OK;
}

int gmf_layernorm_forward(gmf_handle* h, const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          long long rows, int C, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && gamma && beta && y && mean && rstd, GMF_ERR_BAD_ARG, "layernorm_forward: null pointer");
  GMF_REQUIRE(rows > 0 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "layernorm_forward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_ln_fwd(x, gamma, beta, y, mean, rstd, (long)rows, C, S(stream)));
  return GMF_OK;
}

int gmf_layernorm_backward(gmf_handle* h, const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* dx_add, float* dx, long long rows, int C, gmf_stream_t stream) {
  GMF_REQUIRE(h && dy && x && gamma && mean && rstd && dx, GMF_ERR_BAD_ARG, "layernorm_backward: null pointer");
  GMF_REQUIRE(rows > 0 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "layernorm_backward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_ln_bwd(dy, x, gamma, mean, rstd, dx_add, dx, (long)rows, C, S(stream)));
  return GMF_OK;
}

int gmf_softmax_rows(gmf_handle* h, int backward, const float* a, const float* b, const float* mul, float* out, long long rows, int T,
                     float scale, gmf_stream_t stream) {
  GMF_REQUIRE(h && a && out && (!backward || b), GMF_ERR_BAD_ARG, "softmax_rows: null pointer");
  GMF_REQUIRE(rows > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "softmax_rows: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_softmax(backward != 0, a, b, mul, out, (long)rows, T, scale, S(stream)));
  return GMF_OK;
}

int gmf_geglu(gmf_handle* h, int backward, const float* hdn, const float* dg, float* out, long long rows, int H,
              gmf_stream_t stream) {
  GMF_REQUIRE(h && hdn && out && (!backward || dg), GMF_ERR_BAD_ARG, "geglu: null pointer");
  GMF_REQUIRE(rows > 0 && H > 0, GMF_ERR_UNSUPPORTED_SHAPE, "geglu: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_geglu(backward != 0, hdn, dg, out, (long)rows, H, S(stream)));
  return GMF_OK;
}

int gmf_colsum(gmf_handle* h, const float* x, const float* y, const float* mean, const float* rstd, const float* cmean,
               const float* crstd, int center_x, int shift, int L, long long rows, int C, const float* relu_y, int dual, float* out,
               gmf_stream_t stream) {
  GMF_REQUIRE(h && x && out, GMF_ERR_BAD_ARG, "colsum: null pointer");
  GMF_REQUIRE((mean == nullptr) == (rstd == nullptr) && (!mean || y), GMF_ERR_BAD_ARG, "colsum: mean and rstd come together, with y");
  GMF_REQUIRE(rows > 0 && C > 0 && L > 0 && rows % L == 0, GMF_ERR_UNSUPPORTED_SHAPE, "colsum: rows must be a positive multiple of L");
  GMF_REQUIRE(shift >= -1 && shift <= 1, GMF_ERR_BAD_ARG, "colsum: shift must be -1, 0 or 1");
  SetDevice sd(h, stream);
  GMF_REQUIRE(!dual || y, GMF_ERR_BAD_ARG, "colsum: dual needs y (without y both sums are the same)");
  const size_t n_part = (size_t)gmf::colsum_chunks((long)rows) * C * (dual ? 2 : 1);
  if (int rc = arena_reserve(h, arena_need(n_part, 4))) return rc;
  float* part = arena_take<float>(h, n_part);
  GMF_REQUIRE(!center_x || cmean, GMF_ERR_BAD_ARG, "colsum: center_x needs cmean");
  GMF_HIP(gmf::launch_colsum(x, y, mean, rstd, cmean, crstd, center_x ? 1 : 0, shift, L, (long)rows, C, part, out, S(stream), relu_y,
                             dual != 0));
  return GMF_OK;
}

int gmf_batchnorm_train_forward(gmf_handle* h, const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                float* rstd, float* running_mean, float* running_var, long long rows, int C, float eps,
                                float momentum, int relu, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && gamma && beta && y && mean && rstd, GMF_ERR_BAD_ARG, "batchnorm_train_forward: null pointer");
  GMF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GMF_ERR_BAD_ARG, "batchnorm_train_forward: running stats come together");
  GMF_REQUIRE(rows > 1 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "batchnorm_train_forward: need more than one row");
  SetDevice sd(h, stream);
  const size_t n_part = (size_t)gmf::colsum_chunks((long)rows) * C;
  if (int rc = arena_reserve(h, arena_need(n_part, 4) + 2 * arena_need((size_t)C, 4))) return rc;
  float* part = arena_take<float>(h, n_part);
  float* sum = arena_take<float>(h, (size_t)C);
  float* sumsq = arena_take<float>(h, (size_t)C);
  hipStream_t st = S(stream);
  GMF_HIP(gmf::launch_colsum(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, (int)rows, (long)rows, C, part, sum, st));
  GMF_HIP(gmf::launch_bn_finish(sum, nullptr, mean, rstd, nullptr, nullptr, C, (long)rows, eps, momentum, st));
  // sum of squared deviations: x' = x - mean (centred), y' = (x - mean) * 1
  GMF_HIP(gmf::launch_colsum(x, x, nullptr, nullptr, mean, nullptr, 1, 0, (int)rows, (long)rows, C, part, sumsq, st));
  GMF_HIP(gmf::launch_bn_finish(sum, sumsq, mean, rstd, running_mean, running_var, C, (long)rows, eps, momentum, st));
  GMF_HIP(gmf::launch_bn_apply(x, mean, rstd, gamma, beta, y, (long)rows, C, relu ? 1 : 0, st));
  return GMF_OK;
}

int gmf_batchnorm_train_backward(gmf_handle* h, const float* dy, const float* x, const float* y_relu, const float* mean,
                                 const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, long long rows, int C,
                                 gmf_stream_t stream) {
  GMF_REQUIRE(h && dy && x && mean && rstd && gamma && dx && dgamma && dbeta, GMF_ERR_BAD_ARG, "batchnorm_train_backward: null pointer");
  GMF_REQUIRE(rows > 1 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "batchnorm_train_backward: need more than one row");
  SetDevice sd(h, stream);
  const size_t n_part = (size_t)gmf::colsum_chunks((long)rows) * C * 2;
  if (int rc = arena_reserve(h, arena_need(n_part, 4))) return rc;
  float* part = arena_take<float>(h, n_part);
  hipStream_t st = S(stream);
  // ONE pass over dy: dgamma = sum g xhat, dbeta = sum g, with g = dy masked by the output of the ReLU that followed
  GMF_HIP(gmf::launch_colsum(dy, x, nullptr, nullptr, mean, rstd, 0, 0, (int)rows, (long)rows, C, part, dgamma, st, y_relu, true, dbeta));
  GMF_HIP(gmf::launch_bn_bwd(dy, x, mean, rstd, gamma, dbeta, dgamma, dx, (long)rows, C, st, y_relu));
  return GMF_OK;
}

int gmf_normalize_rows(gmf_handle* h, int backward, const float* a, const float* dy, float* nrm, float* out, long long rows, int C,
                       gmf_stream_t stream) {
  GMF_REQUIRE(h && a && nrm && out && (!backward || dy), GMF_ERR_BAD_ARG, "normalize_rows: null pointer");
  GMF_REQUIRE(rows > 0 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "normalize_rows: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_normalize(backward != 0, a, dy, nrm, out, (long)rows, C, S(stream)));
  return GMF_OK;
}

int gmf_relu_backward(gmf_handle* h, const float* dy, const float* y, float* out, long long total, gmf_stream_t stream) {
  GMF_REQUIRE(h && dy && y && out, GMF_ERR_BAD_ARG, "relu_backward: null pointer");
  GMF_REQUIRE(total > 0, GMF_ERR_UNSUPPORTED_SHAPE, "relu_backward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_relu_bwd(dy, y, out, (long)total, S(stream)));
  return GMF_OK;
}

int gmf_classification_backward(gmf_handle* h, const float* pred, const float* gt, const float* weight, int B, int N, int balanced,
                                float* d_pred, gmf_stream_t stream) {
  GMF_REQUIRE(h && pred && gt && d_pred, GMF_ERR_BAD_ARG, "classification_backward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "classification_backward: empty input");
  SetDevice sd(h, stream);
  if (int rc = arena_reserve(h, arena_need(1, 4))) return rc;
  float* pw = arena_take<float>(h, 1);
  GMF_HIP(gmf::launch_bce_bwd(pred, gt, weight, d_pred, pw, balanced, (long)B * N, S(stream)));
  return GMF_OK;
}

int gmf_spectral_matching_dense_backward(gmf_handle* h, const float* M, int ldm, const float* gt_labels, int B, int N, int balanced,
                                         float* dM, gmf_stream_t stream) {
  GMF_REQUIRE(h && M && gt_labels && dM, GMF_ERR_BAD_ARG, "spectral_matching_dense_backward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0 && ldm >= N, GMF_ERR_UNSUPPORTED_SHAPE, "spectral_matching_dense_backward: need ldm >= N > 0");
  SetDevice sd(h, stream);
  if (int rc = arena_reserve(h, arena_need((size_t)4 * B, 4))) return rc;
  float* consts = arena_take<float>(h, (size_t)4 * B);
  GMF_HIP(gmf::launch_sm_dense_bwd(M, ldm, gt_labels, consts, dM, B, N, balanced, S(stream)));
  return GMF_OK;
}

int gmf_similarity_backward(gmf_handle* h, const float* feat_n, const float* dM, int B, int N, float sigma, float* d_feat_n,
                            float* d_sigma, gmf_stream_t stream) {
  GMF_REQUIRE(h && feat_n && dM && d_feat_n && d_sigma, GMF_ERR_BAD_ARG, "similarity_backward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "similarity_backward: empty input");
  GMF_REQUIRE(sigma != 0.f || h->sigma_dev, GMF_ERR_BAD_ARG, "similarity_backward: sigma must be non-zero");
  SetDevice sd(h, stream);
  const size_t nn = (size_t)B * N * N, rows = (size_t)B * N;
  const size_t n_part = (size_t)gmf::colsum_chunks((long)rows);
  if (int rc = arena_reserve(h, 2 * arena_need(nn, 4) + arena_need(rows, 4) + arena_need(n_part, 4))) return rc;
  float* Sm = arena_take<float>(h, nn);
  float* G = arena_take<float>(h, nn);
  float* rowdsig = arena_take<float>(h, rows);
  float* part = arena_take<float>(h, n_part);
  hipStream_t st = S(stream);
  const long ld = 128, sF = (long)N * 128, sN = (long)N * N;
  // S = Fn Fn^T per pair, G = dM * [0 <= u <= 1] / sigma^2, dFn = G Fn + G^T Fn, dsigma = sum of the row partials
  GMF_HIP(gmf::launch_gemm_f32(false, true, feat_n, feat_n, Sm, nullptr, nullptr, N, N, 128, ld, ld, N, sF, sF, sN, B, 1.0f, nullptr, 1, 0, st));
  GMF_HIP(gmf::launch_sim_bwd_G(Sm, dM, G, rowdsig, B, N, sigma, st, h->sigma_dev));
  GMF_HIP(gmf::launch_gemm_f32(false, false, G, feat_n, d_feat_n, nullptr, nullptr, N, 128, N, N, ld, ld, sN, sF, sF, B, 1.0f, nullptr, 1, 0, st));
  GMF_HIP(gmf::launch_gemm_f32(true, false, G, feat_n, d_feat_n, nullptr, d_feat_n, N, 128, N, N, ld, ld, sN, sF, sF, B, 1.0f, nullptr, 1, 0, st));
  GMF_HIP(gmf::launch_colsum(rowdsig, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, (int)rows, (long)rows, 1, part, d_sigma, st));
  return GMF_OK;
}

int gmf_compat_dense(gmf_handle* h, const float* src_keypts, const float* tgt_keypts, int B, int N, float sigma_d, float* out,
                     gmf_stream_t stream) {
  GMF_REQUIRE(h && src_keypts && tgt_keypts && out, GMF_ERR_BAD_ARG, "compat_dense: null pointer");
  GMF_REQUIRE(B > 0 && N > 0 && N <= 65535 && B <= 65535, GMF_ERR_UNSUPPORTED_SHAPE, "compat_dense: need 0 < B, N <= 65535");
  GMF_REQUIRE(sigma_d > 0.f, GMF_ERR_BAD_ARG, "compat_dense: sigma_d must be positive");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_compat_dense(src_keypts, tgt_keypts, out, B, N, sigma_d, S(stream)));
  return GMF_OK;
}

int gmf_transformation_loss_backward(gmf_handle* h, const float* trans, const float* src_keypts, const float* tgt_keypts,
                                     const float* probs, int B, int N, float* d_trans, gmf_stream_t stream) {
  GMF_REQUIRE(h && trans && src_keypts && tgt_keypts && probs && d_trans, GMF_ERR_BAD_ARG, "transformation_loss_backward: null pointer");
  GMF_REQUIRE(B > 0 && N > 0, GMF_ERR_UNSUPPORTED_SHAPE, "transformation_loss_backward: empty input");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_tl_backward(trans, src_keypts, tgt_keypts, probs, d_trans, B, N, S(stream)));
  return GMF_OK;
}

int gmf_pose_head_backward(gmf_handle* h, const gmf_pose_params* p, const float* feat_n, const float* src_keypts,
                           const float* tgt_keypts, const int* knn_idx, const float* fitness, const float* d_final_trans, int B,
                           int N, float* d_feat_n, float* d_sigma, gmf_stream_t stream) {
  GMF_REQUIRE(h && p && feat_n && src_keypts && tgt_keypts && knn_idx && fitness && d_final_trans && d_feat_n && d_sigma,
              GMF_ERR_BAD_ARG, "pose_head_backward: null pointer");
  const int Sn = p->num_seeds, k = p->k, iters = p->num_iterations;
  GMF_REQUIRE(B > 0 && N > 1 && Sn > 0 && Sn <= N, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head_backward: need N > 1, 0 < num_seeds <= N");
  GMF_REQUIRE(k > 0 && k <= 64 && k <= N - 1, GMF_ERR_UNSUPPORTED_SHAPE, "pose_head_backward: need 0 < k <= min(64, N-1)");
  GMF_REQUIRE(iters > 0 && iters <= 64, GMF_ERR_BAD_ARG, "pose_head_backward: bad num_iterations");
  GMF_REQUIRE(p->refine_iters == 0, GMF_ERR_BAD_ARG, "pose_head_backward: the post-refinement (test mode) is not differentiable");
  GMF_REQUIRE((p->sigma > 0.f || h->sigma_dev) && p->sigma_d > 0.f, GMF_ERR_BAD_ARG, "pose_head_backward: sigma, sigma_d must be positive");
  SetDevice sd(h, stream);
  hipStream_t st = S(stream);
  const size_t BS = (size_t)B * Sn;
  if (int rc = arena_reserve(h, arena_need(BS * iters * k, 4) + arena_need(BS * iters, 1) + arena_need(1, 4)))
    return rc;
  float* snaps = arena_take<float>(h, BS * iters * k);
  unsigned char* conv = arena_take<unsigned char>(h, BS * iters);
  int* stop_it = arena_take<int>(h, 1);
  // the forward's iterates and convergence flags of every seed (the stop iteration is a property of all seeds of a pair)
  GMF_HIP(gmf::launch_seed_power(feat_n, src_keypts, tgt_keypts, knn_idx, snaps, conv, nullptr, B, N, Sn, k, iters, p->sigma, p->sigma_d, st, nullptr, h->sigma_dev));
  GMF_HIP(hipMemsetAsync(d_feat_n, 0, (size_t)B * N * kC * sizeof(float), st));
  if (B > 1) GMF_HIP(gmf::launch_stop_iteration(conv, B, Sn, iters, stop_it, st));     // the reference's allclose spans the batch
  GMF_HIP(gmf::launch_pose_best_backward(feat_n, src_keypts, tgt_keypts, knn_idx, fitness, snaps, conv, d_final_trans, d_feat_n,
                                         d_sigma, B, N, Sn, k, iters, p->sigma, p->sigma_d, B > 1 ? stop_it : nullptr, st, h->sigma_dev));
  return GMF_OK;
}

int gmf_weighted_procrustes_backward(gmf_handle* h, const float* X, const float* Y, const float* w, const int* offsets, int B,
                                     float eps, const float* d_R, const float* d_t, float* d_w, gmf_stream_t stream) {
  GMF_REQUIRE(h && X && Y && w && offsets && d_R && d_t && d_w, GMF_ERR_BAD_ARG, "weighted_procrustes_backward: null pointer");
  GMF_REQUIRE(B > 0, GMF_ERR_UNSUPPORTED_SHAPE, "weighted_procrustes_backward: empty batch");
  SetDevice sd(h, stream);
  GMF_HIP(gmf::launch_wp_backward(X, Y, w, offsets, B, eps, d_R, d_t, d_w, S(stream)));
  return GMF_OK;
}

}  // extern "C"
