// C ABI, training primitives (include/gmf_hip.h, section "training primitives"): argument checks and the split-K / column-sum
// workspaces around the kernels of train_kernels.hip.  Kept apart from gmf_api.cpp so that the inference launch sequences
// (and the source hash bench.py ties its profile to) do not move when the training path grows.
#include "api_common.hpp"

extern "C" {

int gmf_gemm_f32(gmf_handle* h, int trans_a, int trans_b, const float* A, const float* B, float* C, const float* bias,
                 const float* residual, int M, int N, int K, long long lda, long long ldb, long long ldc, long long stride_a,
                 long long stride_b, long long stride_c, int batch, float alpha, gmf_stream_t stream) {
  GMF_REQUIRE(h && A && B && C, GMF_ERR_BAD_ARG, "gemm_f32: null pointer");
  GMF_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, GMF_ERR_UNSUPPORTED_SHAPE, "gemm_f32: empty problem");
  GMF_REQUIRE((long long)batch * ((M + 127) / 128) <= 2000000 , GMF_ERR_UNSUPPORTED_SHAPE, "gemm_f32: grid too large");
  SetDevice sd(h);
  const int ksplits = gmf::gemm_ksplits(M, N, K, batch);
  float* part = nullptr;
  if (ksplits > 1) {
    if (int rc = arena_reserve(h, arena_need((size_t)batch * ksplits * M * N, 4))) return rc;
    part = arena_take<float>(h, (size_t)batch * ksplits * M * N);
  }
  GMF_HIP(gmf::launch_gemm_f32(trans_a != 0, trans_b != 0, A, B, C, bias, residual, M, N, K, (long)lda, (long)ldb, (long)ldc,
                               (long)stride_a, (long)stride_b, (long)stride_c, batch, alpha, part, ksplits, S(stream)));
  return GMF_OK;
}

int gmf_lcpe(gmf_handle* h, int backward, const float* x, const float* w, const float* bias, float* y, int rows, int L, int C,
             gmf_stream_t stream) {
  GMF_REQUIRE(h && x && w && y && (backward || bias), GMF_ERR_BAD_ARG, "lcpe: null pointer");
  GMF_REQUIRE(rows > 0 && L > 0 && C > 0 && rows % L == 0, GMF_ERR_UNSUPPORTED_SHAPE, "lcpe: rows must be a positive multiple of L");
  SetDevice sd(h);
  GMF_HIP(gmf::launch_lcpe(backward != 0, x, w, bias, y, rows, L, C, S(stream)));
  return GMF_OK;
}

int gmf_layernorm_forward(gmf_handle* h, const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          long long rows, int C, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && gamma && beta && y && mean && rstd, GMF_ERR_BAD_ARG, "layernorm_forward: null pointer");
  GMF_REQUIRE(rows > 0 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "layernorm_forward: empty input");
  SetDevice sd(h);
  GMF_HIP(gmf::launch_ln_fwd(x, gamma, beta, y, mean, rstd, (long)rows, C, S(stream)));
  return GMF_OK;
}

int gmf_layernorm_backward(gmf_handle* h, const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* dx_add, float* dx, long long rows, int C, gmf_stream_t stream) {
  GMF_REQUIRE(h && dy && x && gamma && mean && rstd && dx, GMF_ERR_BAD_ARG, "layernorm_backward: null pointer");
  GMF_REQUIRE(rows > 0 && C > 0, GMF_ERR_UNSUPPORTED_SHAPE, "layernorm_backward: empty input");
  SetDevice sd(h);
  GMF_HIP(gmf::launch_ln_bwd(dy, x, gamma, mean, rstd, dx_add, dx, (long)rows, C, S(stream)));
  return GMF_OK;
}

int gmf_softmax_rows(gmf_handle* h, int backward, const float* a, const float* b, float* out, long long rows, int T, float scale,
                     gmf_stream_t stream) {
  GMF_REQUIRE(h && a && out && (!backward || b), GMF_ERR_BAD_ARG, "softmax_rows: null pointer");
  GMF_REQUIRE(rows > 0 && T > 0, GMF_ERR_UNSUPPORTED_SHAPE, "softmax_rows: empty input");
  SetDevice sd(h);
  GMF_HIP(gmf::launch_softmax(backward != 0, a, b, out, (long)rows, T, scale, S(stream)));
  return GMF_OK;
}

int gmf_geglu(gmf_handle* h, int backward, const float* hdn, const float* dg, float* out, long long rows, int H,
              gmf_stream_t stream) {
  GMF_REQUIRE(h && hdn && out && (!backward || dg), GMF_ERR_BAD_ARG, "geglu: null pointer");
  GMF_REQUIRE(rows > 0 && H > 0, GMF_ERR_UNSUPPORTED_SHAPE, "geglu: empty input");
  SetDevice sd(h);
  GMF_HIP(gmf::launch_geglu(backward != 0, hdn, dg, out, (long)rows, H, S(stream)));
  return GMF_OK;
}

int gmf_colsum(gmf_handle* h, const float* x, const float* y, const float* mean, const float* rstd, int shift, int L, long long rows,
               int C, float* out, gmf_stream_t stream) {
  GMF_REQUIRE(h && x && out, GMF_ERR_BAD_ARG, "colsum: null pointer");
  GMF_REQUIRE((mean == nullptr) == (rstd == nullptr) && (!mean || y), GMF_ERR_BAD_ARG, "colsum: mean and rstd come together, with y");
  GMF_REQUIRE(rows > 0 && C > 0 && L > 0 && rows % L == 0, GMF_ERR_UNSUPPORTED_SHAPE, "colsum: rows must be a positive multiple of L");
  GMF_REQUIRE(shift >= -1 && shift <= 1, GMF_ERR_BAD_ARG, "colsum: shift must be -1, 0 or 1");
  SetDevice sd(h);
  const size_t n_part = (size_t)gmf::colsum_chunks((long)rows) * C;
  if (int rc = arena_reserve(h, arena_need(n_part, 4))) return rc;
  float* part = arena_take<float>(h, n_part);
  GMF_HIP(gmf::launch_colsum(x, y, mean, rstd, shift, L, (long)rows, C, part, out, S(stream)));
  return GMF_OK;
}

}  // extern "C"
