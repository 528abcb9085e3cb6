// A C++ host of the C ABI (include/gmf_hip.h) with no Python and no torch in the process: device buffers from hipMalloc,
// the DGR pose calls on synthetic scenes with a known rigid motion, status codes and the error string - and, given a dump file,
// THE HOT PATH: state_dict tensors -> gmf_encoder_pack_weights -> gmf_encoder_forward -> gmf_pose_head, checked against the
// reference's own outputs (golden F4) carried in the same dump.
//   hipcc -O2 -I include tests/abi_cpp/abi_host.cpp -L gmf_amd -lgmf_hip -Wl,-rpath,$PWD/gmf_amd -o abi_host && ./abi_host [dump]
// Dump format (written by tests/test_gpu_parity.py::test_cpp_host_of_the_c_abi, little endian): "GMFD", int32 count, then per
// tensor: int32 name length, name, int32 ndim, int64 shape[4], float32 data.  Names: the reference's state_dict keys,
// "input.corr_pos" [B,N,6], "input.src_keypts", "input.tgt_keypts" [B,N,3], "input.p_tokens", "input.q_tokens" [B,T,128],
// "expect.logits" [B,N], "expect.final_trans" [B,4,4], "param.num_layers" [1].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "gmf_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

struct Named { std::string name; int ndim; long long shape[4]; std::vector<float> data; };

static bool read_dump(const char* path, std::vector<Named>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char magic[4];
  int32_t n = 0;
  bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "GMFD", 4) == 0 && std::fread(&n, 4, 1, f) == 1 && n > 0 && n < 100000;
  for (int i = 0; ok && i < n; ++i) {
    Named t;
    int32_t len = 0, nd = 0;
    int64_t shp[4];
    ok = std::fread(&len, 4, 1, f) == 1 && len > 0 && len < 4096;
    if (!ok) break;
    t.name.resize(len);
    ok = std::fread(&t.name[0], 1, len, f) == (size_t)len && std::fread(&nd, 4, 1, f) == 1 && std::fread(shp, 8, 4, f) == 4;
    if (!ok) break;
    t.ndim = nd;
    size_t numel = 1;
    for (int d = 0; d < 4; ++d) { t.shape[d] = shp[d]; numel *= (size_t)shp[d]; }
    t.data.resize(numel);
    ok = std::fread(t.data.data(), 4, numel, f) == numel;
    out.push_back(std::move(t));
  }
  std::fclose(f);
  return ok;
}

// state_dict -> packed weights -> logits -> pose, all through the C ABI, against the reference's outputs in the dump
static int run_hot_path(gmf_handle* h, const char* path, hipStream_t st) {
  std::vector<Named> ts;
  if (!read_dump(path, ts)) { std::printf("cannot read dump %s\n", path); return 1; }
  std::map<std::string, const Named*> by;
  std::vector<gmf_tensor> sd;
  for (const Named& t : ts) {
    by[t.name] = &t;
    if (t.name.rfind("input.", 0) == 0 || t.name.rfind("expect.", 0) == 0 || t.name.rfind("param.", 0) == 0) continue;
    gmf_tensor g;
    g.name = t.name.c_str(); g.data = t.data.data(); g.ndim = t.ndim;
    for (int d = 0; d < 4; ++d) g.shape[d] = t.shape[d];
    sd.push_back(g);
  }
  for (const char* k : {"input.corr_pos", "input.src_keypts", "input.tgt_keypts", "input.p_tokens", "input.q_tokens", "expect.logits",
                        "expect.final_trans", "param.num_layers"})
    if (!by.count(k)) { std::printf("dump lacks %s\n", k); return 1; }
  const int L = (int)by["param.num_layers"]->data[0];
  const Named& cp = *by["input.corr_pos"];
  const int B = (int)cp.shape[0], N = (int)cp.shape[1], T = (int)by["input.p_tokens"]->shape[1];
  gmf_packed_encoder* pk = nullptr;
  int rc = gmf_encoder_pack_weights(h, sd.data(), (int)sd.size(), L, 0, &pk);
  if (rc != GMF_OK) { std::printf("gmf_encoder_pack_weights: status %d, %s\n", rc, gmf_last_error_string(h)); return 1; }
  float sigma = 0.f, sigma_d = 0.f;
  int split = 0;
  gmf_packed_encoder_info(pk, &sigma, &sigma_d, &split, nullptr);
  std::printf("packed %d tensors, %d layers: sigma %.3f sigma_d %.3f split_fp16 %d\n", (int)sd.size(), L, sigma, sigma_d, split);
  auto up = [&](const Named& t, float** d) {
    if (hipMalloc((void**)d, t.data.size() * 4) != hipSuccess) return false;
    return hipMemcpy(*d, t.data.data(), t.data.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
  };
  float *dcp, *dsrc, *dtgt, *dp, *dq, *dlog, *dfn, *dT, *dlab;
  if (!up(cp, &dcp) || !up(*by["input.src_keypts"], &dsrc) || !up(*by["input.tgt_keypts"], &dtgt) || !up(*by["input.p_tokens"], &dp) ||
      !up(*by["input.q_tokens"], &dq)) { std::printf("upload failed\n"); return 2; }
  CHECK_HIP(hipMalloc(&dlog, (size_t)B * N * 4)); CHECK_HIP(hipMalloc(&dfn, (size_t)B * N * 128 * 4));
  CHECK_HIP(hipMalloc(&dT, (size_t)B * 16 * 4)); CHECK_HIP(hipMalloc(&dlab, (size_t)B * N * 4));
  rc = gmf_encoder_forward(h, gmf_packed_encoder_weights(pk), dcp, dsrc, dtgt, dp, dq, B, N, T, dlog, dfn, nullptr, st);
  if (rc != GMF_OK) { std::printf("gmf_encoder_forward: status %d, %s\n", rc, gmf_last_error_string(h)); return 1; }
  gmf_pose_params pp;
  pp.num_seeds = (int)(N * 0.1); pp.k = 40 < N - 1 ? 40 : N - 1; pp.num_iterations = 10; pp.use_nms = 1; pp.refine_iters = 20;
  pp.sigma = sigma; pp.sigma_d = sigma_d; pp.inlier_threshold = 0.10f; pp.nms_radius = 0.10f; pp.refine_threshold = 0.10f;
  rc = gmf_pose_head(h, &pp, dfn, dsrc, dtgt, dlog, nullptr, B, N, dT, dlab, nullptr, nullptr, nullptr, nullptr, st);
  if (rc != GMF_OK) { std::printf("gmf_pose_head: status %d, %s\n", rc, gmf_last_error_string(h)); return 1; }
  CHECK_HIP(hipStreamSynchronize(st));
  int flags = -1;
  gmf_status_read(h, &flags, 1);
  std::vector<float> lg((size_t)B * N), Tm((size_t)B * 16);
  CHECK_HIP(hipMemcpy(lg.data(), dlog, lg.size() * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(Tm.data(), dT, Tm.size() * 4, hipMemcpyDeviceToHost));
  float el = 0.f, eT = 0.f;
  const Named& xl = *by["expect.logits"];
  const Named& xT = *by["expect.final_trans"];
  for (size_t i = 0; i < lg.size(); ++i) el = std::fmax(el, std::fabs(lg[i] - xl.data[i]));
  for (size_t i = 0; i < Tm.size(); ++i) eT = std::fmax(eT, std::fabs(Tm[i] - xT.data[i]));
  std::printf("hot path (B = %d, N = %d, T = %d): max |logit - reference| = %.3e, max |T - reference| = %.3e, status word %d\n", B, N, T, el, eT, flags);
  // the same pairs through the RAGGED entry points (host array of per-pair sizes, packed rows = the uniform layout when all
  // sizes are equal): logits and poses of the reference again
  std::vector<int> n_points((size_t)B, N);
  CHECK_HIP(hipMemset(dlog, 0, (size_t)B * N * 4));
  CHECK_HIP(hipMemset(dT, 0, (size_t)B * 16 * 4));
  rc = gmf_encoder_forward_ragged(h, gmf_packed_encoder_weights(pk), dcp, dsrc, dtgt, dp, dq, n_points.data(), B, T, dlog, dfn, nullptr, st);
  if (rc == GMF_OK) rc = gmf_pose_head_ragged(h, &pp, 0.1, dfn, dsrc, dtgt, dlog, n_points.data(), B, dT, dlab, nullptr, nullptr, nullptr, nullptr, st);
  if (rc != GMF_OK) { std::printf("ragged entry points: status %d, %s\n", rc, gmf_last_error_string(h)); return 1; }
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(lg.data(), dlog, lg.size() * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(Tm.data(), dT, Tm.size() * 4, hipMemcpyDeviceToHost));
  float el2 = 0.f, eT2 = 0.f;
  for (size_t i = 0; i < lg.size(); ++i) el2 = std::fmax(el2, std::fabs(lg[i] - xl.data[i]));
  for (size_t i = 0; i < Tm.size(); ++i) eT2 = std::fmax(eT2, std::fabs(Tm[i] - xT.data[i]));
  std::printf("ragged entry points, same pairs: max |logit - reference| = %.3e, max |T - reference| = %.3e\n", el2, eT2);
  // a caller-provided workspace that is too small is reported, not overrun
  void* small = nullptr;
  CHECK_HIP(hipMalloc(&small, 1 << 20));
  rc = gmf_set_workspace(h, small, 1 << 20);
  int rc2 = gmf_encoder_forward(h, gmf_packed_encoder_weights(pk), dcp, dsrc, dtgt, dp, dq, B, N, T, dlog, dfn, nullptr, st);
  std::printf("1 MiB caller workspace -> status %d (GMF_ERR_WORKSPACE = %d), wanted %lld bytes\n", rc2, (int)GMF_ERR_WORKSPACE, gmf_workspace_wanted(h));
  const bool ws_ok = rc == GMF_OK && rc2 == GMF_ERR_WORKSPACE && gmf_workspace_wanted(h) > (1 << 20);
  gmf_set_workspace(h, nullptr, 0);
  (void)hipFree(small);
  gmf_packed_encoder_free(pk);
  for (float* d : {dcp, dsrc, dtgt, dp, dq, dlog, dfn, dT, dlab}) (void)hipFree(d);
  return (el < 1e-4f && eT < 1e-4f && el2 < 1e-4f && eT2 < 1e-4f && flags == 0 && ws_ok) ? 0 : 1;
}

int main(int argc, char** argv) {
  gmf_handle* h = nullptr;
  if (gmf_create(0, &h) != GMF_OK || !h) { std::printf("gmf_create failed\n"); return 1; }
  if (gmf_abi_version() != GMF_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 1; }

  const int B = 4, N = 3000;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.f, 3.f), U01(0.f, 1.f);
  std::normal_distribution<float> G(0.f, 0.01f);
  std::vector<float> X((size_t)B * N * 3), Y((size_t)B * N * 3), W((size_t)B * N);
  std::vector<float> Rgt(B * 9), tgt(B * 3);
  std::vector<int> off(B + 1);
  for (int b = 0; b <= B; ++b) off[b] = b * N;
  for (int b = 0; b < B; ++b) {
    // rotation about a random axis (Rodrigues)
    float ax[3] = {U01(rng) - 0.5f, U01(rng) - 0.5f, U01(rng) - 0.5f};
    const float n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    for (float& a : ax) a /= n;
    const float th = 0.3f + U01(rng), c = std::cos(th), s = std::sin(th), C1 = 1.f - c;
    float* R = &Rgt[b * 9];
    R[0] = c + ax[0] * ax[0] * C1;         R[1] = ax[0] * ax[1] * C1 - ax[2] * s; R[2] = ax[0] * ax[2] * C1 + ax[1] * s;
    R[3] = ax[1] * ax[0] * C1 + ax[2] * s; R[4] = c + ax[1] * ax[1] * C1;         R[5] = ax[1] * ax[2] * C1 - ax[0] * s;
    R[6] = ax[2] * ax[0] * C1 - ax[1] * s; R[7] = ax[2] * ax[1] * C1 + ax[0] * s; R[8] = c + ax[2] * ax[2] * C1;
    for (int k = 0; k < 3; ++k) tgt[b * 3 + k] = U01(rng) - 0.5f;
    for (int i = 0; i < N; ++i) {
      float* x = &X[((size_t)b * N + i) * 3];
      float* y = &Y[((size_t)b * N + i) * 3];
      for (int k = 0; k < 3; ++k) x[k] = U(rng);
      const bool inlier = U01(rng) < 0.4f;
      for (int k = 0; k < 3; ++k)
        y[k] = inlier ? R[3 * k] * x[0] + R[3 * k + 1] * x[1] + R[3 * k + 2] * x[2] + tgt[b * 3 + k] + G(rng) : U(rng);
      W[(size_t)b * N + i] = inlier ? 0.6f + 0.4f * U01(rng) : (U01(rng) < 0.7f ? 0.f : 0.1f * U01(rng));
    }
  }
  float *dX, *dY, *dW, *dR, *dt, *dS;
  int* dOff;
  CHECK_HIP(hipMalloc(&dX, X.size() * 4)); CHECK_HIP(hipMalloc(&dY, Y.size() * 4)); CHECK_HIP(hipMalloc(&dW, W.size() * 4));
  CHECK_HIP(hipMalloc(&dR, B * 9 * 4)); CHECK_HIP(hipMalloc(&dt, B * 3 * 4)); CHECK_HIP(hipMalloc(&dS, B * 3 * 4));
  CHECK_HIP(hipMalloc(&dOff, (B + 1) * 4));
  CHECK_HIP(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dOff, off.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));

  auto max_err = [&](const char* what, float tolR, float tolt) {
    std::vector<float> R(B * 9), t(B * 3);
    if (hipMemcpy(R.data(), dR, B * 9 * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(t.data(), dt, B * 3 * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    float eR = 0, et = 0;
    for (int i = 0; i < B * 9; ++i) eR = std::fmax(eR, std::fabs(R[i] - Rgt[i]));
    for (int i = 0; i < B * 3; ++i) et = std::fmax(et, std::fabs(t[i] - tgt[i]));
    std::printf("%s: max |R - R_gt| = %.3e, max |t - t_gt| = %.3e\n", what, eR, et);
    return eR < tolR && et < tolt;
  };

  int rc = gmf_weighted_procrustes(h, dX, dY, dW, dOff, B, 1.1920929e-7f, dR, dt, st);
  CHECK_HIP(hipStreamSynchronize(st));
  if (rc != GMF_OK) { std::printf("weighted_procrustes: %s\n", gmf_last_error_string(h)); return 1; }
  if (!max_err("gmf_weighted_procrustes", 5e-2f, 1e-1f)) return 1;

  rc = gmf_global_registration(h, dX, dY, dW, dOff, B, 1.1920929e-7f, 0.1f, 1000, 20, 1e-4, dR, dt, dS, N, st);
  CHECK_HIP(hipStreamSynchronize(st));
  if (rc != GMF_OK) { std::printf("global_registration: %s\n", gmf_last_error_string(h)); return 1; }
  if (!max_err("gmf_global_registration", 5e-3f, 2e-2f)) return 1;
  std::vector<float> S(B * 3);
  CHECK_HIP(hipMemcpy(S.data(), dS, B * 3 * 4, hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b) std::printf("  pair %d: %d Adam steps, loss %.4f, break count %d\n", b, (int)S[3 * b], S[3 * b + 1], (int)S[3 * b + 2]);


  // validation step: one-hot unit features -> M[i][j] = 1 when the hot index matches (i != j), clamp(1 - 1/sigma^2) otherwise;
  // the fused spectral-matching loss equals the loss of the written matrix; a perfect pose scores 100 % recall
  {
    const int VB = 2, VN = 300, ldm = 320;
    const float sigma = 1.25f, off_val = 1.0f - 1.0f / (sigma * sigma);
    std::vector<float> F((size_t)VB * VN * 128, 0.f), gt((size_t)VB * VN), Mh((size_t)VB * VN * ldm);
    for (int b = 0; b < VB; ++b)
      for (int i = 0; i < VN; ++i) { F[((size_t)b * VN + i) * 128 + (i * 7 + b) % 128] = 1.f; gt[(size_t)b * VN + i] = (i % 3 == 0) ? 1.f : 0.f; }
    float *dF, *dG, *dM, *dL;
    CHECK_HIP(hipMalloc(&dF, F.size() * 4)); CHECK_HIP(hipMalloc(&dG, gt.size() * 4)); CHECK_HIP(hipMalloc(&dM, Mh.size() * 4));
    CHECK_HIP(hipMalloc(&dL, 16 * 4));
    CHECK_HIP(hipMemcpy(dF, F.data(), F.size() * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dG, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
    rc = gmf_similarity_matrix(h, dF, VB, VN, sigma, dM, ldm, st);
    if (rc == GMF_OK) rc = gmf_spectral_matching_loss(h, dM, ldm, dG, VB, VN, 1, dL, st);
    if (rc == GMF_OK) rc = gmf_spectral_matching_loss_fused(h, dF, dG, VB, VN, sigma, 1, dL + 1, st);
    CHECK_HIP(hipStreamSynchronize(st));
    if (rc != GMF_OK) { std::printf("validation step: %s\n", gmf_last_error_string(h)); return 1; }
    CHECK_HIP(hipMemcpy(Mh.data(), dM, Mh.size() * 4, hipMemcpyDeviceToHost));
    float eM = 0.f;
    for (int b = 0; b < VB; ++b)
      for (int i = 0; i < VN; ++i)
        for (int j = 0; j < VN; ++j) {
          const float want = i == j ? 0.f : (((i * 7 + b) % 128 == (j * 7 + b) % 128) ? 1.f : off_val);
          eM = std::fmax(eM, std::fabs(Mh[((size_t)b * VN + i) * ldm + j] - want));
        }
    float L[2];
    CHECK_HIP(hipMemcpy(L, dL, 8, hipMemcpyDeviceToHost));
    std::printf("gmf_similarity_matrix: max |M - expected| = %.3e; spectral matching loss %.6f (from M) %.6f (fused)\n", eM, L[0], L[1]);
    if (eM > 1e-6f || std::fabs(L[0] - L[1]) > 1e-6f) return 1;
  }

  // error path: a null pointer must come back as a status code with a message, never as an abort
  rc = gmf_weighted_procrustes(h, nullptr, dY, dW, dOff, B, 1.1920929e-7f, dR, dt, st);
  if (rc == GMF_OK) { std::printf("null pointer was accepted\n"); return 1; }
  std::printf("bad call -> status %d, \"%s\"\n", rc, gmf_last_error_string(h));

  if (argc > 1) {
    if (int r = run_hot_path(h, argv[1], st)) return r;
    std::printf("hot path OK\n");
  }

  gmf_destroy(h);
  std::printf("ABI host OK\n");
  return 0;
}
