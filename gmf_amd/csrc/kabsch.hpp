// 3x3 SVD -> Kabsch rotation in fp64 registers, and its derivative.  Shared by the pose-head kernels (pose_kernels.hip) and
// their backward (pose_backward.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace gmf {

#ifndef GMF_DEVINL
#define GMF_DEVINL __device__ __forceinline__
#endif

GMF_DEVINL void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

GMF_DEVINL void any_perp(const double* a, double* p) {
  // unit vector perpendicular to unit a
  double ax = fabs(a[0]), ay = fabs(a[1]), az = fabs(a[2]);
  double e[3] = {0, 0, 0};
  if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
  cross3(a, e, p);
  const double n = rsqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  p[0] *= n; p[1] *= n; p[2] *= n;
}

// ---------------------------------------------------------------------------------------
// H (row-major 3x3) = U' diag(sig) V'^T with U' = [u1 u2 u1 x u2], V' = [v1 v2 v1 x v2] proper rotations over the two
// dominant singular pairs (one-sided Jacobi) and sig[2] = (u1 x u2)^T H (v1 x v2) SIGNED.  ok = false: H == 0 (or NaN).
// ---------------------------------------------------------------------------------------
struct KabschFrames {
  double u[3][3], v[3][3];      // u[i] = i-th column of U' as a vector
  double sig[3];
  bool ok;
};

GMF_DEVINL void kabsch_frames(const double* Hin, KabschFrames& f) {
  double A[3][3], V[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { A[r][c] = Hin[3 * r + c]; V[r][c] = (r == c) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
      double al = 0, be = 0, ga = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) { al += A[r][p] * A[r][p]; be += A[r][q] * A[r][q]; ga += A[r][p] * A[r][q]; }
      const double lim = 1e-15 * sqrt(al * be);
      if (fabs(ga) > lim && ga != 0.0) {
        off = fmax(off, fabs(ga) / fmax(sqrt(al * be), 1e-300));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = rsqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double ap = A[r][p], aq = A[r][q];
          A[r][p] = cs * ap - sn * aq; A[r][q] = sn * ap + cs * aq;
          const double vp = V[r][p], vq = V[r][q];
          V[r][p] = cs * vp - sn * vq; V[r][q] = sn * vp + cs * vq;
        }
      }
    }
    if (off < 1e-15) break;
  }
  double sg[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) sg[c] = sqrt(A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c]);
  int j1 = 0;
  if (sg[1] > sg[j1]) j1 = 1;
  if (sg[2] > sg[j1]) j1 = 2;
  int j2 = (j1 == 0) ? 1 : 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) if (c != j1 && sg[c] > sg[j2]) j2 = c;
  f.ok = sg[j1] > 0.0;
  if (!f.ok) return;
#pragma unroll
  for (int r = 0; r < 3; ++r) { f.u[0][r] = A[r][j1] / sg[j1]; f.v[0][r] = V[r][j1]; }
  f.sig[0] = sg[j1];
  if (sg[j2] > 1e-14 * sg[j1]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) { f.u[1][r] = A[r][j2] / sg[j2]; f.v[1][r] = V[r][j2]; }
    f.sig[1] = sg[j2];
  } else {                           // rank 1: any completion (LAPACK's choice is arbitrary too)
    any_perp(f.u[0], f.u[1]);
    any_perp(f.v[0], f.v[1]);
    f.sig[1] = 0.0;
  }
  cross3(f.u[0], f.u[1], f.u[2]);
  cross3(f.v[0], f.v[1], f.v[2]);
  double s3 = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) s3 += f.u[2][r] * Hin[3 * r + c] * f.v[2][c];
  f.sig[2] = s3;
}

// ---------------------------------------------------------------------------------------
// 3x3 SVD -> rotation.  H = U S V^T.  Returns  R = V diag(1,1,det(V U^T)) U^T  (Kabsch, common.py:43-45)
// in the determinant-free form  R = v1 u1^T + v2 u2^T + (v1 x v2)(u1 x u2)^T  over the two dominant
// singular pairs, which equals the reference's formula for any sign convention of the SVD and stays
// well defined when the smallest singular value is 0 (planar neighbourhoods).
// ---------------------------------------------------------------------------------------
GMF_DEVINL void kabsch_rotation_from_H(const double* Hin, double* R) {
  KabschFrames f;
  kabsch_frames(Hin, f);
  if (!f.ok) {                       // H == 0 (or NaN): identity
#pragma unroll
    for (int r = 0; r < 9; ++r) R[r] = (r % 4 == 0) ? 1.0 : 0.0;
    return;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[3 * r + c] = f.v[0][r] * f.u[0][c] + f.v[1][r] * f.u[1][c] + f.v[2][r] * f.u[2][c];
}

// ---------------------------------------------------------------------------------------
// Derivative of R = kabsch_rotation_from_H(H): given gR = dL/dR returns gH = dL/dH.
// With the signed decomposition above R = V' U'^T, and for dH:  dR = V' X U'^T,  X_ij = (H'_ji - H'_ij)/(sig_i + sig_j),
// H' = U'^T dH V'  (differential of the SVD; no singularity at equal singular values, only where sig_i + sig_j = 0 -
// the reflection boundary, where torch's svd backward is unbounded too; such terms are dropped).  Hence
//   gH = U' G V'^T,   G_pq = (Y_qp - Y_pq)/(sig_p + sig_q)  (p != q),   Y = V'^T gR U'.
// R and (if asked) the frames' sig come back too.
// ---------------------------------------------------------------------------------------
GMF_DEVINL void kabsch_backward(const double* Hin, const double* gR, double* gH) {
  KabschFrames f;
  kabsch_frames(Hin, f);
#pragma unroll
  for (int e = 0; e < 9; ++e) gH[e] = 0.0;
  if (!f.ok) return;
  double Y[3][3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double y = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) y += f.v[p][r] * gR[3 * r + c] * f.u[q][c];
      Y[p][q] = y;
    }
  double G[3][3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double den = f.sig[p] + f.sig[q];
      G[p][q] = (p != q && fabs(den) > 1e-14 * f.sig[0]) ? (Y[q][p] - Y[p][q]) / den : 0.0;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double g = 0.0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) g += f.u[p][r] * G[p][q] * f.v[q][c];
      gH[3 * r + c] = g;
    }
}

}  // namespace gmf
