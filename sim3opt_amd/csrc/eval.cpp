// eval.cpp -- trajectory evaluation harness (host C++).
//
// Restates the reference's only quantitative quality metric:
//   estimateSimilarityTransform (Eigen::umeyama, with scaling) ... kitti_surf.cpp:1091-1161
//   RMSE / max deviation of aligned keyframe positions vs KITTI GT .. kitti_surf.cpp:1427-1463
// Umeyama 1991: R = U S V^T from the SVD of the cross-covariance, c = tr(D S)/var(x),
// t = mu_y - c R mu_x.  The 3x3 SVD is a cyclic Jacobi eigen-decomposition (no Eigen here).
#include <cmath>
#include <cstring>

#include "../../include/sim3opt.h"

namespace {

// eigen-decomposition of a symmetric 3x3 matrix: A = V diag(w) V^T, w descending
void eig3(const double Ain[9], double V[9], double w[3]) {
  double A[9];
  std::memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        const double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A J
          const double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- J^T A
          const double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
  for (int i = 0; i < 2; ++i)  // sort descending, permuting the columns of V
    for (int j = 0; j < 2 - i; ++j)
      if (w[j] < w[j + 1]) {
        const double tw = w[j]; w[j] = w[j + 1]; w[j + 1] = tw;
        for (int k = 0; k < 3; ++k) { const double tv = V[3 * k + j]; V[3 * k + j] = V[3 * k + j + 1]; V[3 * k + j + 1] = tv; }
      }
}

double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

}  // namespace

extern "C" int sim3opt_align_trajectory(int32_t n, const double* query_xyz, const double* train_xyz,
                                        int32_t with_scale, double S[16], double* rmse,
                                        double* max_dev) {
  if (n < 3 || !query_xyz || !train_xyz || !S) return SIM3OPT_ERR_ARG;
  double mx[3] = {0, 0, 0}, my[3] = {0, 0, 0};
  for (int32_t k = 0; k < n; ++k)
    for (int i = 0; i < 3; ++i) { mx[i] += query_xyz[3 * k + i]; my[i] += train_xyz[3 * k + i]; }
  for (int i = 0; i < 3; ++i) { mx[i] /= n; my[i] /= n; }
  double C[9] = {0}, varx = 0;  // C = 1/n sum (y - my)(x - mx)^T
  for (int32_t k = 0; k < n; ++k) {
    double dx[3], dy[3];
    for (int i = 0; i < 3; ++i) { dx[i] = query_xyz[3 * k + i] - mx[i]; dy[i] = train_xyz[3 * k + i] - my[i]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] += dy[i] * dx[j];
    varx += dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
  }
  for (double& c : C) c /= n;
  varx /= n;
  if (!(varx > 0)) return SIM3OPT_ERR_ARG;
  // SVD of C through the eigen-decomposition of C^T C = V D^2 V^T; U = C V D^-1
  double CtC[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double a = 0;
    for (int k = 0; k < 3; ++k) a += C[3 * k + i] * C[3 * k + j];
    CtC[3 * i + j] = a;
  }
  double V[9], w[3], d[3], U[9];
  eig3(CtC, V, w);
  for (int j = 0; j < 3; ++j) d[j] = std::sqrt(w[j] > 0 ? w[j] : 0);
  for (int j = 0; j < 2; ++j) {
    if (!(d[j] > 1e-14 * (d[0] + 1e-300))) return SIM3OPT_ERR_ARG;  // points on a line
    for (int i = 0; i < 3; ++i) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += C[3 * i + k] * V[3 * k + j];
      U[3 * i + j] = a / d[j];
    }
  }
  {  // third left vector: cross product (defined even when d[2] = 0, coplanar points)
    const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    double u2[3];
    cross3(u0, u1, u2);
    double sgn = 1.0;
    if (d[2] > 1e-14 * d[0]) {  // keep the orientation of C v2
      double cv[3] = {0, 0, 0};
      for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) cv[i] += C[3 * i + k] * V[3 * k + 2];
      if (cv[0] * u2[0] + cv[1] * u2[1] + cv[2] * u2[2] < 0) sgn = -1.0;
    }
    U[2] = sgn * u2[0]; U[5] = sgn * u2[1]; U[8] = sgn * u2[2];
  }
  const double s3 = det3(U) * det3(V) < 0 ? -1.0 : 1.0;
  double R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + s3 * U[3 * i + 2] * V[3 * j + 2];
  const double c = with_scale ? (d[0] + d[1] + s3 * d[2]) / varx : 1.0;
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = my[i] - c * (R[3 * i] * mx[0] + R[3 * i + 1] * mx[1] + R[3 * i + 2] * mx[2]);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) S[4 * i + j] = c * R[3 * i + j];
    S[4 * i + 3] = t[i];
  }
  S[12] = S[13] = S[14] = 0; S[15] = 1;
  double tot = 0, mxd = 0;  // kitti_surf.cpp:1446-1457
  for (int32_t k = 0; k < n; ++k) {
    double dev2 = 0;
    for (int i = 0; i < 3; ++i) {
      const double a = S[4 * i] * query_xyz[3 * k] + S[4 * i + 1] * query_xyz[3 * k + 1] +
                       S[4 * i + 2] * query_xyz[3 * k + 2] + S[4 * i + 3];
      const double e = train_xyz[3 * k + i] - a;
      dev2 += e * e;
    }
    tot += dev2;
    if (std::sqrt(dev2) > mxd) mxd = std::sqrt(dev2);
  }
  if (rmse) *rmse = std::sqrt(tot / n);
  if (max_dev) *max_dev = mxd;
  return SIM3OPT_OK;
}
