// sim3_math.hpp -- Sim(3) group arithmetic for the MI355X path (host + device).
//
// Replaces g2o::Sim3 (g2o/types/sim3/sim3.h @ 8564e1e, not in tree; call sites
// kitti_surf.cpp:33, 200, 533-538, 593-594, 608, 653-660, 686-698) for the
// device-resident LM loop.  Formula authority: sim3_rv.h:125-190 (exp),
// :242-320 (ln), :199-220 (inverse/compose); tangent order is g2o's
// [omega(0:3), upsilon(3:6), sigma(6)].
//
// Everything is plain FP64 scalar code on fixed-size arrays so that the same
// source compiles for gfx950 and for the host (used by the C-ABI get/set path
// and the graph builders).  No Eigen/Sophus: neither exists in this image.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define S3_HD __host__ __device__ __forceinline__
#else
#define S3_HD inline
#endif

// The reference differentiates numerically with delta = 1e-9 (g2o BaseBinaryEdge): a last-bit
// difference in a residual becomes 1e-7 in a Jacobian entry, and the as-written B coefficient
// (w_coeffs below) amplifies that further.  To stay as close to a CPU evaluation of the same formulae
// as the hardware allows, every function here (a) performs its operations in the order a plain
// C / Eigen statement of sim3_rv.h performs them (matrix products for Omega^2, partially pivoted LU
// for W^-1 t, like TooN::LU at sim3_rv.h:305-307 and Eigen's lu().solve in g2o) and (b) forbids
// fusing a*b+c into one rounding (x86-64 code of the reference's era has no FMA).  What is left
// between this code on gfx950 and the CPU oracle is the last bit of sin/cos/acos/exp/log.
#if defined(__clang__)
#define S3_STRICT_FP _Pragma("clang fp contract(off)")
#else
#define S3_STRICT_FP
#endif

namespace sim3 {

// 8 doubles = 64 bytes = half a 128-B HBM line; vertex states and edge
// measurements are stored as arrays of this struct (AoS: one gather = one
// contiguous 64-B read).
struct alignas(64) Sim3 {
  double q[4];  // x, y, z, w  (Eigen coeffs() order, kitti_surf.cpp:698)
  double t[3];
  double s;
};

struct Opts {
  double eps;          // 1e-5, branch threshold (sim3_rv.h:133, :258)
  int small_rot_half;  // 0: R = I + W + W^2 (sim3_rv.h:151); 1: I + W + W^2/2 (later g2o)
  int fix_small_b;     // 0: B as written (sim3_rv.h:166, :290); 1: exact small-theta limit
};

S3_HD void quat_from_R(const double R[9], double q[4]) {
  S3_STRICT_FP
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else if (R[0] >= R[4] && R[0] >= R[8]) {
    double t = sqrt(R[0] - R[4] - R[8] + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[7] - R[5]) * t;
    q[1] = (R[3] + R[1]) * t;
    q[2] = (R[6] + R[2]) * t;
  } else if (R[4] > R[0] && R[4] >= R[8]) {
    double t = sqrt(R[4] - R[8] - R[0] + 1.0);
    q[1] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[2] - R[6]) * t;
    q[2] = (R[7] + R[5]) * t;
    q[0] = (R[1] + R[3]) * t;
  } else {
    double t = sqrt(R[8] - R[0] - R[4] + 1.0);
    q[2] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[3] - R[1]) * t;
    q[0] = (R[2] + R[6]) * t;
    q[1] = (R[5] + R[7]) * t;
  }
}

S3_HD void R_from_quat(const double q[4], double R[9]) {
  S3_STRICT_FP
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

S3_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
  S3_STRICT_FP
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

S3_HD void quat_rot(const double q[4], const double v[3], double o[3]) {
  S3_STRICT_FP
  const double ux = 2 * (q[1] * v[2] - q[2] * v[1]);
  const double uy = 2 * (q[2] * v[0] - q[0] * v[2]);
  const double uz = 2 * (q[0] * v[1] - q[1] * v[0]);
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

// a * b : x -> a(b(x))                                   (sim3_rv.h:214-220)
S3_HD Sim3 mul(const Sim3& a, const Sim3& b) {
  S3_STRICT_FP
  Sim3 r;
  double rt[3];
  quat_mul(a.q, b.q, r.q);
  quat_rot(a.q, b.t, rt);
  r.t[0] = a.s * rt[0] + a.t[0];
  r.t[1] = a.s * rt[1] + a.t[1];
  r.t[2] = a.s * rt[2] + a.t[2];
  r.s = a.s * b.s;
  return r;
}

S3_HD Sim3 inverse(const Sim3& a) {                     // sim3_rv.h:199-203
  S3_STRICT_FP
  Sim3 r;
  r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
  const double k = -1.0 / a.s;
  const double tmp[3] = {k * a.t[0], k * a.t[1], k * a.t[2]};
  quat_rot(r.q, tmp, r.t);
  r.s = 1.0 / a.s;
  return r;
}

// A, B, C of W = A*Omega + B*Omega^2 + C*I            (sim3_rv.h:143-181, :261-303)
S3_HD void w_coeffs(double sigma, double s, double theta, bool small_theta, double eps, int fixb,
                    double& A, double& B, double& C) {
  S3_STRICT_FP
  if (fabs(sigma) < eps) {
    C = 1.0;
    if (small_theta) {
      A = 0.5;
      B = 1.0 / 6.0;
    } else {
      const double th2 = theta * theta;
      A = (1 - cos(theta)) / th2;
      B = (theta - sin(theta)) / (th2 * theta);
    }
  } else {
    C = (s - 1) / sigma;
    if (small_theta) {
      const double sg2 = sigma * sigma;
      A = ((sigma - 1) * s + 1) / sg2;
      // as written in sim3_rv.h:166 / :290 this is NOT the small-theta limit (it behaves like
      // 1/sigma^3); log() reaches it for theta < 4.5e-3.  Kept by default for parity with the
      // reference; fixb selects the exact limit ((sigma^2/2 - sigma + 1) s - 1) / sigma^3.
      B = ((0.5 * sg2 - sigma + 1) * s - (fixb ? 1.0 : 0.0)) / (sg2 * sigma);
    } else {
      const double a = s * sin(theta), b = s * cos(theta);
      const double th2 = theta * theta, c = th2 + sigma * sigma;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / c) * 1.0 / th2;
    }
  }
}

// Omega = [w]x (sim3_rv.h:38-50) and Omega^2 by the matrix product, row-major
S3_HD void skew_and_square(const double w[3], double Om[9], double Om2[9]) {
  S3_STRICT_FP
  Om[0] = 0;     Om[1] = -w[2]; Om[2] = w[1];
  Om[3] = w[2];  Om[4] = 0;     Om[5] = -w[0];
  Om[6] = -w[1]; Om[7] = w[0];  Om[8] = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += Om[3 * i + k] * Om[3 * k + j];
      Om2[3 * i + j] = acc;
    }
}

// Sim3(Vector7) of g2o                                    (sim3_rv.h:125-190)
S3_HD Sim3 exp(const double xi[7], const Opts& o) {
  S3_STRICT_FP
  const double* om = xi;
  const double sigma = xi[6];
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  double Om[9], Om2[9], R[9], W[9];
  skew_and_square(om, Om, Om2);
  const double s = ::exp(sigma);
  const bool small = theta < o.eps;
  double A, B, C;
  w_coeffs(sigma, s, theta, small, o.eps, o.fix_small_b, A, B, C);
  if (small) {
    const double h = o.small_rot_half ? 0.5 : 1.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Om[i] + h * Om2[i];
  } else {
    const double k1 = sin(theta) / theta, k2 = (1 - cos(theta)) / (theta * theta);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = k1 * Om[i] + k2 * Om2[i];
  }
  R[0] += 1; R[4] += 1; R[8] += 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = A * Om[i] + B * Om2[i];
  W[0] += C; W[4] += C; W[8] += C;
  Sim3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i) r.t[i] = W[3 * i] * xi[3] + W[3 * i + 1] * xi[4] + W[3 * i + 2] * xi[5];
  quat_from_R(R, r.q);
  r.s = s;
  return r;
}

// W x = t by LU with partial pivoting (TooN::LU, sim3_rv.h:305-307; Eigen lu().solve in g2o).  Where
// the as-written B makes W nearly singular the pivoting order decides the rounding, so it is the
// reference's, not an adjugate.  (Row swaps by selects: no dynamically indexed registers.)
S3_HD void solve33(const double Win[9], const double tin[3], double x[3]) {
  S3_STRICT_FP
  double M[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i][j] = Win[3 * i + j];
    M[i][3] = tin[i];
  }
  auto swap_rows = [&](int a, int b, bool sw) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double u = M[a][j], v = M[b][j];
      M[a][j] = sw ? v : u;
      M[b][j] = sw ? u : v;
    }
  };
  auto eliminate = [&](int c) {
#pragma unroll
    for (int r = c + 1; r < 3; ++r) {
      const double f = M[r][c] / M[c][c];
#pragma unroll
      for (int j = c; j < 4; ++j) M[r][j] -= f * M[c][j];
    }
  };
  {  // column 0: pivot = first row of maximal |.|, swapped with row 0
    const bool p1 = fabs(M[1][0]) > fabs(M[0][0]);
    const double best = p1 ? fabs(M[1][0]) : fabs(M[0][0]);
    const bool p2 = fabs(M[2][0]) > best;
    swap_rows(0, 1, p1 && !p2);
    swap_rows(0, 2, p2);
    eliminate(0);
  }
  swap_rows(1, 2, fabs(M[2][1]) > fabs(M[1][1]));
  eliminate(1);
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    double acc = M[i][3];
#pragma unroll
    for (int j = i + 1; j < 3; ++j) acc -= M[i][j] * x[j];
    x[i] = acc / M[i][i];
  }
}

// Sim3::log()                                              (sim3_rv.h:242-320)
S3_HD void log(const Sim3& S, const Opts& o, double xi[7]) {
  S3_STRICT_FP
  const double s = S.s, sigma = ::log(s);
  double R[9];
  R_from_quat(S.q, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};  // sim3_rv.h:51-54
  const bool small = d > 1 - o.eps;
  double theta = 0.0, k = 0.5;
  if (!small) {
    theta = acos(d);
    k = theta / (2 * sqrt(1 - d * d));
  }
  const double om[3] = {k * dR[0], k * dR[1], k * dR[2]};
  double A, B, C;
  w_coeffs(sigma, s, theta, small, o.eps, o.fix_small_b, A, B, C);
  double Om[9], Om2[9], W[9], up[3];
  skew_and_square(om, Om, Om2);
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = A * Om[i] + B * Om2[i];
  W[0] += C; W[4] += C; W[8] += C;
  solve33(W, S.t, up);
  xi[0] = om[0]; xi[1] = om[1]; xi[2] = om[2];
  xi[3] = up[0]; xi[4] = up[1]; xi[5] = up[2];
  xi[6] = sigma;
}

// EdgeSim3::computeError: e = log(C * S0 * S1^-1)  (edges set up at kitti_surf.cpp:633-638, :663-668)
S3_HD void edge_error(const Sim3& C, const Sim3& S0, const Sim3& S1, const Opts& o, double e[7]) {
  const Sim3 S1i = inverse(S1);
  const Sim3 E = mul(mul(C, S0), S1i);
  log(E, o, e);
}

}  // namespace sim3
