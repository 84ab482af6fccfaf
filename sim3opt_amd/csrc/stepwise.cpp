// stepwise.cpp -- stage 1 of the reference's stepwise optimisation.
//
// testStepwiseSim3Optimization, "scale_dlt" (kitti_surf.cpp:887-933): every edge (v0, v1, C) gives
// one homogeneous equation  s_C * x[v0] - x[v1] = 0  on the vertex scales (odometry rows have
// s_C = 1, loop rows the measured scale ratio); the reference takes the last right-singular vector
// of the dense (#edges x #vertices) matrix with Eigen::JacobiSVD, divides by its first entry and
// writes the result into the scale of every vertex estimate (rotation and translation untouched).
// Here: the same vector as the eigenvector of A^T A for its smallest eigenvalue, by inverse
// iteration on a dense Cholesky factor (host code; A^T A is a weighted graph Laplacian).
// Stages 2 and 3 are LM runs on the GPU: sim3opt_options.dof_mask = 0x78 (rotations frozen) and 127.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/sim3opt.h"

extern "C" int sim3opt_stepwise_scale_init(sim3opt_graph* g, double* sigma_ratio) {
  if (!g) return SIM3OPT_ERR_ARG;
  const int32_t n = sim3opt_num_vertices(g), m = sim3opt_num_edges(g);
  if (n < 2 || m < 1 || n > 4096) return SIM3OPT_ERR_ARG;  // dense stage: small graphs only
  std::vector<double> st(8 * (size_t)n);
  int rc = sim3opt_get_vertices(g, st.data());
  if (rc != SIM3OPT_OK) return rc;
  // vertex ids must be the dense 0..n-1 the reference uses (kitti_surf.cpp:604, :618): id == index
  std::vector<double> M((size_t)n * n, 0.0);  // A^T A (symmetric)
  double trace = 0.0;
  for (int32_t k = 0; k < m; ++k) {
    int32_t a, b;
    double meas[8];
    rc = sim3opt_get_edge(g, k, &a, &b, meas);
    if (rc != SIM3OPT_OK) return rc;
    if (a < 0 || a >= n || b < 0 || b >= n) return SIM3OPT_ERR_ARG;  // needs dense ids 0..n-1
    const double s = meas[7];
    M[(size_t)a * n + a] += s * s;
    M[(size_t)b * n + b] += 1.0;
    M[(size_t)a * n + b] -= s;
    M[(size_t)b * n + a] -= s;
    trace += s * s + 1.0;
  }
  // Cholesky of M + mu I (lower, in place)
  const double mu = 1e-13 * trace / n + 1e-300;
  std::vector<double> Lm(M);
  for (int32_t j = 0; j < n; ++j) Lm[(size_t)j * n + j] += mu;
  for (int32_t j = 0; j < n; ++j) {
    double d = Lm[(size_t)j * n + j];
    for (int32_t k = 0; k < j; ++k) d -= Lm[(size_t)k * n + j] * Lm[(size_t)k * n + j];
    if (!(d > 0.0)) return SIM3OPT_ERR_STATE;
    d = std::sqrt(d);
    Lm[(size_t)j * n + j] = d;
    for (int32_t i = j + 1; i < n; ++i) {
      double v = Lm[(size_t)j * n + i];
      for (int32_t k = 0; k < j; ++k) v -= Lm[(size_t)k * n + i] * Lm[(size_t)k * n + j];
      Lm[(size_t)j * n + i] = v / d;  // L(i, j) stored at column j, row i
    }
  }
  std::vector<double> x(n, 1.0), y(n);
  double lam_min = 0.0;
  for (int it = 0; it < 500; ++it) {
    // solve L L^T y = x
    for (int32_t i = 0; i < n; ++i) {
      double v = x[i];
      for (int32_t k = 0; k < i; ++k) v -= Lm[(size_t)k * n + i] * y[k];
      y[i] = v / Lm[(size_t)i * n + i];
    }
    for (int32_t i = n - 1; i >= 0; --i) {
      double v = y[i];
      for (int32_t k = i + 1; k < n; ++k) v -= Lm[(size_t)i * n + k] * y[k];
      y[i] = v / Lm[(size_t)i * n + i];
    }
    double nrm = 0.0, dot = 0.0;
    for (int32_t i = 0; i < n; ++i) nrm += y[i] * y[i];
    nrm = std::sqrt(nrm);
    double diff = 0.0;
    for (int32_t i = 0; i < n; ++i) {
      const double v = y[i] / nrm;
      dot += v * x[i];
      diff = std::fmax(diff, std::fabs(std::fabs(v) - std::fabs(x[i])));
      x[i] = v;
    }
    lam_min = 1.0 / nrm - mu;  // Rayleigh estimate of the smallest eigenvalue (x was unit length)
    if (it > 0 && diff < 1e-16) break;
    (void)dot;
  }
  if (sigma_ratio) {  // sigma_min / sigma_max estimate (the reference warns below 5e-4)
    double mx = 0.0;
    for (int32_t j = 0; j < n; ++j) mx = std::fmax(mx, M[(size_t)j * n + j]);
    *sigma_ratio = std::sqrt(std::fmax(lam_min, 0.0) / (2.0 * mx));
  }
  if (x[0] == 0.0) return SIM3OPT_ERR_STATE;
  for (int32_t i = 0; i < n; ++i) {
    const double s = x[i] / x[0];  // allScales / allScales[0]
    if (!(s > 0.0)) return SIM3OPT_ERR_STATE;
    st[8 * (size_t)i + 7] = s;
  }
  return sim3opt_set_vertices(g, st.data());
}
