// sim3opt_g2o.hpp -- header-only C++ shim with the g2o operator surface the reference uses,
// forwarding to the C-ABI of libsim3opt (include/sim3opt.h).
//
// It lets the graph-building code of testDirectSim3Optimization / testStepwiseSim3Optimization
// (kitti_surf.cpp:552-558, 597-675, 681-701, 1028-1075) keep its shape:
//
//     g2o::SparseOptimizer optimizer;                                   // :552
//     optimizer.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(   // :556-558
//         g2o::make_unique<g2o::BlockSolverX>(std::move(linearSolver))));
//     vio::VertexSim3Expmap* v = new vio::VertexSim3Expmap();           // :602
//     v->setEstimate(g2o::Sim3(Rcw, tcw, 1.0)); v->setFixed(i == 0); v->setId(i);
//     optimizer.addVertex(v);                                           // :611-620
//     vio::EdgeSim3* e = new vio::EdgeSim3();                           // :633
//     e->setVertex(1, optimizer.vertex(j)); e->setVertex(0, optimizer.vertex(i));
//     e->setMeasurement(Sji); e->information() = I7; optimizer.addEdge(e);
//     optimizer.initializeOptimization(); optimizer.optimize(100);      // :674-675
//     g2o::Sim3 S = static_cast<vio::VertexSim3Expmap*>(optimizer.vertex(i))->estimate();   // :688-689
//
// The core below depends on nothing but <array>/<map>/<memory>.  Where Eigen is installed
// (the reference's build has it; this image does not, so that part is not compile-tested here)
// the section at the bottom adds the Eigen-typed constructors and accessors the reference calls
// (Sim3(Matrix3d, Vector3d, double), rotation(), translation(), information() = Matrix7d).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "sim3opt.h"

namespace sim3opt_shim {

// g2o::Sim3 -- unit quaternion (x,y,z,w), translation, scale                (sim3_rv.h:71-226)
struct Sim3 {
  std::array<double, 8> v{{0, 0, 0, 1, 0, 0, 0, 1}};
  Sim3() = default;
  explicit Sim3(const double state[8]) { for (int i = 0; i < 8; ++i) v[i] = state[i]; }
  // from a row-major 3x3 rotation, translation and scale (g2o: Sim3(Matrix3, Vector3, double))
  Sim3(const double R[9], const double t[3], double s) {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      double k = std::sqrt(tr + 1.0);
      v[3] = 0.5 * k; k = 0.5 / k;
      v[0] = (R[7] - R[5]) * k; v[1] = (R[2] - R[6]) * k; v[2] = (R[3] - R[1]) * k;
    } else {
      int i = 0;
      if (R[4] > R[0]) i = 1;
      if (R[8] > R[4 * i]) i = 2;
      const int j = (i + 1) % 3, k3 = (j + 1) % 3;
      double k = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k3] + 1.0);
      v[i] = 0.5 * k; k = 0.5 / k;
      v[3] = (R[3 * k3 + j] - R[3 * j + k3]) * k;
      v[j] = (R[3 * j + i] + R[3 * i + j]) * k;
      v[k3] = (R[3 * k3 + i] + R[3 * i + k3]) * k;
    }
    v[4] = t[0]; v[5] = t[1]; v[6] = t[2]; v[7] = s;
  }
  double scale() const { return v[7]; }
  const double* quaternion_xyzw() const { return v.data(); }
  const double* translation_ptr() const { return v.data() + 4; }
  static void rot(const double* q, const double* p, double* o) {
    const double ux = 2 * (q[1] * p[2] - q[2] * p[1]), uy = 2 * (q[2] * p[0] - q[0] * p[2]),
                 uz = 2 * (q[0] * p[1] - q[1] * p[0]);
    o[0] = p[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = p[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = p[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
  }
  Sim3 inverse() const {                                                    // sim3_rv.h:199-203
    Sim3 r;
    r.v[0] = -v[0]; r.v[1] = -v[1]; r.v[2] = -v[2]; r.v[3] = v[3];
    const double k = -1.0 / v[7];
    const double tmp[3] = {k * v[4], k * v[5], k * v[6]};
    rot(r.v.data(), tmp, r.v.data() + 4);
    r.v[7] = 1.0 / v[7];
    return r;
  }
  Sim3 operator*(const Sim3& b) const {                                     // sim3_rv.h:214-220
    Sim3 r;
    const double *a = v.data(), *c = b.v.data();
    r.v[0] = a[3] * c[0] + a[0] * c[3] + a[1] * c[2] - a[2] * c[1];
    r.v[1] = a[3] * c[1] + a[1] * c[3] + a[2] * c[0] - a[0] * c[2];
    r.v[2] = a[3] * c[2] + a[2] * c[3] + a[0] * c[1] - a[1] * c[0];
    r.v[3] = a[3] * c[3] - a[0] * c[0] - a[1] * c[1] - a[2] * c[2];
    double rt[3];
    rot(a, c + 4, rt);
    for (int i = 0; i < 3; ++i) r.v[4 + i] = a[7] * rt[i] + a[4 + i];
    r.v[7] = a[7] * c[7];
    return r;
  }
};

class SparseOptimizer;

class Vertex {  // g2o::OptimizableGraph::Vertex, as far as the reference touches it
 public:
  virtual ~Vertex() = default;
  void setId(int id) { id_ = id; }
  int id() const { return id_; }
  void setFixed(bool f) { fixed_ = f; }
  bool fixed() const { return fixed_; }
  void setMarginalized(bool) {}  // kitti_surf.cpp:619 always passes false; nothing is marginalised
 protected:
  friend class SparseOptimizer;
  int id_ = -1;
  bool fixed_ = false;
  SparseOptimizer* owner_ = nullptr;
};

class VertexSim3Expmap : public Vertex {  // vio::VertexSim3Expmap (kitti_surf.cpp:602-620)
 public:
  void setEstimate(const Sim3& s);
  Sim3 estimate() const;
 private:
  Sim3 est_;
};

class EdgeSim3 {  // vio::EdgeSim3 (kitti_surf.cpp:633-638, :663-668)
 public:
  void setVertex(int slot, Vertex* v) { v_[slot] = v; }
  void setMeasurement(const Sim3& m) { meas_ = m; }
  std::array<double, 49>& information() { has_info_ = true; return info_; }  // column-major
  void setRobustKernelHuber(double delta) { kernel_ = SIM3OPT_KERNEL_HUBER; kdelta_ = delta; }
 private:
  friend class SparseOptimizer;
  Vertex* v_[2] = {nullptr, nullptr};
  Sim3 meas_;
  std::array<double, 49> info_{};
  bool has_info_ = false;
  int kernel_ = SIM3OPT_KERNEL_NONE;
  double kdelta_ = 0.0;
};

// Tag types so that kitti_surf.cpp:553-557 compiles unchanged; the solver stack they name is
// what libsim3opt implements internally (LM + block solver + linear solver).
template <typename M> struct LinearSolverEigen {};
struct BlockSolverX {
  using PoseMatrixType = int;
  using LinearSolverType = LinearSolverEigen<PoseMatrixType>;
  explicit BlockSolverX(std::unique_ptr<LinearSolverType>) {}
};
struct OptimizationAlgorithmLevenberg {
  explicit OptimizationAlgorithmLevenberg(std::unique_ptr<BlockSolverX>) {}
  void setUserLambdaInit(double v) { user_lambda_init = v; }
  void setMaxTrialsAfterFailure(int n) { max_trials = n; }
  double user_lambda_init = 0.0;
  int max_trials = 10;
};
template <typename T, typename... A> std::unique_ptr<T> make_unique(A&&... a) {
  return std::unique_ptr<T>(new T(std::forward<A>(a)...));
}

class SparseOptimizer {  // g2o::SparseOptimizer (kitti_surf.cpp:552, 558, 620, 638, 674-675, 688)
 public:
  SparseOptimizer() : g_(sim3opt_create()) { if (!g_) throw std::bad_alloc(); }
  ~SparseOptimizer() { sim3opt_destroy(g_); }
  SparseOptimizer(const SparseOptimizer&) = delete;
  SparseOptimizer& operator=(const SparseOptimizer&) = delete;

  void setAlgorithm(OptimizationAlgorithmLevenberg* a) {  // takes ownership like g2o
    alg_.reset(a);
    sim3opt_options o;
    sim3opt_get_options(g_, &o);
    o.user_lambda_init = a->user_lambda_init;
    o.max_trials = a->max_trials;
    sim3opt_set_options(g_, &o);
  }
  void setVerbose(bool v) {
    sim3opt_options o;
    sim3opt_get_options(g_, &o);
    o.verbose = v ? 1 : 0;
    sim3opt_set_options(g_, &o);
  }
  bool addVertex(VertexSim3Expmap* v) {  // owns the vertex, like g2o
    if (sim3opt_add_vertex(g_, v->id_, v->estimate().v.data(), v->fixed_ ? 1 : 0) != SIM3OPT_OK) {
      delete v;
      return false;
    }
    v->owner_ = this;
    verts_[v->id_].reset(v);
    return true;
  }
  bool addEdge(EdgeSim3* e) {
    std::unique_ptr<EdgeSim3> own(e);
    if (!e->v_[0] || !e->v_[1]) return false;
    return sim3opt_add_edge(g_, e->v_[0]->id(), e->v_[1]->id(), e->meas_.v.data(),
                            e->has_info_ ? e->info_.data() : nullptr, e->kernel_,
                            e->kdelta_) == SIM3OPT_OK;
  }
  Vertex* vertex(int id) {
    auto it = verts_.find(id);
    return it == verts_.end() ? nullptr : it->second.get();
  }
  bool initializeOptimization() { return sim3opt_initialize(g_) == SIM3OPT_OK; }
  int optimize(int iterations) { return sim3opt_optimize(g_, iterations); }
  void computeActiveErrors() {}
  double activeChi2() { double c = 0; sim3opt_chi2(g_, &c); return c; }
  double activeRobustChi2() { return activeChi2(); }
  const char* lastError() const { return sim3opt_last_error(g_); }
  sim3opt_graph* handle() { return g_; }

 private:
  friend class VertexSim3Expmap;
  sim3opt_graph* g_;
  std::unique_ptr<OptimizationAlgorithmLevenberg> alg_;
  std::map<int, std::unique_ptr<VertexSim3Expmap>> verts_;
};

inline void VertexSim3Expmap::setEstimate(const Sim3& s) {
  est_ = s;
  if (owner_) sim3opt_set_vertex(owner_->g_, id_, est_.v.data());  // warm start, kitti_surf.cpp:1037-1038
}
inline Sim3 VertexSim3Expmap::estimate() const {
  if (!owner_) return est_;
  Sim3 s;
  sim3opt_get_vertex(owner_->g_, id_, s.v.data());
  return s;
}

}  // namespace sim3opt_shim

#if defined(SIM3OPT_G2O_NAMES)
// Opt-in: expose the shim under the names the reference spells.
namespace g2o {
using sim3opt_shim::BlockSolverX;
using sim3opt_shim::LinearSolverEigen;
using sim3opt_shim::make_unique;
using sim3opt_shim::OptimizationAlgorithmLevenberg;
using sim3opt_shim::Sim3;
using sim3opt_shim::SparseOptimizer;
}  // namespace g2o
namespace vio {
using sim3opt_shim::EdgeSim3;
using sim3opt_shim::VertexSim3Expmap;
}  // namespace vio
#endif

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<Eigen/Geometry>)
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace sim3opt_shim {
// Eigen-typed conveniences matching the reference's call sites (kitti_surf.cpp:200, 608, 693-698).
inline Sim3 makeSim3(const Eigen::Matrix3d& R, const Eigen::Vector3d& t, double s) {
  const double Rr[9] = {R(0, 0), R(0, 1), R(0, 2), R(1, 0), R(1, 1), R(1, 2), R(2, 0), R(2, 1), R(2, 2)};
  const double tt[3] = {t[0], t[1], t[2]};
  return Sim3(Rr, tt, s);
}
inline Eigen::Quaterniond rotation(const Sim3& S) { return Eigen::Quaterniond(S.v[3], S.v[0], S.v[1], S.v[2]); }
inline Eigen::Vector3d translation(const Sim3& S) { return Eigen::Vector3d(S.v[4], S.v[5], S.v[6]); }
inline void setInformation(EdgeSim3& e, const Eigen::Matrix<double, 7, 7>& M) {
  auto& a = e.information();
  for (int c = 0; c < 7; ++c) for (int r = 0; r < 7; ++r) a[7 * c + r] = M(r, c);
}
}  // namespace sim3opt_shim
#endif
#endif
