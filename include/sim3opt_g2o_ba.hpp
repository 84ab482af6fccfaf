// sim3opt_g2o_ba.hpp -- header-only C++ shim with the g2o operator surface of the reference's
// ba_demo (bal_example.cpp:44-243), forwarding to the sim3opt_ba_* entry points of libsim3opt
// (include/sim3opt.h, "bundle adjustment hand-off").
//
// With -DSIM3OPT_G2O_BA_NAMES the graph-building code of ba_demo compiles in its own call forms:
//
//     g2o::SparseOptimizer optimizer;  optimizer.setVerbose(verbose);                      // :71-72
//     std::unique_ptr<g2o::BlockSolver_6_3::LinearSolverType> linearSolver =
//         g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType> >(); // :78-79
//     optimizer.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(
//         g2o::make_unique<g2o::BlockSolver_6_3>(std::move(linearSolver))));               // :82-85
//     g2o::CameraParameters* cam_params = new g2o::CameraParameters(f, principal_point, 0.);
//     cam_params->setId(0);  optimizer.addParameter(cam_params);                           // :90-97
//     g2o::VertexSE3Expmap* cam = new g2o::VertexSE3Expmap();  cam->setId(id);
//     optimizer.addVertex(cam);                                                            // :112-118
//     g2o::VertexSBAPointXYZ* p = new g2o::VertexSBAPointXYZ();  p->setId(id);
//     p->setMarginalized(true);  optimizer.addVertex(p);                                   // :120-130
//     g2o::EdgeProjectXYZ2UV* e = new g2o::EdgeProjectXYZ2UV();
//     e->setVertex(0, point);  e->setVertex(1, cam);
//     e->setInformation(Eigen::Matrix2d::Identity() / (PIXEL_NOISE * PIXEL_NOISE));
//     e->setMeasurement(Eigen::Vector2d(obsX, obsY));
//     g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;  rk->setDelta(2.5);
//     e->setRobustKernel(rk);  e->setParameterId(0, 0);  optimizer.addEdge(e);
//     e->computeError();  e->error().norm();                                               // :145-162
//     cam->setEstimate(g2o::SE3Quat(qw2c, trans));  point->setEstimate(p);                 // :170-193
//     optimizer.initializeOptimization();  optimizer.optimize(maxIterations);              // :198, :213
//     g2o::SE3Quat est = cam->estimate();  est.rotation().conjugate();  est.translation(); // :229-236
//
// What the library implements behind it is exactly that stack (LM, Schur complement on the
// marginalised points, Huber kernel) on the GPU.  Its limits, checked at initializeOptimization():
// every edge carries the same isotropic information and the same Huber delta (or none), all edges
// use one CameraParameters, points cannot be fixed (cameras can).  ba_demo satisfies all of them.
//
// sim3opt_g2o.hpp exports g2o::SparseOptimizer for the Sim(3) pose graphs (kitti_surf.cpp); this
// header exports the one for bundle adjustment.  The reference uses them in different translation
// units, and so must a caller: defining both SIM3OPT_G2O_NAMES and SIM3OPT_G2O_BA_NAMES is an error.
#pragma once

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "sim3opt_g2o.hpp"

#if defined(SIM3OPT_G2O_NAMES) && defined(SIM3OPT_G2O_BA_NAMES)
#error "SIM3OPT_G2O_NAMES and SIM3OPT_G2O_BA_NAMES both export g2o::SparseOptimizer: one per translation unit"
#endif

namespace sim3opt_shim {
namespace ba {

#if defined(SIM3OPT_SHIM_EIGEN)
using Vector2 = Eigen::Matrix<double, 2, 1>;
inline Vector2 make_vector2(double a, double b) { Vector2 v; v[0] = a; v[1] = b; return v; }
inline Vector3 make_vector3(double a, double b, double c) { Vector3 v; v[0] = a; v[1] = b; v[2] = c; return v; }
#else
struct Vector2 : Vec<2> {
  double norm() const { return std::sqrt(d[0] * d[0] + d[1] * d[1]); }
};
inline Vector2 make_vector2(double a, double b) { Vector2 v; v[0] = a; v[1] = b; return v; }
inline Vector3 make_vector3(double a, double b, double c) { Vector3 v; v[0] = a; v[1] = b; v[2] = c; return v; }
#endif

// g2o::SE3Quat as far as ba_demo touches it (se3quat.h): unit quaternion + translation, T_w2c here
class SE3Quat {
 public:
  SE3Quat() {}
  // SE3Quat(const Quaterniond&, const Vector3d&) (:172); the rotation is normalised with w >= 0 as
  // g2o's normalizeRotation does
  template <class Q, class V> SE3Quat(const Q& q, const V& t) {
    q_[0] = q.x(); q_[1] = q.y(); q_[2] = q.z(); q_[3] = q.w();
    for (int i = 0; i < 3; ++i) t_[i] = detail::el(t, i);
    normalize();
  }
  static SE3Quat from_qt(const double* qt) {
    SE3Quat s;
    for (int i = 0; i < 4; ++i) s.q_[i] = qt[i];
    for (int i = 0; i < 3; ++i) s.t_[i] = qt[4 + i];
    return s;
  }
  Quaternion rotation() const { return make_quaternion(q_[0], q_[1], q_[2], q_[3]); }
  Vector3 translation() const { return make_vector3(t_[0], t_[1], t_[2]); }
  template <class V> Vector3 map(const V& p) const {
    const double in[3] = {detail::el(p, 0), detail::el(p, 1), detail::el(p, 2)};
    double o[3];
    detail::quat_rot(q_, in, o);
    return make_vector3(o[0] + t_[0], o[1] + t_[1], o[2] + t_[2]);
  }
  SE3Quat inverse() const {
    SE3Quat r;
    r.q_[0] = -q_[0]; r.q_[1] = -q_[1]; r.q_[2] = -q_[2]; r.q_[3] = q_[3];
    double o[3];
    detail::quat_rot(r.q_, t_, o);
    for (int i = 0; i < 3; ++i) r.t_[i] = -o[i];
    return r;
  }
  void to_qt(double* qt) const {
    for (int i = 0; i < 4; ++i) qt[i] = q_[i];
    for (int i = 0; i < 3; ++i) qt[4 + i] = t_[i];
  }

 private:
  void normalize() {
    double n = std::sqrt(q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]);
    if (q_[3] < 0) n = -n;
    if (n != 0) for (double& c : q_) c /= n;
  }
  double q_[4] = {0, 0, 0, 1};
  double t_[3] = {0, 0, 0};
};

class SparseOptimizer;

class Vertex {  // g2o::OptimizableGraph::Vertex, as far as ba_demo touches it
 public:
  virtual ~Vertex() {}
  void setId(int id) { id_ = id; }
  int id() const { return id_; }
  void setFixed(bool f) { fixed_ = f; }
  bool fixed() const { return fixed_; }
  void setMarginalized(bool m) { marginalized_ = m; }
  bool marginalized() const { return marginalized_; }
  virtual int dimension() const = 0;

 protected:
  int id_ = -1;
  bool fixed_ = false, marginalized_ = false;
};

class VertexSE3Expmap : public Vertex {  // g2o::VertexSE3Expmap (bal_example.cpp:112-118, :170-175)
 public:
  void setEstimate(const SE3Quat& e) { est_ = e; }
  const SE3Quat& estimate() const { return est_; }
  int dimension() const override { return 6; }

 private:
  SE3Quat est_;
};

class VertexSBAPointXYZ : public Vertex {  // g2o::VertexSBAPointXYZ (bal_example.cpp:120-130, :188-193)
 public:
  template <class V> void setEstimate(const V& p) { for (int i = 0; i < 3; ++i) p_[i] = detail::el(p, i); }
  Vector3 estimate() const { return make_vector3(p_[0], p_[1], p_[2]); }
  int dimension() const override { return 3; }
  const double* data() const { return p_; }
  double* data() { return p_; }

 private:
  double p_[3] = {0, 0, 0};
};

class CameraParameters {  // g2o::CameraParameters(focal_length, principal_point, baseline) (:93-94)
 public:
  template <class V> CameraParameters(double f, const V& pp, double baseline)
      : focal_length(f), baseline(baseline) {
    principle_point[0] = detail::el(pp, 0);
    principle_point[1] = detail::el(pp, 1);
  }
  void setId(int id) { id_ = id; }
  int id() const { return id_; }
  template <class V> Vector2 cam_map(const V& X) const {
    const double x = detail::el(X, 0), y = detail::el(X, 1), z = detail::el(X, 2);
    return make_vector2(focal_length * x / z + principle_point[0], focal_length * y / z + principle_point[1]);
  }
  double focal_length;
  double principle_point[2];  // (g2o's own spelling)
  double baseline;

 private:
  int id_ = -1;
};

class RobustKernelHuber {  // g2o::RobustKernelHuber (:150-152)
 public:
  void setDelta(double d) { delta_ = d; }
  double delta() const { return delta_; }

 private:
  double delta_ = 1.0;
};

class EdgeProjectXYZ2UV {  // g2o::EdgeProjectXYZ2UV (:145-162): vertex 0 the point, vertex 1 the camera
 public:
  void setVertex(int i, Vertex* v) {
    if (i == 0) point_ = dynamic_cast<VertexSBAPointXYZ*>(v);
    if (i == 1) cam_ = dynamic_cast<VertexSE3Expmap*>(v);
  }
  template <class M> void setInformation(const M& m) {
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) info_[2 * r + c] = m(r, c);
  }
  template <class V> void setMeasurement(const V& z) { z_[0] = detail::el(z, 0); z_[1] = detail::el(z, 1); }
  void setRobustKernel(RobustKernelHuber* rk) { kernel_.reset(rk); }  // owned, like g2o
  bool setParameterId(int arg, int param_id) {
    if (arg != 0) return false;
    param_id_ = param_id;
    return true;
  }
  // obs - cam_map(T_w2c.map(point))      (EdgeProjectXYZ2UV::computeError)
  void computeError() {
    if (!cam_ || !point_ || !params_) return;
    const Vector2 uv = params_->cam_map(cam_->estimate().map(point_->estimate()));
    err_[0] = z_[0] - uv[0];
    err_[1] = z_[1] - uv[1];
  }
  Vector2 error() const { return make_vector2(err_[0], err_[1]); }

 private:
  friend class SparseOptimizer;
  VertexSBAPointXYZ* point_ = nullptr;
  VertexSE3Expmap* cam_ = nullptr;
  const CameraParameters* params_ = nullptr;
  std::unique_ptr<RobustKernelHuber> kernel_;
  int param_id_ = -1;
  double info_[4] = {1, 0, 0, 1};
  double z_[2] = {0, 0};
  double err_[2] = {0, 0};
};

// Tag types so that bal_example.cpp:73-85 compiles unchanged; the stack they name is what
// sim3opt_amd/csrc/ba.hip implements (LM + Schur complement + reduced-system solve).
struct BlockSolver_6_3 {
  using PoseMatrixType = int;
  // unique_ptr<LinearSolverType> accepts LinearSolverEigen<...> and LinearSolverDense<...> alike
  struct LinearSolverType {
    LinearSolverType() {}
    virtual ~LinearSolverType() {}
  };
  template <class S> explicit BlockSolver_6_3(std::unique_ptr<S>) {}
};
template <typename M> struct LinearSolver63Eigen : BlockSolver_6_3::LinearSolverType {};
template <typename M> struct LinearSolver63Dense : BlockSolver_6_3::LinearSolverType {};
struct OptimizationAlgorithmLevenberg {
  explicit OptimizationAlgorithmLevenberg(std::unique_ptr<BlockSolver_6_3>) {}
  void setUserLambdaInit(double v) { user_lambda_init = v; }
  void setMaxTrialsAfterFailure(int n) { max_trials = n; }
  double user_lambda_init = 0.0;
  int max_trials = 10;
};

class SparseOptimizer {  // g2o::SparseOptimizer (bal_example.cpp:71-72, :85, :95, :115, :125, :155, :198, :213)
 public:
  SparseOptimizer() : b_(sim3opt_ba_create()) {
    if (!b_) throw std::bad_alloc();
    sim3opt_ba_options_default(&opt_);
  }
  ~SparseOptimizer() { sim3opt_ba_destroy(b_); }
  SparseOptimizer(const SparseOptimizer&) = delete;
  SparseOptimizer& operator=(const SparseOptimizer&) = delete;

  void setVerbose(bool v) { opt_.verbose = v ? 1 : 0; }
  void setAlgorithm(OptimizationAlgorithmLevenberg* a) {  // takes ownership like g2o
    alg_.reset(a);
    opt_.user_lambda_init = a->user_lambda_init;
    opt_.max_trials = a->max_trials;
  }
  bool addParameter(CameraParameters* p) {
    std::unique_ptr<CameraParameters> own(p);
    if (params_.count(p->id())) return false;
    params_[p->id()] = std::move(own);
    return true;
  }
  bool addVertex(Vertex* v) {  // owns the vertex, like g2o
    std::unique_ptr<Vertex> own(v);
    if (v->id() < 0 || verts_.count(v->id())) return false;
    verts_[v->id()] = std::move(own);
    return true;
  }
  bool addEdge(EdgeProjectXYZ2UV* e) {
    std::unique_ptr<EdgeProjectXYZ2UV> own(e);
    if (!e->cam_ || !e->point_) return false;
    auto it = params_.find(e->param_id_);  // OptimizableGraph::addEdge resolves the parameters
    if (it == params_.end()) return false;
    e->params_ = it->second.get();
    edges_.push_back(std::move(own));
    return true;
  }
  Vertex* vertex(int id) {
    auto it = verts_.find(id);
    return it == verts_.end() ? nullptr : it->second.get();
  }
  const std::map<int, std::unique_ptr<Vertex>>& vertices() const { return verts_; }

  // Hands the problem to the library.  Cameras / points are numbered in ascending vertex id.
  bool initializeOptimization() {
    cams_.clear();
    pts_.clear();
    std::map<const Vertex*, int> index;
    for (auto& kv : verts_) {
      Vertex* v = kv.second.get();
      if (auto* c = dynamic_cast<VertexSE3Expmap*>(v)) { index[v] = (int)cams_.size(); cams_.push_back(c); }
      else if (auto* p = dynamic_cast<VertexSBAPointXYZ*>(v)) {
        if (p->fixed()) return fail("fixed points are not supported");
        index[v] = (int)pts_.size();
        pts_.push_back(p);
      }
    }
    if (cams_.empty() || pts_.empty() || edges_.empty()) return fail("empty problem");
    const EdgeProjectXYZ2UV& e0 = *edges_[0];
    const double delta = e0.kernel_ ? e0.kernel_->delta() : 0.0;
    if (!(e0.info_[0] > 0) || e0.info_[0] != e0.info_[3] || e0.info_[1] != 0 || e0.info_[2] != 0)
      return fail("information must be a positive multiple of the identity");
    std::vector<int32_t> oc(edges_.size()), op(edges_.size());
    std::vector<double> uv(2 * edges_.size());
    for (size_t k = 0; k < edges_.size(); ++k) {
      const EdgeProjectXYZ2UV& e = *edges_[k];
      for (int i = 0; i < 4; ++i)
        if (e.info_[i] != e0.info_[i]) return fail("edges with different information");
      if ((e.kernel_ ? e.kernel_->delta() : 0.0) != delta) return fail("edges with different robust kernels");
      if (e.params_ != e0.params_) return fail("edges with different camera parameters");
      auto ic = index.find(e.cam_), ip = index.find(e.point_);
      if (ic == index.end() || ip == index.end()) return fail("edge on a vertex that was not added");
      oc[k] = ic->second;
      op[k] = ip->second;
      uv[2 * k] = e.z_[0];
      uv[2 * k + 1] = e.z_[1];
    }
    std::vector<double> cq(7 * cams_.size()), pp(3 * pts_.size());
    std::vector<uint8_t> fixed(cams_.size());
    for (size_t c = 0; c < cams_.size(); ++c) {
      cams_[c]->estimate().to_qt(&cq[7 * c]);
      fixed[c] = cams_[c]->fixed() ? 1 : 0;
    }
    for (size_t p = 0; p < pts_.size(); ++p) std::copy(pts_[p]->data(), pts_[p]->data() + 3, &pp[3 * p]);
    opt_.huber_delta = delta;
    opt_.pixel_noise = 1.0 / std::sqrt(e0.info_[0]);
    const CameraParameters& K = *e0.params_;
    if (sim3opt_ba_set_options(b_, &opt_) != SIM3OPT_OK) return false;
    if (sim3opt_ba_set_problem(b_, (int32_t)cams_.size(), cq.data(), (int32_t)pts_.size(), pp.data(),
                               (int32_t)edges_.size(), oc.data(), op.data(), uv.data(), K.focal_length,
                               K.principle_point[0], K.principle_point[1]) != SIM3OPT_OK)
      return false;
    if (sim3opt_ba_set_fixed_cameras(b_, fixed.data()) != SIM3OPT_OK) return false;
    ready_ = true;
    return true;
  }
  // LM iterations performed; the estimates of every vertex object are updated afterwards
  int optimize(int iterations) {
    if (!ready_) return -1;
    const int n = sim3opt_ba_optimize(b_, iterations);
    std::vector<double> cq(7 * cams_.size()), pp(3 * pts_.size());
    if (sim3opt_ba_get_cameras(b_, cq.data()) != SIM3OPT_OK || sim3opt_ba_get_points(b_, pp.data()) != SIM3OPT_OK)
      return 0;
    for (size_t c = 0; c < cams_.size(); ++c) cams_[c]->setEstimate(SE3Quat::from_qt(&cq[7 * c]));
    for (size_t p = 0; p < pts_.size(); ++p) std::copy(&pp[3 * p], &pp[3 * p] + 3, pts_[p]->data());
    return n;
  }
  void computeActiveErrors() {}
  double activeRobustChi2() { double c = 0; sim3opt_ba_chi2(b_, &c); return c; }
  double chi2() { return activeRobustChi2(); }
  const char* lastError() const { return err_.empty() ? sim3opt_ba_last_error(b_) : err_.c_str(); }
  sim3opt_ba* handle() { return b_; }

 private:
  bool fail(const char* why) { err_ = why; return false; }
  sim3opt_ba* b_;
  sim3opt_ba_options opt_;
  bool ready_ = false;
  std::string err_;
  std::unique_ptr<OptimizationAlgorithmLevenberg> alg_;
  std::map<int, std::unique_ptr<CameraParameters>> params_;
  std::map<int, std::unique_ptr<Vertex>> verts_;
  std::vector<std::unique_ptr<EdgeProjectXYZ2UV>> edges_;
  std::vector<VertexSE3Expmap*> cams_;
  std::vector<VertexSBAPointXYZ*> pts_;
};

}  // namespace ba
}  // namespace sim3opt_shim

#if defined(SIM3OPT_G2O_BA_NAMES)
// Opt-in: expose the shim under the names ba_demo spells.
namespace g2o {
using sim3opt_shim::make_unique;
using sim3opt_shim::ba::BlockSolver_6_3;
using sim3opt_shim::ba::CameraParameters;
using sim3opt_shim::ba::EdgeProjectXYZ2UV;
using sim3opt_shim::ba::OptimizationAlgorithmLevenberg;
using sim3opt_shim::ba::RobustKernelHuber;
using sim3opt_shim::ba::SE3Quat;
using sim3opt_shim::ba::SparseOptimizer;
using sim3opt_shim::ba::VertexSBAPointXYZ;
using sim3opt_shim::ba::VertexSE3Expmap;
template <typename M> using LinearSolverEigen = sim3opt_shim::ba::LinearSolver63Eigen<M>;
template <typename M> using LinearSolverDense = sim3opt_shim::ba::LinearSolver63Dense<M>;
}  // namespace g2o
#endif
