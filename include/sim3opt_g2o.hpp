// sim3opt_g2o.hpp -- header-only C++ shim with the g2o / vio_g2o operator surface the reference
// calls, forwarding to the C-ABI of libsim3opt (include/sim3opt.h).
//
// With -DSIM3OPT_G2O_NAMES the graph-building code of testDirectSim3Optimization
// (kitti_surf.cpp:552-558, :597-701) and testStepwiseSim3Optimization (:726-735, :774-886,
// :1020-1075) compiles in its own call forms:
//
//     g2o::SparseOptimizer optimizer;                                              // :552
//     std::unique_ptr<g2o::BlockSolverX::LinearSolverType> linearSolver =
//         g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType> >();
//     optimizer.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(              // :556-558
//         g2o::make_unique<g2o::BlockSolverX>(std::move(linearSolver))));
//     vio::VertexSim3Expmap* v = new vio::VertexSim3Expmap();                      // :602
//     g2o::Sim3 Siw(Rcw, tcw, 1.0);  v->setEstimate(Siw);  v->setFixed(true);  v->setId(i);
//     v->setMarginalized(false);  optimizer.addVertex(v);                          // :608-620
//     vio::EdgeSim3* e = new vio::EdgeSim3();                                      // :633
//     e->setVertex(1, optimizer.vertex(j));  e->setVertex(0, optimizer.vertex(i));
//     e->setMeasurement(Sji);  e->information() = matLambdasim;  optimizer.addEdge(e);
//     optimizer.initializeOptimization();  optimizer.optimize(100);                // :674-675
//     g2o::Sim3 S = static_cast<vio::VertexSim3Expmap*>(optimizer.vertex(i))->estimate();
//     S.inverse().translation().transpose();  S.rotation().coeffs();  S.scale();   // :691-698
//     vio::G2oVertexScaleTrans* vST = ...; vST->setEstimate(toScaleTrans(Siw));
//     vST->Rw2i = Sophus::SO3d(Rcw);                                               // :779-793
//     g2o::Sim3 C(vST->Rw2i.unit_quaternion(), stw2i.tail<3>(), stw2i[0]);        // :1035
//
// Matrix / vector / quaternion arguments are templates: anything with m(r, c), v[i] or v(i),
// q.x() .. q.w() works -- Eigen at the user's site, or the fixed-size mock the tests compile against
// (tests/mock_eigen; this image has no Eigen).  Return types (rotation(), translation(),
// G2oVertexScaleTrans::estimate()) are Eigen's when <Eigen/Core> and <Eigen/Geometry> are on the
// include path, else the small types below.  tests/cxx/shim_conformance.cpp is the conformance test.
//
// How the stepwise optimizers map onto the library (kitti_surf.cpp:774-886): an optimizer of
// G2oVertexScaleTrans / G2oEdgeScaleTrans is a Sim(3) graph whose rotations are frozen
// (options.dof_mask = 0x78) with vertex state (Rw2i, t, s) and edge measurement
// (Rw2i(v1) Rw2i(v0)^T, t, s); one of G2oVertexScale / G2oEdgeScale freezes everything but the scale
// (dof_mask = 0x40).  vio_g2o's own edge classes are not available (build.sh:92 fetches them), so the
// residual is the Sim(3) one restricted to those components (stepwise stages).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <utility>
#include <vector>

#include "sim3opt.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<Eigen/Geometry>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define SIM3OPT_SHIM_EIGEN 1
#endif
#endif

namespace sim3opt_shim {

namespace detail {
// element of a foreign vector type: v[i] if it has one, else v(i)
template <class V> auto at(const V& v, int i, int) -> decltype(static_cast<double>(v[i])) { return v[i]; }
template <class V> auto at(const V& v, int i, long) -> decltype(static_cast<double>(v(i))) { return v(i); }
template <class V> double el(const V& v, int i) { return at(v, i, 0); }
// quaternion-like: has w()
template <class T> auto is_quat(const T& q, int) -> decltype(static_cast<double>(q.w()), std::true_type());
template <class T> std::false_type is_quat(const T&, long);

inline void quat_from_rowmajor(const double R[9], double q[4]) {  // Eigen's Quaternion(Matrix3) rule
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double k = std::sqrt(tr + 1.0);
    q[3] = 0.5 * k; k = 0.5 / k;
    q[0] = (R[7] - R[5]) * k; q[1] = (R[2] - R[6]) * k; q[2] = (R[3] - R[1]) * k;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k3 = (j + 1) % 3;
    double k = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k3] + 1.0);
    q[i] = 0.5 * k; k = 0.5 / k;
    q[3] = (R[3 * k3 + j] - R[3 * j + k3]) * k;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * k;
    q[k3] = (R[3 * k3 + i] + R[3 * i + k3]) * k;
  }
}
inline void quat_mul(const double* a, const double* c, double* o) {
  o[0] = a[3] * c[0] + a[0] * c[3] + a[1] * c[2] - a[2] * c[1];
  o[1] = a[3] * c[1] + a[1] * c[3] + a[2] * c[0] - a[0] * c[2];
  o[2] = a[3] * c[2] + a[2] * c[3] + a[0] * c[1] - a[1] * c[0];
  o[3] = a[3] * c[3] - a[0] * c[0] - a[1] * c[1] - a[2] * c[2];
}
inline void quat_rot(const double* q, const double* p, double* o) {
  const double ux = 2 * (q[1] * p[2] - q[2] * p[1]), uy = 2 * (q[2] * p[0] - q[0] * p[2]),
               uz = 2 * (q[0] * p[1] - q[1] * p[0]);
  o[0] = p[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = p[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = p[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
}  // namespace detail

#if defined(SIM3OPT_SHIM_EIGEN)
using Vector3 = Eigen::Matrix<double, 3, 1>;
using Vector4 = Eigen::Matrix<double, 4, 1>;
using Quaternion = Eigen::Quaternion<double>;
inline Quaternion make_quaternion(double x, double y, double z, double w) { return Quaternion(w, x, y, z); }
#else
// What the reference does with these return values (kitti_surf.cpp:693-698, :1034-1035): index, divide
// by a scalar, transpose() into a stream, tail<3>(), coeffs().
template <int N> struct Vec {
  std::array<double, N> d{};
  double& operator[](int i) { return d[i]; }
  double operator[](int i) const { return d[i]; }
  double& operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
  Vec operator/(double s) const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = d[i] / s; return r; }
  Vec operator*(double s) const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = d[i] * s; return r; }
  const Vec& transpose() const { return *this; }
  template <int M> Vec<M> tail() const { Vec<M> r; for (int i = 0; i < M; ++i) r.d[i] = d[N - M + i]; return r; }
  template <int M> Vec<M> head() const { Vec<M> r; for (int i = 0; i < M; ++i) r.d[i] = d[i]; return r; }
};
template <int N> std::ostream& operator<<(std::ostream& os, const Vec<N>& v) {
  for (int i = 0; i < N; ++i) os << (i ? " " : "") << v.d[i];
  return os;
}
using Vector3 = Vec<3>;
using Vector4 = Vec<4>;
struct Quaternion {
  double q[4] = {0, 0, 0, 1};  // x y z w
  double x() const { return q[0]; }
  double y() const { return q[1]; }
  double z() const { return q[2]; }
  double w() const { return q[3]; }
  Vector4 coeffs() const { Vector4 v; for (int i = 0; i < 4; ++i) v[i] = q[i]; return v; }
};
inline Quaternion make_quaternion(double x, double y, double z, double w) {
  Quaternion r; r.q[0] = x; r.q[1] = y; r.q[2] = z; r.q[3] = w; return r;
}
#endif

// g2o::Sim3 -- unit quaternion (x,y,z,w), translation, scale                (sim3_rv.h:71-226)
struct Sim3 {
  std::array<double, 8> v{{0, 0, 0, 1, 0, 0, 0, 1}};
  Sim3() = default;
  explicit Sim3(const double state[8]) { for (int i = 0; i < 8; ++i) v[i] = state[i]; }
  // row-major 3x3 array, translation array, scale
  Sim3(const double R[9], const double t[3], double s) {
    detail::quat_from_rowmajor(R, v.data());
    v[4] = t[0]; v[5] = t[1]; v[6] = t[2]; v[7] = s;
  }
  // g2o: Sim3(const Matrix3& R, const Vector3& t, double s) and Sim3(const Quaternion& r, const
  // Vector3& t, double s) (kitti_surf.cpp:200, :608, :1035) for any matrix / quaternion / vector type
  template <class Rot, class Vec3,
            class = decltype(detail::is_quat(std::declval<const Rot&>(), 0)),
            class = decltype(detail::el(std::declval<const Vec3&>(), 0))>
  Sim3(const Rot& r, const Vec3& t, double s) {
    set_rotation(r, decltype(detail::is_quat(r, 0))());
    for (int i = 0; i < 3; ++i) v[4 + i] = detail::el(t, i);
    v[7] = s;
  }
  double scale() const { return v[7]; }
  Quaternion rotation() const { return make_quaternion(v[0], v[1], v[2], v[3]); }
  Vector3 translation() const { Vector3 t; t[0] = v[4]; t[1] = v[5]; t[2] = v[6]; return t; }
  const double* quaternion_xyzw() const { return v.data(); }
  const double* translation_ptr() const { return v.data() + 4; }
  Sim3 inverse() const {                                                    // sim3_rv.h:199-203
    Sim3 r;
    r.v[0] = -v[0]; r.v[1] = -v[1]; r.v[2] = -v[2]; r.v[3] = v[3];
    const double k = -1.0 / v[7];
    const double tmp[3] = {k * v[4], k * v[5], k * v[6]};
    detail::quat_rot(r.v.data(), tmp, r.v.data() + 4);
    r.v[7] = 1.0 / v[7];
    return r;
  }
  Sim3 operator*(const Sim3& b) const {                                     // sim3_rv.h:214-220
    Sim3 r;
    detail::quat_mul(v.data(), b.v.data(), r.v.data());
    double rt[3];
    detail::quat_rot(v.data(), b.v.data() + 4, rt);
    for (int i = 0; i < 3; ++i) r.v[4 + i] = v[7] * rt[i] + v[4 + i];
    r.v[7] = v[7] * b.v[7];
    return r;
  }
  template <class Vec3> Vector3 map(const Vec3& p) const {                  // s R p + t
    const double pp[3] = {detail::el(p, 0), detail::el(p, 1), detail::el(p, 2)};
    double rp[3];
    detail::quat_rot(v.data(), pp, rp);
    Vector3 o;
    for (int i = 0; i < 3; ++i) o[i] = v[7] * rp[i] + v[4 + i];
    return o;
  }

 private:
  template <class Q> void set_rotation(const Q& q, std::true_type) {
    const double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z() + q.w() * q.w());
    v[0] = q.x() / n; v[1] = q.y() / n; v[2] = q.z() / n; v[3] = q.w() / n;
  }
  template <class M> void set_rotation(const M& R, std::false_type) {
    const double Rr[9] = {R(0, 0), R(0, 1), R(0, 2), R(1, 0), R(1, 1), R(1, 2), R(2, 0), R(2, 1), R(2, 2)};
    detail::quat_from_rowmajor(Rr, v.data());
  }
};

// edge->information() = M  for an N x N matrix-like M (kitti_surf.cpp:637, :667, :832, :839, :846)
template <int N> class InformationRef {
 public:
  InformationRef(double* a, bool* set) : a_(a), set_(set) { *set_ = true; }
  template <class M, class = decltype(static_cast<double>(std::declval<const M&>()(0, 0)))>
  InformationRef& operator=(const M& m) {
    for (int c = 0; c < N; ++c)
      for (int r = 0; r < N; ++r) a_[N * c + r] = m(r, c);
    return *this;
  }
  double& operator()(int r, int c) { return a_[N * c + r]; }
  double& operator[](int k) { return a_[k]; }  // column-major

 private:
  double* a_;
  bool* set_;
};

class SparseOptimizer;
class G2oEdgeScaleTrans;

// Sophus::SO3d stand-in for G2oVertexScaleTrans::Rw2i (kitti_surf.cpp:793, :1035, :1063): assignable
// from anything that has unit_quaternion() (Sophus::SO3d), a quaternion, or a 3x3 matrix.
class FrozenRotation {
 public:
  FrozenRotation() = default;
  template <class R> FrozenRotation& operator=(const R& r) {
    assign(r, 0);
    return *this;
  }
  Quaternion unit_quaternion() const { return make_quaternion(q_[0], q_[1], q_[2], q_[3]); }
  const double* xyzw() const { return q_; }

 private:
  template <class R> auto assign(const R& r, int) -> decltype(r.unit_quaternion(), void()) {
    const auto q = r.unit_quaternion();
    q_[0] = q.x(); q_[1] = q.y(); q_[2] = q.z(); q_[3] = q.w();
  }
  template <class R> void assign(const R& r, long) {
    const Sim3 s(r, std::array<double, 3>{{0, 0, 0}}, 1.0);
    for (int i = 0; i < 4; ++i) q_[i] = s.v[i];
  }
  double q_[4] = {0, 0, 0, 1};
};

enum class GraphKind { Unset, Sim3, ScaleTrans, Scale };

class Vertex {  // g2o::OptimizableGraph::Vertex, as far as the reference touches it
 public:
  virtual ~Vertex() = default;
  void setId(int id) { id_ = id; }
  int id() const { return id_; }
  void setFixed(bool f) { fixed_ = f; }
  bool fixed() const { return fixed_; }
  void setMarginalized(bool) {}  // kitti_surf.cpp:619, :805-813 always pass false
 protected:
  friend class SparseOptimizer;
  friend class G2oEdgeScaleTrans;
  virtual GraphKind kind() const = 0;
  virtual void state(double s[8]) const = 0;  // the Sim(3) state the library holds for this vertex
  void push();                                 // estimate changed after addVertex: warm start
  bool pull(double s[8]) const;
  int id_ = -1;
  bool fixed_ = false;
  SparseOptimizer* owner_ = nullptr;
};

class VertexSim3Expmap : public Vertex {  // vio::VertexSim3Expmap (kitti_surf.cpp:602-620)
 public:
  void setEstimate(const Sim3& s) { est_ = s; push(); }
  Sim3 estimate() const { Sim3 s = est_; pull(s.v.data()); return s; }
 private:
  GraphKind kind() const override { return GraphKind::Sim3; }
  void state(double s[8]) const override { for (int i = 0; i < 8; ++i) s[i] = est_.v[i]; }
  Sim3 est_;
};

class G2oVertexScale : public Vertex {  // vio::G2oVertexScale (kitti_surf.cpp:779, :787, :926-927)
 public:
  void setEstimate(double s) { s_ = s; push(); }
  double estimate() const { double st[8]; return pull(st) ? st[7] : s_; }
 private:
  GraphKind kind() const override { return GraphKind::Scale; }
  void state(double s[8]) const override {
    const double id[8] = {0, 0, 0, 1, 0, 0, 0, s_};
    for (int i = 0; i < 8; ++i) s[i] = id[i];
  }
  double s_ = 1.0;
};

class G2oVertexScaleTrans : public Vertex {  // vio::G2oVertexScaleTrans (kitti_surf.cpp:780, :788-793)
 public:
  template <class V4> void setEstimate(const V4& st) {  // [s, t]  (toScaleTrans, kitti_surf.cpp:533-538)
    for (int i = 0; i < 4; ++i) st_[i] = detail::el(st, i);
    push();
  }
  Vector4 estimate() const {
    double s[8];
    Vector4 r;
    if (pull(s)) { r[0] = s[7]; r[1] = s[4]; r[2] = s[5]; r[3] = s[6]; }
    else for (int i = 0; i < 4; ++i) r[i] = st_[i];
    return r;
  }
  FrozenRotation Rw2i;
 private:
  GraphKind kind() const override { return GraphKind::ScaleTrans; }
  void state(double s[8]) const override {
    for (int i = 0; i < 4; ++i) s[i] = Rw2i.xyzw()[i];
    s[4] = st_[1]; s[5] = st_[2]; s[6] = st_[3]; s[7] = st_[0];
  }
  double st_[4] = {1, 0, 0, 0};
};

class Edge {  // binary edge, as far as the reference touches it
 public:
  virtual ~Edge() = default;
  void setVertex(int slot, Vertex* v) { v_[slot] = v; }
  void setRobustKernelHuber(double delta) { kernel_ = SIM3OPT_KERNEL_HUBER; kdelta_ = delta; }
 protected:
  friend class SparseOptimizer;
  virtual GraphKind kind() const = 0;
  virtual void measurement(double m[8]) const = 0;
  virtual bool information77(double out[49]) const = 0;  // false: identity
  Vertex* v_[2] = {nullptr, nullptr};
  int kernel_ = SIM3OPT_KERNEL_NONE;
  double kdelta_ = 0.0;
};

class EdgeSim3 : public Edge {  // vio::EdgeSim3 (kitti_surf.cpp:633-638, :663-668)
 public:
  void setMeasurement(const Sim3& m) { meas_ = m; }
  InformationRef<7> information() { return InformationRef<7>(info_.data(), &has_info_); }
 private:
  GraphKind kind() const override { return GraphKind::Sim3; }
  void measurement(double m[8]) const override { for (int i = 0; i < 8; ++i) m[i] = meas_.v[i]; }
  bool information77(double out[49]) const override {
    if (!has_info_) return false;
    for (int i = 0; i < 49; ++i) out[i] = info_[i];
    return true;
  }
  Sim3 meas_;
  std::array<double, 49> info_{};
  bool has_info_ = false;
};

class G2oEdgeScale : public Edge {  // vio::G2oEdgeScale (kitti_surf.cpp:827-832): s_v1 = s_meas s_v0
 public:
  void setMeasurement(double s) { s_ = s; }
  InformationRef<1> information() { return InformationRef<1>(&w_, &has_info_); }
 private:
  GraphKind kind() const override { return GraphKind::Scale; }
  void measurement(double m[8]) const override {
    const double id[8] = {0, 0, 0, 1, 0, 0, 0, s_};
    for (int i = 0; i < 8; ++i) m[i] = id[i];
  }
  bool information77(double out[49]) const override {
    if (!has_info_ || w_ == 1.0) return false;
    for (int i = 0; i < 49; ++i) out[i] = (i % 8 == 0) ? 1.0 : 0.0;
    out[48] = w_;  // sigma is tangent component 6
    return true;
  }
  double s_ = 1.0, w_ = 1.0;
  bool has_info_ = false;
};

class G2oEdgeScaleTrans : public Edge {  // vio::G2oEdgeScaleTrans (kitti_surf.cpp:834-839)
 public:
  template <class V4> void setMeasurement(const V4& st) { for (int i = 0; i < 4; ++i) st_[i] = detail::el(st, i); }
  InformationRef<4> information() { return InformationRef<4>(info_.data(), &has_info_); }
 private:
  GraphKind kind() const override { return GraphKind::ScaleTrans; }
  void measurement(double m[8]) const override {
    // rotation of the measurement = Rw2i(v1) Rw2i(v0)^T: with the rotations frozen the rotational
    // part of log(C S_v0 S_v1^-1) vanishes identically
    double q0[8], q1[8];
    static_cast<const Vertex*>(v_[0])->state(q0);
    static_cast<const Vertex*>(v_[1])->state(q1);
    const double c0[4] = {-q0[0], -q0[1], -q0[2], q0[3]};
    detail::quat_mul(q1, c0, m);
    m[4] = st_[1]; m[5] = st_[2]; m[6] = st_[3]; m[7] = st_[0];
  }
  bool information77(double out[49]) const override {
    if (!has_info_) return false;
    static const int map[4] = {6, 3, 4, 5};  // [s, t] -> tangent [omega(0:3), upsilon(3:6), sigma(6)]
    for (int i = 0; i < 49; ++i) out[i] = (i % 8 == 0) ? 1.0 : 0.0;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) out[7 * map[c] + map[r]] = info_[4 * c + r];
    return true;
  }
  double st_[4] = {1, 0, 0, 0};
  std::array<double, 16> info_{};
  bool has_info_ = false;
};

// Tag types so that kitti_surf.cpp:553-557 / :728-732 compile unchanged; the solver stack they name
// is what libsim3opt implements internally (LM + block solver + exact / PCG linear solver).
template <typename M> struct LinearSolverEigen {};
struct BlockSolverX {
  using PoseMatrixType = int;
  using LinearSolverType = LinearSolverEigen<PoseMatrixType>;
  explicit BlockSolverX(std::unique_ptr<LinearSolverType>) {}
};
struct OptimizationAlgorithmLevenberg {
  explicit OptimizationAlgorithmLevenberg(std::unique_ptr<BlockSolverX>) {}
  void setUserLambdaInit(double v) { user_lambda_init = v; }   // kittiDetector.h:779-782
  void setMaxTrialsAfterFailure(int n) { max_trials = n; }     // kittiDetector.h:730
  double user_lambda_init = 0.0;
  int max_trials = 10;
};
template <typename T, typename... A> std::unique_ptr<T> make_unique(A&&... a) {
  return std::unique_ptr<T>(new T(std::forward<A>(a)...));
}

class SparseOptimizer {  // g2o::SparseOptimizer (kitti_surf.cpp:552, 558, 620, 638, 674-675, 688, 726)
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
  bool addVertex(Vertex* v) {  // owns the vertex, like g2o
    std::unique_ptr<Vertex> own(v);
    if (!claim(v->kind())) return false;
    double s[8];
    v->state(s);
    if (sim3opt_add_vertex(g_, v->id_, s, v->fixed_ ? 1 : 0) != SIM3OPT_OK) return false;
    v->owner_ = this;
    verts_[v->id_] = std::move(own);
    return true;
  }
  bool addEdge(Edge* e) {
    std::unique_ptr<Edge> own(e);
    if (!e->v_[0] || !e->v_[1] || !claim(e->kind())) return false;
    double m[8], info[49];
    e->measurement(m);
    const bool has = e->information77(info);
    return sim3opt_add_edge(g_, e->v_[0]->id(), e->v_[1]->id(), m, has ? info : nullptr, e->kernel_,
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
  double chi2() { return activeChi2(); }
  const char* lastError() const { return sim3opt_last_error(g_); }
  sim3opt_graph* handle() { return g_; }

 private:
  friend class Vertex;
  // the first vertex / edge decides what this optimizer optimises (kitti_surf.cpp:809-814)
  bool claim(GraphKind k) {
    if (kind_ == GraphKind::Unset) {
      kind_ = k;
      sim3opt_options o;
      sim3opt_get_options(g_, &o);
      o.dof_mask = k == GraphKind::Sim3 ? 127 : (k == GraphKind::ScaleTrans ? 0x78 : 0x40);
      sim3opt_set_options(g_, &o);
    }
    return kind_ == k;
  }
  sim3opt_graph* g_;
  GraphKind kind_ = GraphKind::Unset;
  std::unique_ptr<OptimizationAlgorithmLevenberg> alg_;
  std::map<int, std::unique_ptr<Vertex>> verts_;
};

inline void Vertex::push() {
  if (!owner_) return;
  double s[8];
  state(s);
  sim3opt_set_vertex(owner_->g_, id_, s);  // warm start, kitti_surf.cpp:926-932, :1037-1038
}
inline bool Vertex::pull(double s[8]) const {
  return owner_ && sim3opt_get_vertex(owner_->g_, id_, s) == SIM3OPT_OK;
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
using sim3opt_shim::G2oEdgeScale;
using sim3opt_shim::G2oEdgeScaleTrans;
using sim3opt_shim::G2oVertexScale;
using sim3opt_shim::G2oVertexScaleTrans;
using sim3opt_shim::VertexSim3Expmap;
}  // namespace vio
#endif
