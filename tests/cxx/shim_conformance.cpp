// shim_conformance.cpp -- conformance test of include/sim3opt_g2o.hpp, the g2o-named binding of
// libsim3opt (the complete list of methods the reference's callers use).
//
// Independently written: one short block per method on a five-vertex ring, asserting on results, then
// (on a GPU box) a solve of that ring and an end-to-end run of the KITTI-00 fixture taken from the
// library's own loader.  The reference lines a block answers to are cited in comments only
// (kitti_surf.cpp / kittiDetector.h of /root/reference).
//
//   shim_conformance host                               value types + container semantics (no GPU)
//   shim_conformance ring                               five-vertex ring through initialize / optimize (GPU)
//   shim_conformance kitti <dir> <out prefix> <one>     direct + staged runs on the fixture (GPU)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <sophus/se3.hpp>

#include "sim3opt_g2o.hpp"

namespace {

int g_failed = 0, g_checked = 0;
void expect(bool ok, const char* what, int line) {
  ++g_checked;
  if (!ok) {
    ++g_failed;
    std::fprintf(stderr, "FAILED line %d: %s\n", line, what);
  }
}
#define EXPECT(cond) expect((cond), #cond, __LINE__)
bool near(double a, double b, double tol = 1e-14) { return std::fabs(a - b) <= tol * (1.0 + std::fabs(b)); }

// rotation about a unit axis, as a 3 x 3 matrix and as a quaternion
Eigen::Matrix3d axis_rotation(double ax, double ay, double az, double angle) {
  const double c = std::cos(angle), s = std::sin(angle), k = 1.0 - c;
  Eigen::Matrix3d R;
  R(0, 0) = c + ax * ax * k;      R(0, 1) = ax * ay * k - az * s; R(0, 2) = ax * az * k + ay * s;
  R(1, 0) = ay * ax * k + az * s; R(1, 1) = c + ay * ay * k;      R(1, 2) = ay * az * k - ax * s;
  R(2, 0) = az * ax * k - ay * s; R(2, 1) = az * ay * k + ax * s; R(2, 2) = c + az * az * k;
  return R;
}
Eigen::Quaterniond axis_quaternion(double ax, double ay, double az, double angle) {
  const double h = std::sin(0.5 * angle);
  return Eigen::Quaterniond(std::cos(0.5 * angle), ax * h, ay * h, az * h);
}

// ---------------------------------------------------------------------------------------------
// g2o::Sim3 as a value type
// ---------------------------------------------------------------------------------------------
void sim3_value_type() {
  const double inv3 = 1.0 / std::sqrt(3.0);
  const Eigen::Matrix3d R = axis_rotation(inv3, inv3, inv3, 0.7);
  const Eigen::Vector3d t(0.5, -1.25, 2.0);

  // Sim3(R, t, s) from a 3 x 3 matrix                                  (kitti_surf.cpp:199-200, :608)
  const g2o::Sim3 fromR(R, t, 1.5);
  const Eigen::Quaterniond want = axis_quaternion(inv3, inv3, inv3, 0.7);
  EXPECT(near(fromR.rotation().x(), want.x()) && near(fromR.rotation().y(), want.y()));
  EXPECT(near(fromR.rotation().z(), want.z()) && near(fromR.rotation().w(), want.w()));
  EXPECT(fromR.scale() == 1.5);
  EXPECT(fromR.translation()[0] == 0.5 && fromR.translation()[1] == -1.25 && fromR.translation()[2] == 2.0);

  // Sim3(q, t, s) from a quaternion; a non-unit one is normalised          (kitti_surf.cpp:1035)
  const g2o::Sim3 fromQ(Eigen::Quaterniond(2 * want.w(), 2 * want.x(), 2 * want.y(), 2 * want.z()), t, 1.5);
  for (int i = 0; i < 8; ++i) EXPECT(near(fromQ.v[i], fromR.v[i]));

  // rotation().coeffs() is (x, y, z, w) and streams through transpose()    (kitti_surf.cpp:698)
  std::ostringstream os;
  os << fromR.rotation().coeffs().transpose();
  double c[4] = {0, 0, 0, 0};
  std::istringstream is(os.str());
  is >> c[0] >> c[1] >> c[2] >> c[3];
  EXPECT(near(c[0], want.x(), 1e-5) && near(c[3], want.w(), 1e-5));  // (default stream precision)

  // inverse() and operator*: S S^-1 is the identity; (A B) maps like A(B(.))   (kitti_surf.cpp:653-660, :691)
  const g2o::Sim3 I = fromR * fromR.inverse();
  EXPECT(near(I.rotation().w(), 1.0) && near(I.scale(), 1.0));
  for (int i = 0; i < 3; ++i) EXPECT(std::fabs(I.translation()[i]) < 1e-14 && std::fabs(I.v[i]) < 1e-15);
  const g2o::Sim3 other(axis_rotation(0, 0, 1, -0.3), Eigen::Vector3d(1, 2, 3), 0.5);
  const Eigen::Vector3d p(0.1, 0.2, -0.3);
  const auto lhs = (fromR * other).map(p);
  const auto rhs = fromR.map(other.map(p));
  for (int i = 0; i < 3; ++i) EXPECT(near(lhs[i], rhs[i]));
  // map is x -> s R x + t
  const Eigen::Vector3d Rp = R * p;
  for (int i = 0; i < 3; ++i) EXPECT(near(fromR.map(p)[i], 1.5 * Rp[i] + t[i]));
  // translation() / scale(): an expression the callers form                 (kitti_surf.cpp:693)
  const auto scaled = fromR.translation() / fromR.scale();
  EXPECT(near(scaled[1], -1.25 / 1.5));
}

// ---------------------------------------------------------------------------------------------
// information() = M for the three edge kinds; frozen rotation holder; solver-stack tags
// ---------------------------------------------------------------------------------------------
void small_types() {
  // 7 x 7: arrives column-major                                            (kitti_surf.cpp:592, :637)
  Eigen::Matrix<double, 7, 7> M7 = Eigen::Matrix<double, 7, 7>::Identity();
  M7(2, 5) = 0.25;
  M7(5, 2) = 0.25;
  double store[49];
  bool touched = false;
  sim3opt_shim::InformationRef<7>(store, &touched) = M7;
  EXPECT(touched && store[0] == 1.0 && store[7 * 5 + 2] == 0.25 && store[7 * 2 + 5] == 0.25 && store[1] == 0.0);
  // 1 x 1 and 4 x 4                                                        (kitti_surf.cpp:832, :839)
  vio::G2oEdgeScale es;
  Eigen::Matrix<double, 1, 1> M1;
  M1(0, 0) = 4.0;
  es.information() = M1;
  vio::G2oEdgeScaleTrans est;
  Eigen::Matrix<double, 4, 4> M4 = Eigen::Matrix<double, 4, 4>::Identity() * 2.0;
  est.information() = M4;
  est.information()(0, 0) = 3.0;  // element access on the same proxy

  // Rw2i accepts an SO3d, a quaternion, a matrix                           (kitti_surf.cpp:793)
  vio::G2oVertexScaleTrans vst;
  const Eigen::Matrix3d R = axis_rotation(1, 0, 0, 0.4);
  vst.Rw2i = Sophus::SO3d(R);
  const double w = vst.Rw2i.unit_quaternion().w(), x = vst.Rw2i.unit_quaternion().x();
  EXPECT(near(w, std::cos(0.2)) && near(x, std::sin(0.2)));
  vst.Rw2i = axis_quaternion(1, 0, 0, 0.4);
  EXPECT(near(vst.Rw2i.unit_quaternion().x(), x));
  vst.Rw2i = R;
  EXPECT(near(vst.Rw2i.unit_quaternion().w(), w));

  // [s, t] estimate of the scale + translation vertex; tail<3>() is t     (kitti_surf.cpp:533-538, :1034)
  Eigen::Vector4d st;
  st[0] = 2.0; st[1] = 1.0; st[2] = -1.0; st[3] = 0.5;
  vst.setEstimate(st);
  EXPECT(vst.estimate()[0] == 2.0 && vst.estimate().tail<3>()[2] == 0.5);

  // solver stack: the tags compile, own each other, carry the two LM settings   (kitti_surf.cpp:553-558,
  // kittiDetector.h:730, :779-782)
  auto linear = g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType>>();
  std::unique_ptr<g2o::BlockSolverX::LinearSolverType> as_base = std::move(linear);
  auto* lm = new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<g2o::BlockSolverX>(std::move(as_base)));
  lm->setUserLambdaInit(50.0);
  lm->setMaxTrialsAfterFailure(5);
  g2o::SparseOptimizer opt;
  opt.setAlgorithm(lm);  // owned from here on
  sim3opt_options o;
  sim3opt_get_options(opt.handle(), &o);
  EXPECT(o.user_lambda_init == 50.0 && o.max_trials == 5);
  opt.setVerbose(true);
  sim3opt_get_options(opt.handle(), &o);
  EXPECT(o.verbose == 1);
  opt.setVerbose(false);
}

// ---------------------------------------------------------------------------------------------
// a ring of five poses: v_i at angle 72 i degrees on the unit circle, looking along the tangent
// ---------------------------------------------------------------------------------------------
struct Ring {
  static constexpr int N = 5;
  g2o::Sim3 truth[N], guess[N];
  Ring() {
    for (int i = 0; i < N; ++i) {
      const double a = 2.0 * M_PI * i / N;
      const g2o::Sim3 Swi(axis_rotation(0, 0, 1, a), Eigen::Vector3d(std::cos(a), std::sin(a), 0.1 * i), 1.0);
      truth[i] = Swi.inverse();  // the estimates are S_iw
      // the guess drifts in scale and position
      const g2o::Sim3 drift(axis_rotation(0, 1, 0, 0.01 * i), Eigen::Vector3d(0.02 * i, -0.01 * i, 0.0), 1.0 + 0.03 * i);
      guess[i] = drift * truth[i];
    }
  }
  // constraint of an edge (v0 = a, v1 = b): e = log(C S_a S_b^-1) vanishes at the truth
  g2o::Sim3 constraint(int a, int b) const { return truth[b] * truth[a].inverse(); }
};

// builds the ring in `opt`: vertex 0 fixed, edges (i, i-1) and the closing edge (0, N-1)
void build_ring(g2o::SparseOptimizer& opt, const Ring& ring, bool with_kernel) {
  for (int i = 0; i < Ring::N; ++i) {
    auto* v = new vio::VertexSim3Expmap();
    v->setId(10 * i);  // ids need not be dense
    v->setEstimate(ring.guess[i]);
    v->setFixed(i == 0);
    v->setMarginalized(false);
    EXPECT(opt.addVertex(v));
  }
  Eigen::Matrix<double, 7, 7> info = Eigen::Matrix<double, 7, 7>::Identity();
  for (int i = 0; i < Ring::N; ++i) {
    const int a = (i + 1) % Ring::N, b = i;
    auto* e = new vio::EdgeSim3();
    e->setVertex(0, opt.vertex(10 * a));
    e->setVertex(1, opt.vertex(10 * b));
    e->setMeasurement(ring.constraint(a, b));
    e->information() = info;
    if (with_kernel && i == 2) e->setRobustKernelHuber(1.0);
    EXPECT(opt.addEdge(e));
  }
}

void container_semantics() {
  const Ring ring;
  g2o::SparseOptimizer opt;
  opt.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<g2o::BlockSolverX>(g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType>>())));
  build_ring(opt, ring, false);

  // vertex(id): the object that was added, or null                        (kitti_surf.cpp:634-635, :688)
  EXPECT(opt.vertex(20) != nullptr && opt.vertex(20)->id() == 20);
  EXPECT(opt.vertex(7) == nullptr);
  EXPECT(opt.vertex(0)->fixed() && !opt.vertex(10)->fixed());
  // what the library holds: five vertices, five edges, endpoints in slot order, measurement as given
  EXPECT(sim3opt_num_vertices(opt.handle()) == 5 && sim3opt_num_edges(opt.handle()) == 5);
  int32_t a = -1, b = -1;
  double m[8];
  EXPECT(sim3opt_get_edge(opt.handle(), 1, &a, &b, m) == SIM3OPT_OK && a == 20 && b == 10);
  const g2o::Sim3 c12 = ring.constraint(2, 1);
  for (int i = 0; i < 8; ++i) EXPECT(m[i] == c12.v[i]);
  // estimate() before any optimisation is what was set                    (kitti_surf.cpp:689)
  const g2o::Sim3 e3 = static_cast<vio::VertexSim3Expmap*>(opt.vertex(30))->estimate();
  for (int i = 0; i < 8; ++i) EXPECT(e3.v[i] == ring.guess[3].v[i]);
  // setEstimate after addVertex is a warm start: the library sees it       (kitti_surf.cpp:1037-1038)
  static_cast<vio::VertexSim3Expmap*>(opt.vertex(30))->setEstimate(ring.truth[3]);
  double s[8];
  EXPECT(sim3opt_get_vertex(opt.handle(), 30, s) == SIM3OPT_OK);
  for (int i = 0; i < 8; ++i) EXPECT(s[i] == ring.truth[3].v[i]);

  // refusals: a second vertex with the same id, an edge with a missing endpoint, a vertex of another
  // kind in a Sim(3) optimizer
  auto* dup = new vio::VertexSim3Expmap();
  dup->setId(20);
  EXPECT(!opt.addVertex(dup));
  auto* dangling = new vio::EdgeSim3();
  dangling->setVertex(0, opt.vertex(10));
  EXPECT(!opt.addEdge(dangling));
  auto* wrong_kind = new vio::G2oVertexScale();
  wrong_kind->setId(99);
  EXPECT(!opt.addVertex(wrong_kind));
  EXPECT(sim3opt_num_vertices(opt.handle()) == 5 && sim3opt_num_edges(opt.handle()) == 5);

  // an optimizer of scale vertices freezes everything but sigma, one of scale + translation vertices
  // freezes the rotations                                                  (kitti_surf.cpp:779-793, :809-814)
  g2o::SparseOptimizer scales, scale_trans;
  auto* vs = new vio::G2oVertexScale();
  vs->setId(0);
  vs->setEstimate(1.25);
  EXPECT(scales.addVertex(vs));
  auto* vt = new vio::G2oVertexScaleTrans();
  vt->setId(0);
  EXPECT(scale_trans.addVertex(vt));
  sim3opt_options o;
  sim3opt_get_options(scales.handle(), &o);
  EXPECT(o.dof_mask == 0x40);
  sim3opt_get_options(scale_trans.handle(), &o);
  EXPECT(o.dof_mask == 0x78);
  sim3opt_get_options(opt.handle(), &o);
  EXPECT(o.dof_mask == 127);
  EXPECT(static_cast<vio::G2oVertexScale*>(scales.vertex(0))->estimate() == 1.25);
}

// composing the odometry constraint on the caller's side (S_jw S_iw^-1) gives what the library's
// loader stores for the same pair                                          (kitti_surf.cpp:649-668)
void composed_constraints_match_loader(const char* dir) {
  sim3opt_graph* src = sim3opt_create();
  if (sim3opt_load_kitti_direct(src, dir, 1) != SIM3OPT_OK) {
    std::fprintf(stderr, "cannot load %s: %s\n", dir, sim3opt_last_error(src));
    ++g_failed;
    sim3opt_destroy(src);
    return;
  }
  const int ne = sim3opt_num_edges(src);
  double worst = 0.0;
  for (int k = 1; k < ne; k += 37) {  // (edge 0 is the loop; the rest is the chain)
    int32_t a, b;
    double m[8], sa[8], sb[8];
    sim3opt_get_edge(src, k, &a, &b, m);
    sim3opt_get_vertex(src, a, sa);
    sim3opt_get_vertex(src, b, sb);
    const g2o::Sim3 C = g2o::Sim3(sb) * g2o::Sim3(sa).inverse();
    for (int i = 0; i < 8; ++i) worst = std::fmax(worst, std::fabs(C.v[i] - m[i]));
  }
  EXPECT(worst < 1e-13);
  sim3opt_destroy(src);
}

// ---------------------------------------------------------------------------------------------
// GPU: the ring through initializeOptimization() / optimize()
// ---------------------------------------------------------------------------------------------
int solve_ring() {
  const Ring ring;
  g2o::SparseOptimizer opt;
  opt.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<g2o::BlockSolverX>(g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType>>())));
  build_ring(opt, ring, true);
  {  // the well-posed arithmetic: with the reference's as-written small-angle coefficient and its
     // delta = 1e-9 differences LM stalls near an exact solution
    sim3opt_options o;
    sim3opt_get_options(opt.handle(), &o);
    o.fix_small_angle_b = 1;
    o.fd_delta = 1e-6;
    sim3opt_set_options(opt.handle(), &o);
  }
  if (!opt.initializeOptimization()) {  // (kitti_surf.cpp:674)
    std::fprintf(stderr, "initializeOptimization: %s\n", opt.lastError());
    return 3;
  }
  opt.computeActiveErrors();
  const double before = opt.activeChi2();
  EXPECT(before > 1e-3 && opt.chi2() == before && opt.activeRobustChi2() == before);
  const int done = opt.optimize(30);  // (kitti_surf.cpp:675: the count of iterations performed)
  EXPECT(done >= 1 && done <= 30);
  const double after = opt.activeChi2();
  EXPECT(after < 1e-10 * before);
  // the fixed vertex did not move; the others reached the truth (the constraints are exact)
  double worst = 0.0;
  for (int i = 0; i < Ring::N; ++i) {
    const g2o::Sim3 est = static_cast<vio::VertexSim3Expmap*>(opt.vertex(10 * i))->estimate();
    const g2o::Sim3 d = est * ring.truth[i].inverse();
    worst = std::fmax(worst, std::fabs(d.scale() - 1.0));
    for (int c = 0; c < 3; ++c) worst = std::fmax(worst, std::fmax(std::fabs(d.translation()[c]), std::fabs(d.v[c])));
    if (i == 0)
      for (int c = 0; c < 8; ++c) EXPECT(est.v[c] == ring.guess[0].v[c]);
  }
  // (vertex 0 is fixed at its *guess*, which is the truth for i = 0: no drift there)
  EXPECT(worst < 1e-6);
  std::printf("ring: chi2 %.6g -> %.3g after %d iterations, worst deviation %.2e\n", before, after, done, worst);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// GPU: the KITTI-00 fixture end to end.  Poses and constraints come from the library's loader; the
// three optimisation pipelines of the reference (all of Sim(3) at once; scales, then scale +
// translation; the same followed by Sim(3)) are built through the g2o-named classes.
// ---------------------------------------------------------------------------------------------
struct Link {
  int from, to;  // vertex slots 0 and 1
  g2o::Sim3 rel;
};
struct Fixture {
  std::vector<g2o::Sim3> pose;  // S_iw with unit scale, insertion order = vertex id
  std::vector<Link> links;      // loop closures first, then the chain
};

bool read_fixture(const char* dir, bool one_loop, Fixture& out) {
  sim3opt_graph* src = sim3opt_create();
  const bool ok = sim3opt_load_kitti_direct(src, dir, one_loop ? 1 : 0) == SIM3OPT_OK;
  if (!ok) std::fprintf(stderr, "cannot load %s: %s\n", dir, sim3opt_last_error(src));
  for (int i = 0; ok && i < sim3opt_num_vertices(src); ++i) {
    double s[8];
    sim3opt_get_vertex(src, i, s);
    out.pose.emplace_back(s);
  }
  for (int k = 0; ok && k < sim3opt_num_edges(src); ++k) {
    Link l;
    double m[8];
    int32_t a, b;
    sim3opt_get_edge(src, k, &a, &b, m);
    l.from = a;
    l.to = b;
    l.rel = g2o::Sim3(m);
    out.links.push_back(l);
  }
  sim3opt_destroy(src);
  return ok;
}

std::unique_ptr<g2o::SparseOptimizer> fresh_optimizer() {
  std::unique_ptr<g2o::SparseOptimizer> opt(new g2o::SparseOptimizer());
  opt->setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<g2o::BlockSolverX>(g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType>>())));
  return opt;
}

// output format of the reference's result files: id, scale of S_iw, translation and rotation (x y z w)
// of S_wi                                                                 (kitti_surf.cpp:695-698)
void write_result(const std::string& path, const std::vector<g2o::Sim3>& Siw) {
  std::ofstream f(path);
  f.precision(17);
  f << "% id, scale(S_iw), t(S_wi), q(S_wi) xyzw" << std::endl;
  for (size_t i = 0; i < Siw.size(); ++i) {
    const g2o::Sim3 Swi = Siw[i].inverse();
    f << i << " " << Siw[i].scale() << " " << Swi.translation().transpose() << " "
      << Swi.rotation().coeffs().transpose() << std::endl;
  }
}

Eigen::Vector4d scale_and_translation(const g2o::Sim3& S) {
  Eigen::Vector4d v;
  v[0] = S.scale();
  for (int i = 0; i < 3; ++i) v[1 + i] = S.translation()[i];
  return v;
}

int run_all_at_once(const Fixture& fx, const std::string& out_path) {
  auto opt = fresh_optimizer();
  for (size_t i = 0; i < fx.pose.size(); ++i) {
    auto* v = new vio::VertexSim3Expmap();
    v->setId((int)i);
    v->setFixed(i == 0);
    v->setEstimate(fx.pose[i]);
    opt->addVertex(v);
  }
  for (const Link& l : fx.links) {
    auto* e = new vio::EdgeSim3();
    e->setVertex(0, opt->vertex(l.from));
    e->setVertex(1, opt->vertex(l.to));
    e->setMeasurement(l.rel);
    opt->addEdge(e);  // (no information() call: identity)
  }
  if (!opt->initializeOptimization()) {
    std::fprintf(stderr, "initializeOptimization: %s\n", opt->lastError());
    return 3;
  }
  const double before = opt->activeChi2();
  const int done = opt->optimize(100);
  std::vector<g2o::Sim3> result;
  for (size_t i = 0; i < fx.pose.size(); ++i)
    result.push_back(static_cast<vio::VertexSim3Expmap*>(opt->vertex((int)i))->estimate());
  write_result(out_path, result);
  std::printf("all-at-once: chi2 %.10g -> %.10g after %d iterations\n", before, opt->activeChi2(), done);
  return done > 0 ? 0 : 3;
}

// stages: 1 = scales from the null vector of the scale equations, 2 = scale + translation LM with the
// rotations frozen, 3 (optional) = Sim(3) LM warm-started from stage 2     (kitti_surf.cpp:887-934,
// :1020-1047)
int run_staged(const Fixture& fx, bool finish_with_sim3, const std::string& out_path) {
  auto scales = fresh_optimizer(), scale_trans = fresh_optimizer(), full = fresh_optimizer();
  const size_t n = fx.pose.size();
  for (size_t i = 0; i < n; ++i) {
    auto* a = new vio::G2oVertexScale();
    a->setId((int)i);
    a->setFixed(i == 0);
    a->setEstimate(fx.pose[i].scale());
    scales->addVertex(a);
    auto* b = new vio::G2oVertexScaleTrans();
    b->setId((int)i);
    b->setFixed(i == 0);
    b->Rw2i = fx.pose[i].rotation();
    b->setEstimate(scale_and_translation(fx.pose[i]));
    scale_trans->addVertex(b);
    if (finish_with_sim3) {
      auto* c = new vio::VertexSim3Expmap();
      c->setId((int)i);
      c->setFixed(i == 0);
      c->setEstimate(fx.pose[i]);
      full->addVertex(c);
    }
  }
  for (const Link& l : fx.links) {
    auto* a = new vio::G2oEdgeScale();
    a->setVertex(0, scales->vertex(l.from));
    a->setVertex(1, scales->vertex(l.to));
    a->setMeasurement(l.rel.scale());
    scales->addEdge(a);
    auto* b = new vio::G2oEdgeScaleTrans();
    b->setVertex(0, scale_trans->vertex(l.from));
    b->setVertex(1, scale_trans->vertex(l.to));
    b->setMeasurement(scale_and_translation(l.rel));
    scale_trans->addEdge(b);
    if (finish_with_sim3) {
      auto* c = new vio::EdgeSim3();
      c->setVertex(0, full->vertex(l.from));
      c->setVertex(1, full->vertex(l.to));
      c->setMeasurement(l.rel);
      full->addEdge(c);
    }
  }
  // stage 1 (the reference: an Eigen::JacobiSVD in the caller; here the library's own routine)
  double ratio = 0.0;
  if (sim3opt_stepwise_scale_init(scales->handle(), &ratio) != SIM3OPT_OK) {
    std::fprintf(stderr, "scale init: %s\n", scales->lastError());
    return 3;
  }
  for (size_t i = 0; i < n; ++i) {
    auto* b = static_cast<vio::G2oVertexScaleTrans*>(scale_trans->vertex((int)i));
    Eigen::Vector4d st = b->estimate();
    st[0] = static_cast<vio::G2oVertexScale*>(scales->vertex((int)i))->estimate();
    b->setEstimate(st);
  }
  // stage 2
  if (!scale_trans->initializeOptimization()) return 3;
  const double st_before = scale_trans->activeChi2();
  scale_trans->optimize(100);
  const double st_after = scale_trans->activeChi2();
  std::vector<g2o::Sim3> result(n);
  for (size_t i = 0; i < n; ++i) {
    auto* b = static_cast<vio::G2oVertexScaleTrans*>(scale_trans->vertex((int)i));
    const Eigen::Vector4d st = b->estimate();
    result[i] = g2o::Sim3(b->Rw2i.unit_quaternion(), st.tail<3>(), st[0]);
  }
  double final_chi = st_after;
  if (finish_with_sim3) {  // stage 3
    for (size_t i = 0; i < n; ++i)
      static_cast<vio::VertexSim3Expmap*>(full->vertex((int)i))->setEstimate(result[i]);
    if (!full->initializeOptimization()) return 3;
    full->optimize(100);
    final_chi = full->activeChi2();
    for (size_t i = 0; i < n; ++i)
      result[i] = static_cast<vio::VertexSim3Expmap*>(full->vertex((int)i))->estimate();
  }
  write_result(out_path, result);
  std::printf("staged (%d stages): sigma ratio %.3g, scale+translation chi2 %.10g -> %.10g, final chi2 %.10g\n",
              finish_with_sim3 ? 3 : 2, ratio, st_before, st_after, final_chi);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "";
  int rc = 0;
  if (mode == "host") {
    sim3_value_type();
    small_types();
    container_semantics();
    if (argc > 2) composed_constraints_match_loader(argv[2]);
  } else if (mode == "ring") {
    rc = solve_ring();
  } else if (mode == "kitti" && argc >= 5) {
    Fixture fx;
    if (!read_fixture(argv[2], std::atoi(argv[4]) != 0, fx)) return 1;
    const std::string prefix = argv[3];
    rc = run_all_at_once(fx, prefix + "all_at_once.txt");
    if (!rc) rc = run_staged(fx, false, prefix + "staged_2.txt");
    if (!rc) rc = run_staged(fx, true, prefix + "staged_3.txt");
  } else {
    std::fprintf(stderr, "usage: %s host [fixture dir] | ring | kitti <dir> <out prefix> <one_loop>\n", argv[0]);
    return 2;
  }
  if (rc) return rc;
  std::printf("%d checks, %d failed\n", g_checked, g_failed);
  return g_failed ? 1 : 0;
}
