// ba_shim_conformance.cpp -- conformance test of include/sim3opt_g2o_ba.hpp, the g2o-named binding of
// the bundle-adjustment hand-off (sim3opt_ba_*; the reference's ba_demo, bal_example.cpp:44-243).
//
// Independently written: one block per class / method the demo touches, on a three-camera toy scene
// with exact observations, asserting on results; then (on a GPU box) the toy scene and a BAL file are
// optimised through the binding.  Reference lines are cited in comments only.
//
//   ba_shim_conformance host                              value types + container semantics (no GPU)
//   ba_shim_conformance toy                               toy scene through initialize / optimize (GPU)
//   ba_shim_conformance bal <problem.bal> <poses out> [iterations=5]      a BAL file end to end (GPU)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "sim3opt_g2o_ba.hpp"

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
bool near(double a, double b, double tol = 1e-13) { return std::fabs(a - b) <= tol * (1.0 + std::fabs(b)); }

const double kFocal = 718.856, kCx = 607.1928, kCy = 185.2157;  // (kitti_surf.cpp:52-57)

Eigen::Quaterniond yaw(double angle) { return Eigen::Quaterniond(std::cos(0.5 * angle), 0.0, std::sin(0.5 * angle), 0.0); }

// ---------------------------------------------------------------------------------------------
// toy scene: three cameras on the x axis looking down +z, a 3 x 4 wall of points at depth ~8
// ---------------------------------------------------------------------------------------------
struct Toy {
  std::vector<g2o::SE3Quat> cam;        // T_w2c
  std::vector<Eigen::Vector3d> point;   // world
  struct Obs { int c, p; double u, v; };
  std::vector<Obs> obs;
  Toy() {
    for (int c = 0; c < 3; ++c) cam.emplace_back(yaw(0.02 * (c - 1)), Eigen::Vector3d(-0.6 * c, 0.0, 0.0));
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 4; ++q) point.emplace_back(-1.5 + q, -1.0 + r, 8.0 + 0.3 * q - 0.2 * r);
    for (int c = 0; c < 3; ++c)
      for (size_t p = 0; p < point.size(); ++p) {
        const auto X = cam[c].map(point[p]);
        obs.push_back({c, (int)p, kFocal * X[0] / X[2] + kCx, kFocal * X[1] / X[2] + kCy});
      }
  }
};

struct Built {
  std::unique_ptr<g2o::SparseOptimizer> opt;
  std::vector<g2o::VertexSE3Expmap*> cams;
  std::vector<g2o::VertexSBAPointXYZ*> points;
  std::vector<g2o::EdgeProjectXYZ2UV*> edges;
};

// cameras get ids 0.., points follow; `jitter` perturbs the point estimates
Built build(const Toy& toy, double jitter, double huber_delta) {
  Built b;
  b.opt.reset(new g2o::SparseOptimizer());
  b.opt->setVerbose(false);
  std::unique_ptr<g2o::BlockSolver_6_3::LinearSolverType> linear =
      g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>>();
  b.opt->setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<g2o::BlockSolver_6_3>(std::move(linear))));
  auto* K = new g2o::CameraParameters(kFocal, Eigen::Vector2d(kCx, kCy), 0.0);
  K->setId(0);
  EXPECT(b.opt->addParameter(K));
  int id = 0;
  for (const g2o::SE3Quat& T : toy.cam) {
    auto* v = new g2o::VertexSE3Expmap();
    v->setId(id++);
    v->setEstimate(T);
    EXPECT(b.opt->addVertex(v));
    b.cams.push_back(v);
  }
  for (size_t p = 0; p < toy.point.size(); ++p) {
    auto* v = new g2o::VertexSBAPointXYZ();
    v->setId(id++);
    v->setMarginalized(true);
    const double d = jitter * ((int)(p % 3) - 1);
    v->setEstimate(Eigen::Vector3d(toy.point[p][0] + d, toy.point[p][1] - d, toy.point[p][2] + 2 * d));
    EXPECT(b.opt->addVertex(v));
    b.points.push_back(v);
  }
  for (const Toy::Obs& o : toy.obs) {
    auto* e = new g2o::EdgeProjectXYZ2UV();
    e->setVertex(0, b.points[o.p]);
    e->setVertex(1, b.cams[o.c]);
    e->setMeasurement(Eigen::Vector2d(o.u, o.v));
    e->setInformation(Eigen::Matrix2d::Identity());
    if (huber_delta > 0) {
      auto* rk = new g2o::RobustKernelHuber;
      rk->setDelta(huber_delta);
      e->setRobustKernel(rk);
    }
    EXPECT(e->setParameterId(0, 0));
    EXPECT(b.opt->addEdge(e));
    b.edges.push_back(e);
  }
  return b;
}

void value_types() {
  // SE3Quat(q, t): the rotation is normalised with w >= 0; map is R p + t; inverse undoes it  (:172)
  const Eigen::Quaterniond q = yaw(0.3);
  const g2o::SE3Quat T(Eigen::Quaterniond(-2 * q.w(), -2 * q.x(), -2 * q.y(), -2 * q.z()), Eigen::Vector3d(1, 2, 3));
  EXPECT(near(T.rotation().w(), q.w()) && near(T.rotation().y(), q.y()));
  const Eigen::Vector3d p(0.5, -0.25, 4.0);
  const auto Tp = T.map(p);
  EXPECT(near(Tp[0], std::cos(0.3) * 0.5 + std::sin(0.3) * 4.0 + 1.0));
  EXPECT(near(Tp[1], -0.25 + 2.0));
  EXPECT(near(Tp[2], -std::sin(0.3) * 0.5 + std::cos(0.3) * 4.0 + 3.0));
  const auto back = T.inverse().map(Tp);
  for (int i = 0; i < 3; ++i) EXPECT(near(back[i], p[i]));
  EXPECT(T.translation()[2] == 3.0);
  // what the demo writes per camera: centre -R^T t and the conjugate rotation              (:229-236)
  const Eigen::Quaterniond qc = T.rotation().conjugate();
  const Eigen::Vector3d centre = -qc._transformVector(Eigen::Vector3d(T.translation()[0], T.translation()[1], T.translation()[2]));
  const auto centre2 = T.inverse().translation();
  for (int i = 0; i < 3; ++i) EXPECT(near(centre[i], centre2[i]));

  // CameraParameters: pinhole projection, g2o's member names                               (:93-94)
  g2o::CameraParameters K(kFocal, Eigen::Vector2d(kCx, kCy), 0.0);
  K.setId(4);
  EXPECT(K.id() == 4 && K.focal_length == kFocal && K.principle_point[1] == kCy && K.baseline == 0.0);
  const auto uv = K.cam_map(Eigen::Vector3d(1.0, -2.0, 10.0));
  EXPECT(near(uv[0], kFocal * 0.1 + kCx) && near(uv[1], -kFocal * 0.2 + kCy));

  // RobustKernelHuber: delta                                                               (:150-152)
  g2o::RobustKernelHuber rk;
  rk.setDelta(2.5);
  EXPECT(rk.delta() == 2.5);

  // vertex objects: dimension, flags, estimates
  g2o::VertexSE3Expmap c;
  g2o::VertexSBAPointXYZ x;
  EXPECT(c.dimension() == 6 && x.dimension() == 3);
  x.setMarginalized(true);
  c.setFixed(true);
  EXPECT(x.marginalized() && !x.fixed() && c.fixed() && !c.marginalized());
  x.setEstimate(p);
  EXPECT(x.estimate()[2] == 4.0);
}

void container_semantics() {
  const Toy toy;
  Built b = build(toy, 0.0, 2.5);
  g2o::SparseOptimizer& opt = *b.opt;
  // vertex(id) / vertices()                                                               (:115, :125)
  EXPECT(opt.vertex(1) == b.cams[1] && opt.vertex(3) == b.points[0] && opt.vertex(99) == nullptr);
  EXPECT(opt.vertices().size() == 15);
  // computeError() on exact observations: zero; after moving the point: the projected shift  (:156-160)
  b.edges[0]->computeError();
  EXPECT(b.edges[0]->error().norm() < 1e-10);
  const Eigen::Vector3d moved(toy.point[0][0] + 0.1, toy.point[0][1], toy.point[0][2]);
  b.points[0]->setEstimate(moved);
  b.edges[0]->computeError();
  const auto X = toy.cam[0].map(moved);
  EXPECT(near(b.edges[0]->error()[0], toy.obs[0].u - (kFocal * X[0] / X[2] + kCx), 1e-10));
  b.points[0]->setEstimate(toy.point[0]);

  // refusals: second parameter block / vertex with a used id, vertex without id, edge without its
  // parameter block, parameter slot other than 0
  auto* K2 = new g2o::CameraParameters(1.0, Eigen::Vector2d(0, 0), 0.0);
  K2->setId(0);
  EXPECT(!opt.addParameter(K2));
  auto* dup = new g2o::VertexSE3Expmap();
  dup->setId(2);
  EXPECT(!opt.addVertex(dup));
  auto* anonymous = new g2o::VertexSBAPointXYZ();
  EXPECT(!opt.addVertex(anonymous));
  auto* orphan = new g2o::EdgeProjectXYZ2UV();
  orphan->setVertex(0, b.points[1]);
  orphan->setVertex(1, b.cams[1]);
  EXPECT(!orphan->setParameterId(1, 0));
  EXPECT(orphan->setParameterId(0, 7));  // no parameter block 7
  EXPECT(!opt.addEdge(orphan));
  auto* half = new g2o::EdgeProjectXYZ2UV();
  half->setVertex(0, b.points[1]);
  half->setParameterId(0, 0);
  EXPECT(!opt.addEdge(half));
  // LM settings travel with the algorithm object                          (kittiDetector.h:730, :779-782)
  g2o::SparseOptimizer other;
  auto* lm = new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<g2o::BlockSolver_6_3>(
      g2o::make_unique<g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>>()));
  lm->setUserLambdaInit(50.0);
  lm->setMaxTrialsAfterFailure(5);
  other.setAlgorithm(lm);
  EXPECT(lm->user_lambda_init == 50.0 && lm->max_trials == 5);

  // what the binding cannot express is refused at initializeOptimization() with a reason
  {
    Built c = build(toy, 0.0, 0.0);
    c.edges[5]->setInformation(Eigen::Matrix2d::Identity() * 4.0);
    EXPECT(!c.opt->initializeOptimization() && std::string(c.opt->lastError()).find("different information") != std::string::npos);
  }
  {
    Built c = build(toy, 0.0, 2.5);
    c.edges[7]->setRobustKernel(nullptr);
    EXPECT(!c.opt->initializeOptimization() && std::string(c.opt->lastError()).find("robust kernels") != std::string::npos);
  }
  {
    Built c = build(toy, 0.0, 0.0);
    c.points[2]->setFixed(true);
    EXPECT(!c.opt->initializeOptimization() && std::string(c.opt->lastError()).find("fixed points") != std::string::npos);
  }
  {
    g2o::SparseOptimizer empty;
    EXPECT(!empty.initializeOptimization() && empty.optimize(1) == -1);
  }
}

// GPU: perturbed points, first camera fixed; the optimisation returns to the exact scene
int solve_toy() {
  const Toy toy;
  Built b = build(toy, 0.05, 2.5);
  b.cams[0]->setFixed(true);
  b.cams[1]->setFixed(true);  // two fixed cameras pin the gauge (scale included)
  if (!b.opt->initializeOptimization()) {
    std::fprintf(stderr, "initializeOptimization failed: %s\n", b.opt->lastError());
    return 3;
  }
  b.opt->computeActiveErrors();
  const double before = b.opt->activeRobustChi2();
  EXPECT(before > 1.0 && b.opt->chi2() == before);
  const int done = b.opt->optimize(20);
  if (done <= 0) {
    std::fprintf(stderr, "optimize failed: %s\n", b.opt->lastError());
    return 3;
  }
  const double after = b.opt->activeRobustChi2();
  EXPECT(after < 1e-10 * before);
  double worst = 0.0;
  for (size_t p = 0; p < toy.point.size(); ++p)
    for (int i = 0; i < 3; ++i) worst = std::fmax(worst, std::fabs(b.points[p]->estimate()[i] - toy.point[p][i]));
  EXPECT(worst < 1e-5);
  for (int i = 0; i < 3; ++i) EXPECT(b.cams[0]->estimate().translation()[i] == toy.cam[0].translation()[i]);
  std::printf("toy: chi2 %.6g -> %.3g in %d iterations, worst point error %.2e\n", before, after, done, worst);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// a BAL file end to end (the format ba_demo reads, bal_example.cpp:99-193: counts; camera index,
// point index, u, v per observation; nine numbers per camera -- angle-axis, translation, f, k1, k2 --
// three per point)
// ---------------------------------------------------------------------------------------------
struct BalFile {
  int n_cams = 0, n_points = 0;
  struct Obs { int c, p; double u, v; };
  std::vector<Obs> obs;
  std::vector<double> cam9, xyz;
  bool read(const std::string& path) {
    std::ifstream f(path);
    int n_obs = 0;
    if (!(f >> n_cams >> n_points >> n_obs) || n_cams < 1 || n_points < 1 || n_obs < 1) return false;
    obs.resize(n_obs);
    for (Obs& o : obs) f >> o.c >> o.p >> o.u >> o.v;
    cam9.resize(9 * (size_t)n_cams);
    for (double& x : cam9) f >> x;
    xyz.resize(3 * (size_t)n_points);
    for (double& x : xyz) f >> x;
    if (!f) return false;
    for (const Obs& o : obs)
      if (o.c < 0 || o.c >= n_cams || o.p < 0 || o.p >= n_points) return false;
    return true;
  }
};

// rotation vector -> unit quaternion; the half-angle sinc is expanded near zero
Eigen::Quaterniond rotation_vector_to_quaternion(const double* r) {
  const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2], n = std::sqrt(n2);
  const double half_sinc = n > 1e-8 ? std::sin(0.5 * n) / n : 0.5 - n2 / 48.0;
  return Eigen::Quaterniond(std::cos(0.5 * n), half_sinc * r[0], half_sinc * r[1], half_sinc * r[2]);
}

int run_bal(const std::string& in, const std::string& out, int iterations) {
  BalFile bal;
  if (!bal.read(in)) {
    std::fprintf(stderr, "cannot read %s\n", in.c_str());
    return 1;
  }
  g2o::SparseOptimizer opt;
  opt.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<g2o::BlockSolver_6_3>(
      g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>>())));
  auto* K = new g2o::CameraParameters(kFocal, Eigen::Vector2d(kCx, kCy), 0.0);
  K->setId(0);
  opt.addParameter(K);
  std::vector<g2o::VertexSE3Expmap*> cams(bal.n_cams);
  std::vector<g2o::VertexSBAPointXYZ*> pts(bal.n_points);
  for (int c = 0; c < bal.n_cams; ++c) {
    const double* row = &bal.cam9[9 * (size_t)c];
    cams[c] = new g2o::VertexSE3Expmap();
    cams[c]->setId(c);
    cams[c]->setEstimate(g2o::SE3Quat(rotation_vector_to_quaternion(row), Eigen::Vector3d(row[3], row[4], row[5])));
    opt.addVertex(cams[c]);
  }
  for (int p = 0; p < bal.n_points; ++p) {
    pts[p] = new g2o::VertexSBAPointXYZ();
    pts[p]->setId(bal.n_cams + p);
    pts[p]->setMarginalized(true);
    pts[p]->setEstimate(Eigen::Vector3d(bal.xyz[3 * (size_t)p], bal.xyz[3 * (size_t)p + 1], bal.xyz[3 * (size_t)p + 2]));
    opt.addVertex(pts[p]);
  }
  double worst = 0.0;
  for (const BalFile::Obs& o : bal.obs) {
    auto* e = new g2o::EdgeProjectXYZ2UV();
    e->setVertex(0, pts[o.p]);
    e->setVertex(1, cams[o.c]);
    e->setMeasurement(Eigen::Vector2d(o.u, o.v));
    e->setInformation(Eigen::Matrix2d::Identity());  // pixel noise 1                      (:148)
    auto* rk = new g2o::RobustKernelHuber;
    rk->setDelta(2.5);
    e->setRobustKernel(rk);
    e->setParameterId(0, 0);
    if (!opt.addEdge(e)) {
      std::fprintf(stderr, "edge refused\n");
      return 1;
    }
    e->computeError();
    worst = std::fmax(worst, e->error().norm());
  }
  std::printf("largest reprojection error before: %.12g px\n", worst);
  if (!opt.initializeOptimization()) {
    std::fprintf(stderr, "initializeOptimization failed: %s\n", opt.lastError());
    return 3;
  }
  const double before = opt.activeRobustChi2();
  const int done = opt.optimize(iterations);
  if (done <= 0) {
    std::fprintf(stderr, "optimize failed: %s\n", opt.lastError());
    return 3;
  }
  std::printf("ba: chi2 %.17g -> %.17g in %d iterations\n", before, opt.activeRobustChi2(), done);
  // the demo's result file: camera index, camera centre in the world, rotation camera -> world (x y z w)
  std::ofstream f(out);
  f.precision(17);
  f << "% camera, centre in world, q(camera -> world) xyzw" << std::endl;
  for (int c = 0; c < bal.n_cams; ++c) {
    const g2o::SE3Quat Tc2w = cams[c]->estimate().inverse();
    f << c << " " << Tc2w.translation().transpose() << " " << Tc2w.rotation().coeffs().transpose() << std::endl;
  }
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "";
  int rc = 0;
  if (mode == "host") {
    value_types();
    container_semantics();
  } else if (mode == "toy") {
    rc = solve_toy();
  } else if (mode == "bal" && argc >= 4) {
    rc = run_bal(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 5);
  } else {
    std::fprintf(stderr, "usage: %s host | toy | bal <problem.bal> <poses out> [iterations=5]\n", argv[0]);
    return 2;
  }
  if (rc) return rc;
  std::printf("%d checks, %d failed\n", g_checked, g_failed);
  return g_failed ? 1 : 0;
}
