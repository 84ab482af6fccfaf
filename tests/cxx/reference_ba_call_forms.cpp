// reference_ba_call_forms.cpp -- compile (and, on a GPU box, run) test of include/sim3opt_g2o_ba.hpp.
//
// A bundle-adjustment driver written in the call forms of the reference's ba_demo
// (bal_example.cpp:71-97 optimizer and camera parameters, :112-130 vertices, :134-162 edges,
// :164-193 estimates, :198/:213 optimisation, :223-238 pose file): the same g2o / Eigen expressions on
// the same kinds of objects, so that a maintainer can swap the g2o headers for the shim and keep the
// source.  It is NOT a copy of that file: argument parsing, the structure-only branch and the
// statistics file are left out, and the angle-axis conversion (ceres/rotation.h there) is local.
//
//   g++ -std=c++17 -DSIM3OPT_G2O_BA_NAMES -Iinclude -Itests/mock_eigen tests/cxx/reference_ba_call_forms.cpp
//       -Lsim3opt_amd -lsim3opt -Wl,-rpath,$PWD/sim3opt_amd -o reference_ba_call_forms     (one line)
//   ./reference_ba_call_forms <problem.bal> <poses out> [iterations=5]
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "sim3opt_g2o_ba.hpp"

using namespace std;

// w x y z, as ceres::AngleAxisToQuaternion writes them
static void AngleAxisToQuaternion(const double* aa, double* q) {
  const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > 0.0) {
    const double th = sqrt(th2), k = sin(0.5 * th) / th;
    q[0] = cos(0.5 * th); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
  } else {
    q[0] = 1.0; q[1] = 0.5 * aa[0]; q[2] = 0.5 * aa[1]; q[3] = 0.5 * aa[2];
  }
}

int main(int argc, char** argv) {
  if (argc < 3) {
    cerr << "usage: " << argv[0] << " <problem.bal> <poses out> [iterations=5]" << endl;
    return 2;
  }
  const string inputFilename = argv[1], outputFilename = argv[2];
  const int maxIterations = argc > 3 ? atoi(argv[3]) : 5;
  const double PIXEL_NOISE = 1.0;
  const bool ROBUST_KERNEL = true, DENSE = false, verbose = true;

  g2o::SparseOptimizer optimizer;
  optimizer.setVerbose(verbose);
  std::unique_ptr<g2o::BlockSolver_6_3::LinearSolverType> linearSolver;
  if (DENSE) {
    linearSolver = g2o::make_unique<g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType> >();
  } else {
    linearSolver = g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType> >();
  }
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<g2o::BlockSolver_6_3>(std::move(linearSolver)));
  optimizer.setAlgorithm(solver);

  double focal_length = 718.856;
  Eigen::Vector2d principal_point(607.1928, 185.2157);
  g2o::CameraParameters* cam_params = new g2o::CameraParameters(focal_length, principal_point, 0.);
  cam_params->setId(0);
  if (!optimizer.addParameter(cam_params)) {
    assert(false);
  }

  vector<g2o::VertexSE3Expmap*> cameras;
  vector<g2o::VertexSBAPointXYZ*> points;

  ifstream ifs(inputFilename.c_str());
  int numCameras = 0, numPoints = 0, numObservations = 0;
  ifs >> numCameras >> numPoints >> numObservations;
  if (!ifs || numCameras < 1) {
    cerr << "cannot read " << inputFilename << endl;
    return 1;
  }

  int id = 0;
  cameras.reserve(numCameras);
  for (int i = 0; i < numCameras; ++i, ++id) {
    g2o::VertexSE3Expmap* cam = new g2o::VertexSE3Expmap();
    cam->setId(id);
    optimizer.addVertex(cam);
    cameras.push_back(cam);
  }
  points.reserve(numPoints);
  for (int i = 0; i < numPoints; ++i, ++id) {
    g2o::VertexSBAPointXYZ* p = new g2o::VertexSBAPointXYZ();
    p->setId(id);
    p->setMarginalized(true);
    bool addedVertex = optimizer.addVertex(p);
    if (!addedVertex) {
      cerr << "failing adding vertex" << endl;
    }
    points.push_back(p);
  }

  vector<g2o::EdgeProjectXYZ2UV*> edges;
  for (int i = 0; i < numObservations; ++i) {
    int camIndex, pointIndex;
    double obsX, obsY;
    ifs >> camIndex >> pointIndex >> obsX >> obsY;
    assert(camIndex >= 0 && (size_t)camIndex < cameras.size() && "Index out of bounds");
    g2o::VertexSE3Expmap* cam = cameras[camIndex];
    assert(pointIndex >= 0 && (size_t)pointIndex < points.size() && "Index out of bounds");
    g2o::VertexSBAPointXYZ* point = points[pointIndex];

    g2o::EdgeProjectXYZ2UV* e = new g2o::EdgeProjectXYZ2UV();
    e->setVertex(0, point);
    e->setVertex(1, cam);
    e->setInformation(Eigen::Matrix2d::Identity() / (PIXEL_NOISE * PIXEL_NOISE));
    e->setMeasurement(Eigen::Vector2d(obsX, obsY));
    if (ROBUST_KERNEL) {
      g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
      rk->setDelta(2.5);
      e->setRobustKernel(rk);
    }
    e->setParameterId(0, 0);
    bool addedEdge = optimizer.addEdge(e);
    if (!addedEdge) {
      cerr << "error adding edge" << endl;
    }
    edges.push_back(e);
  }

  Eigen::VectorXd cameraParameter(9);
  for (int i = 0; i < numCameras; ++i) {
    for (int j = 0; j < 9; ++j) ifs >> cameraParameter(j);
    g2o::VertexSE3Expmap* cam = cameras[i];
    double angle_axis[3] = {cameraParameter[0], cameraParameter[1], cameraParameter[2]};
    double qw2cData[4];
    AngleAxisToQuaternion(angle_axis, qw2cData);
    Eigen::Quaterniond qw2c(qw2cData[0], qw2cData[1], qw2cData[2], qw2cData[3]);
    Eigen::Vector3d trans(cameraParameter[3], cameraParameter[4], cameraParameter[5]);
    g2o::SE3Quat pose(qw2c, trans);
    cam->setEstimate(pose);
  }
  Eigen::Vector3d p;
  for (int i = 0; i < numPoints; ++i) {
    ifs >> p(0) >> p(1) >> p(2);
    g2o::VertexSBAPointXYZ* point = points[i];
    point->setEstimate(p);
  }
  if (!ifs) {
    cerr << "truncated " << inputFilename << endl;
    return 1;
  }

  // (the reference takes this maximum while the estimates are still unset, :156-160; here after)
  double maxError = 0;
  for (size_t i = 0; i < edges.size(); ++i) {
    g2o::EdgeProjectXYZ2UV* e = edges[i];
    e->computeError();
    Eigen::Vector2d error = e->error();
    double rootChi2 = error.norm();
    maxError = maxError < rootChi2 ? rootChi2 : maxError;
  }
  cout << setprecision(12) << "max edge error norm " << maxError << endl;

  if (!optimizer.initializeOptimization()) {
    cerr << "initializeOptimization failed: " << optimizer.lastError() << endl;
    return 3;
  }
  const double chi0 = optimizer.activeRobustChi2();
  cout << "Performing full BA:" << endl;
  const int done = optimizer.optimize(maxIterations);
  if (done <= 0) {
    cerr << "optimize failed: " << optimizer.lastError() << endl;
    return 3;
  }
  cout << "ba: chi2 " << chi0 << " -> " << optimizer.activeRobustChi2() << " in " << done << " iterations" << endl;

  ofstream fout(outputFilename.c_str());
  fout << setprecision(17) << "% SE3 optimization result: kf id, tcinw, rc2w(qxyzw):" << endl;
  int jack = 0;
  for (vector<g2o::VertexSE3Expmap*>::const_iterator it = cameras.begin(); it != cameras.end(); ++it) {
    g2o::SE3Quat est = (*it)->estimate();
    Eigen::Quaterniond qw2c = est.rotation();
    Eigen::Vector3d twinc(est.translation());
    Eigen::Vector3d tcinw = -qw2c.conjugate()._transformVector(twinc);
    fout << jack << " " << tcinw.transpose() << " " << qw2c.conjugate().coeffs().transpose() << endl;
    ++jack;
  }
  fout.close();
  return 0;
}
