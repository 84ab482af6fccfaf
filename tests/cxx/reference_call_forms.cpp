// reference_call_forms.cpp -- compile (and, on a GPU box, run) test of include/sim3opt_g2o.hpp.
//
// The two graph builders below are written in the call forms of the reference's
// testDirectSim3Optimization (kitti_surf.cpp:552-558, :592-701) and testStepwiseSim3Optimization
// (:726-735, :774-886, :1020-1075): the same g2o / vio / Eigen / Sophus expressions on the same kinds
// of objects, so that a maintainer can swap the g2o headers for the shim and keep the source.  They
// are NOT a copy of that file: keyframes and loop constraints come from the library's loader, the
// scale SVD (an Eigen::JacobiSVD in the caller, :887-934) is the library's stepwise_scale_init, and
// profiling / debug streams are left out.
//
//   g++ -std=c++17 -DSIM3OPT_G2O_NAMES -Iinclude -Itests/mock_eigen tests/cxx/reference_call_forms.cpp
//       -Lsim3opt_amd -lsim3opt -Wl,-rpath,$PWD/sim3opt_amd -o reference_call_forms      (one line)
//   (with a real Eigen + Sophus: -I/usr/include/eigen3 instead of -Itests/mock_eigen)
//   ./reference_call_forms <dir with cc.txt framePoses*.txt loopConstraints.txt> <out prefix> [one_loop=1]
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <sophus/se3.hpp>

#include "sim3opt_g2o.hpp"

using namespace std;

// what the reference's KeyFrame / Constraint give the builders (kittiDetector.h:404-435)
struct KeyFrame {
  int mnId = 0, mnFrameId = 0;
  Sophus::SE3d Tw2c;
  bool isBad() const { return false; }
  Eigen::Matrix<double, 3, 3> GetRotation() const { return Tw2c.rotationMatrix(); }
  Eigen::Matrix<double, 3, 1> GetTranslation() const { return Tw2c.translation(); }
  void SetPose(const Sophus::SE3d& T) { Tw2c = T; }
};
template <class T, int N>
struct Constraint {
  int trans_id1 = 0, trans_id2 = 0;
  T mean;
};

static void LoadFromLibrary(const string& dir, bool one, vector<KeyFrame*>& vpKFs,
                            vector<Constraint<g2o::Sim3, 7>, Eigen::aligned_allocator<Constraint<g2o::Sim3, 7> > >& loops) {
  sim3opt_graph* src = sim3opt_create();
  if (sim3opt_load_kitti_direct(src, dir.c_str(), one ? 1 : 0) != SIM3OPT_OK) {
    cerr << "cannot load " << dir << ": " << sim3opt_last_error(src) << endl;
    exit(1);
  }
  const int nv = sim3opt_num_vertices(src), ne = sim3opt_num_edges(src);
  for (int i = 0; i < nv; ++i) {
    double s[8];
    sim3opt_get_vertex(src, i, s);
    KeyFrame* kf = new KeyFrame();
    kf->mnId = i;  // (image ids are not needed here)
    kf->mnFrameId = i;
    Eigen::Vector3d t;
    t[0] = s[4]; t[1] = s[5]; t[2] = s[6];
    kf->Tw2c = Sophus::SE3d(Eigen::Quaterniond(s[3], s[0], s[1], s[2]), t);
    vpKFs.push_back(kf);
  }
  for (int k = 0; k < ne - (nv - 1); ++k) {  // loop edges come first, then the nv - 1 odometry edges
    int32_t a, b;
    double m[8];
    sim3opt_get_edge(src, k, &a, &b, m);
    Constraint<g2o::Sim3, 7> c;
    c.trans_id1 = a;
    c.trans_id2 = b;
    c.mean = g2o::Sim3(m);
    loops.push_back(c);
  }
  sim3opt_destroy(src);
}

Eigen::Vector4d toScaleTrans(const g2o::Sim3& se3q) {
  Eigen::Vector4d v4;
  v4[0] = se3q.scale();
  v4.tail<3>() = se3q.translation();
  return v4;
}

// ---- direct: all of Sim(3) at once ----
static double DirectCallForms(const string& dir, const string& directFile, bool bUseOneContraint) {
  ofstream logStream(directFile);

  g2o::SparseOptimizer optimizer;
  std::unique_ptr<g2o::BlockSolverX::LinearSolverType> linearSolver =
      g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType> >();
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<g2o::BlockSolverX>(std::move(linearSolver)));
  optimizer.setAlgorithm(solver);

  vector<KeyFrame*> vpKFs;
  std::vector<Constraint<g2o::Sim3, 7>, Eigen::aligned_allocator<Constraint<g2o::Sim3, 7> > > loopConnections;
  LoadFromLibrary(dir, bUseOneContraint, vpKFs, loopConnections);
  assert(vpKFs.front()->mnFrameId == 0);
  unsigned int nMaxKFid = vpKFs.back()->mnFrameId;

  Eigen::Matrix<double, 7, 7> matLambdasim = Eigen::Matrix<double, 7, 7>::Identity();
  vector<g2o::Sim3, Eigen::aligned_allocator<g2o::Sim3> > vScw(nMaxKFid + 1);
  vector<g2o::Sim3, Eigen::aligned_allocator<g2o::Sim3> > vCorrectedSwc(nMaxKFid + 1);

  for (size_t i = 0, iend = vpKFs.size(); i < iend; ++i) {  // keyframe vertices
    KeyFrame* pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    vio::VertexSim3Expmap* vSim3 = new vio::VertexSim3Expmap();
    int nIDi = pKF->mnFrameId;
    Eigen::Matrix<double, 3, 3> Rcw = pKF->GetRotation();
    Eigen::Matrix<double, 3, 1> tcw = pKF->GetTranslation();
    g2o::Sim3 Siw(Rcw, tcw, 1.0);
    vScw[nIDi] = Siw;
    vSim3->setEstimate(Siw);
    if (nIDi == 0) vSim3->setFixed(true);
    vSim3->setId(nIDi);
    vSim3->setMarginalized(false);
    optimizer.addVertex(vSim3);
  }
  for (auto mit = loopConnections.begin(), mend = loopConnections.end(); mit != mend; mit++) {  // loop edges
    const long unsigned int nIDi = vpKFs[mit->trans_id1]->mnFrameId;
    const long unsigned int nIDj = vpKFs[mit->trans_id2]->mnFrameId;
    vio::EdgeSim3* esim = new vio::EdgeSim3();
    esim->setVertex(1, optimizer.vertex(nIDj));
    esim->setVertex(0, optimizer.vertex(nIDi));
    esim->setMeasurement(mit->mean);
    esim->information() = matLambdasim;
    optimizer.addEdge(esim);
  }
  for (size_t i = 1, iend = vpKFs.size(); i < iend; i++) {  // spanning-tree edges
    int nIDi = vpKFs[i]->mnFrameId;
    g2o::Sim3 Swi = vScw[nIDi].inverse();
    int nIDj = vpKFs[i - 1]->mnFrameId;
    g2o::Sim3 Sjw = vScw[nIDj];
    g2o::Sim3 Sji = Sjw * Swi;
    vio::EdgeSim3* e = new vio::EdgeSim3();
    e->setVertex(1, optimizer.vertex(nIDj));
    e->setVertex(0, optimizer.vertex(nIDi));
    e->setMeasurement(Sji);
    e->information() = matLambdasim;
    optimizer.addEdge(e);
  }

  optimizer.initializeOptimization();
  const double chi0 = optimizer.activeChi2();
  const int iters = optimizer.optimize(100);

  logStream << "% sim3 optimization result: kf id, sw2i, scaled tiinw, ri2w(qxyzw):" << endl;
  for (size_t i = 0; i < vpKFs.size(); i++) {
    KeyFrame* pKFi = vpKFs[i];
    const int nIDi = pKFi->mnFrameId;
    g2o::Sim3 CorrectedSiw;
    vio::VertexSim3Expmap* vSim3 = static_cast<vio::VertexSim3Expmap*>(optimizer.vertex(nIDi));
    CorrectedSiw = vSim3->estimate();
    vCorrectedSwc[nIDi] = CorrectedSiw.inverse();
    Sophus::SE3d Tiw(CorrectedSiw.rotation(), CorrectedSiw.translation() / CorrectedSiw.scale());
    logStream << pKFi->mnId << " " << CorrectedSiw.scale() << " "
              << vCorrectedSwc[nIDi].translation().transpose() << " "
              << vCorrectedSwc[nIDi].rotation().coeffs().transpose() << endl;
    pKFi->SetPose(Tiw);
  }
  logStream.close();
  const double chi1 = optimizer.activeChi2();
  printf("direct: chi2 %.10g -> %.10g in %d iterations\n", chi0, chi1, iters);
  for (KeyFrame* kf : vpKFs) delete kf;
  return chi1;
}

// ---- stepwise: scales, then scale + translation, then (num_optimizer == 3) all of Sim(3) ----
static double StepwiseCallForms(const string& dir, const string& outputFile, bool bUseOneContraint,
                                int num_optimizer) {
  ofstream logStream(outputFile);

  g2o::SparseOptimizer* optimizer = new g2o::SparseOptimizer[num_optimizer];
  for (int jack = 0; jack < num_optimizer; ++jack) {
    std::unique_ptr<g2o::BlockSolverX::LinearSolverType> linearSolver =
        g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType> >();
    g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(
        g2o::make_unique<g2o::BlockSolverX>(std::move(linearSolver)));
    optimizer[jack].setAlgorithm(solver);
  }
  vector<KeyFrame*> vpKFs;
  std::vector<Constraint<g2o::Sim3, 7>, Eigen::aligned_allocator<Constraint<g2o::Sim3, 7> > > loopConnections;
  LoadFromLibrary(dir, bUseOneContraint, vpKFs, loopConnections);
  unsigned int nMaxKFid = vpKFs.back()->mnFrameId;

  Eigen::Matrix<double, 1, 1> matLambdas = Eigen::Matrix<double, 1, 1>::Identity();
  Eigen::Matrix<double, 4, 4> matLambdast = Eigen::Matrix<double, 4, 4>::Identity();
  Eigen::Matrix<double, 7, 7> matLambdasim = Eigen::Matrix<double, 7, 7>::Identity();
  vector<g2o::Sim3, Eigen::aligned_allocator<g2o::Sim3> > vScw(nMaxKFid + 1);
  vector<g2o::Sim3, Eigen::aligned_allocator<g2o::Sim3> > vCorrectedSwc(nMaxKFid + 1);

  for (size_t i = 0, iend = vpKFs.size(); i < iend; ++i) {
    KeyFrame* pKF = vpKFs[i];
    vio::G2oVertexScale* vS = new vio::G2oVertexScale();
    vio::G2oVertexScaleTrans* vST = new vio::G2oVertexScaleTrans();
    vio::VertexSim3Expmap* vSim3 = NULL;
    if (num_optimizer == 3) vSim3 = new vio::VertexSim3Expmap();
    int nIDi = pKF->mnFrameId;
    Eigen::Matrix<double, 3, 3> Rcw = pKF->GetRotation();
    Eigen::Matrix<double, 3, 1> tcw = pKF->GetTranslation();
    g2o::Sim3 Siw(Rcw, tcw, 1.0);
    vScw[nIDi] = Siw;
    vS->setEstimate(Siw.scale());
    vST->setEstimate(toScaleTrans(Siw));
    vST->Rw2i = Sophus::SO3d(Rcw);
    if (num_optimizer == 3) vSim3->setEstimate(Siw);
    if (nIDi == 0) {
      vS->setFixed(true);
      vST->setFixed(true);
      if (num_optimizer == 3) vSim3->setFixed(true);
    }
    vS->setId(nIDi);
    vS->setMarginalized(false);
    vST->setId(nIDi);
    vST->setMarginalized(false);
    optimizer[0].addVertex(vS);
    optimizer[1].addVertex(vST);
    if (num_optimizer == 3) {
      vSim3->setId(nIDi);
      vSim3->setMarginalized(false);
      optimizer[2].addVertex(vSim3);
    }
  }
  auto add_edges = [&](long unsigned int nIDi, long unsigned int nIDj, const g2o::Sim3& mean) {
    vio::G2oEdgeScale* es = new vio::G2oEdgeScale();
    es->setVertex(1, optimizer[0].vertex(nIDj));
    es->setVertex(0, optimizer[0].vertex(nIDi));
    es->setMeasurement(mean.scale());
    es->information() = matLambdas;
    optimizer[0].addEdge(es);

    vio::G2oEdgeScaleTrans* est = new vio::G2oEdgeScaleTrans();
    est->setVertex(1, optimizer[1].vertex(nIDj));
    est->setVertex(0, optimizer[1].vertex(nIDi));
    est->setMeasurement(toScaleTrans(mean));
    est->information() = matLambdast;
    optimizer[1].addEdge(est);

    if (num_optimizer == 3) {
      vio::EdgeSim3* esim = new vio::EdgeSim3();
      esim->setVertex(1, optimizer[2].vertex(nIDj));
      esim->setVertex(0, optimizer[2].vertex(nIDi));
      esim->setMeasurement(mean);
      esim->information() = matLambdasim;
      optimizer[2].addEdge(esim);
    }
  };
  for (auto mit = loopConnections.begin(), mend = loopConnections.end(); mit != mend; mit++)
    add_edges(vpKFs[mit->trans_id1]->mnFrameId, vpKFs[mit->trans_id2]->mnFrameId, mit->mean);
  for (size_t i = 1, iend = vpKFs.size(); i < iend; i++) {
    int nIDi = vpKFs[i]->mnFrameId, nIDj = vpKFs[i - 1]->mnFrameId;
    g2o::Sim3 Swi = vScw[nIDi].inverse();
    g2o::Sim3 Sjw = vScw[nIDj];
    g2o::Sim3 Sji = Sjw * Swi;
    add_edges(nIDi, nIDj, Sji);
  }

  // "scale_dlt": the null vector of the scale equations (the reference: Eigen::JacobiSVD in the caller)
  double ratio = 0;
  if (sim3opt_stepwise_scale_init(optimizer[0].handle(), &ratio) != SIM3OPT_OK) {
    cerr << "scale init: " << optimizer[0].lastError() << endl;
    exit(1);
  }
  for (size_t i = 0; i < vpKFs.size(); ++i) {  // update scale estimates
    const int nIDi = vpKFs[i]->mnFrameId;
    vio::G2oVertexScale* vS = static_cast<vio::G2oVertexScale*>(optimizer[0].vertex(nIDi));
    vio::G2oVertexScaleTrans* vST = static_cast<vio::G2oVertexScaleTrans*>(optimizer[1].vertex(nIDi));
    Eigen::Vector4d stw2i = vST->estimate();
    stw2i[0] = vS->estimate();
    vST->setEstimate(stw2i);
  }
  // "scale_trans"
  optimizer[1].initializeOptimization();
  const double chi_st0 = optimizer[1].activeChi2();
  optimizer[1].optimize(100);
  const double chi_st1 = optimizer[1].activeChi2();
  double chi_final = chi_st1;

  if (num_optimizer == 3) {
    for (size_t i = 0; i < vpKFs.size(); ++i) {  // warm start of the Sim(3) stage
      const int nIDi = vpKFs[i]->mnFrameId;
      vio::G2oVertexScaleTrans* vST = static_cast<vio::G2oVertexScaleTrans*>(optimizer[1].vertex(nIDi));
      Eigen::Vector4d stw2i = vST->estimate();
      g2o::Sim3 CorrectedSiw(vST->Rw2i.unit_quaternion(), stw2i.tail<3>(), stw2i[0]);
      vio::VertexSim3Expmap* vSim3 = static_cast<vio::VertexSim3Expmap*>(optimizer[2].vertex(nIDi));
      vSim3->setEstimate(CorrectedSiw);
    }
    optimizer[2].initializeOptimization();
    optimizer[2].optimize(100);
    chi_final = optimizer[2].activeChi2();
  }

  logStream << "% sim3 optimization result: kf frameid, sw2i, scaled tiinw, ri2w(qxyzw):" << endl;
  for (size_t i = 0; i < vpKFs.size(); i++) {
    KeyFrame* pKFi = vpKFs[i];
    const int nIDi = pKFi->mnFrameId;
    g2o::Sim3 CorrectedSiw;
    if (num_optimizer == 3) {
      vio::VertexSim3Expmap* vSim3 = static_cast<vio::VertexSim3Expmap*>(optimizer[2].vertex(nIDi));
      CorrectedSiw = vSim3->estimate();
    } else {
      vio::G2oVertexScaleTrans* vST = static_cast<vio::G2oVertexScaleTrans*>(optimizer[1].vertex(nIDi));
      Eigen::Vector4d stw2i = vST->estimate();
      CorrectedSiw = g2o::Sim3(vST->Rw2i.unit_quaternion(), stw2i.tail<3>(), stw2i[0]);
    }
    vCorrectedSwc[nIDi] = CorrectedSiw.inverse();
    Sophus::SE3d Tiw(CorrectedSiw.rotation(), CorrectedSiw.translation() / CorrectedSiw.scale());
    logStream << pKFi->mnId << " " << CorrectedSiw.scale() << " "
              << vCorrectedSwc[nIDi].translation().transpose() << " "
              << vCorrectedSwc[nIDi].rotation().coeffs().transpose() << endl;
    pKFi->SetPose(Tiw);
  }
  logStream.close();
  printf("stepwise (%d optimizers): scale sigma ratio %.3g, scale-trans chi2 %.10g -> %.10g, final chi2 %.10g\n",
         num_optimizer, ratio, chi_st0, chi_st1, chi_final);
  delete[] optimizer;
  for (KeyFrame* kf : vpKFs) delete kf;
  return chi_final;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <data dir> <out prefix> [one_loop=1]\n", argv[0]);
    return 2;
  }
  const string dir = argv[1], prefix = argv[2];
  const bool one = argc > 3 ? atoi(argv[3]) != 0 : true;
  const double a = DirectCallForms(dir, prefix + "direct_pure.txt", one);
  const double b = StepwiseCallForms(dir, prefix + "stepwise_2solvers.txt", one, 2);
  const double c = StepwiseCallForms(dir, prefix + "stepwise_3solvers.txt", one, 3);
  return (a > 0 && b >= 0 && c > 0) ? 0 : 1;
}
