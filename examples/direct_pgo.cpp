// direct_pgo.cpp -- the reference's testDirectSim3Optimization (kitti_surf.cpp:542-709) written
// against the g2o-named shim (include/sim3opt_g2o.hpp) over the C-ABI of libsim3opt.
//
//   g++ -std=c++17 -DSIM3OPT_G2O_NAMES -Iinclude examples/direct_pgo.cpp
//       -Lsim3opt_amd -lsim3opt -Wl,-rpath,$PWD/sim3opt_amd -o direct_pgo      (one line)
//   ./direct_pgo <dir with cc.txt framePoses*.txt loopConstraints.txt> <out.txt> [one_loop=1] [iters=100]
//
// The loaders of kitti_surf.cpp:145-292 live in the library (sim3opt_load_kitti_direct); here
// they only supply the arrays, the graph itself is built through the g2o-style calls.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sim3opt_g2o.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <data dir> <out file> [one_loop=1] [iters=100]\n", argv[0]);
    return 2;
  }
  const int one = argc > 3 ? std::atoi(argv[3]) : 1;
  const int iters = argc > 4 ? std::atoi(argv[4]) : 100;

  // keyframes + constraints (GetAllKeyFrames / LoadLoopConstraints, kitti_surf.cpp:562-573)
  sim3opt_graph* src = sim3opt_create();
  if (sim3opt_load_kitti_direct(src, argv[1], one) != SIM3OPT_OK) {
    std::fprintf(stderr, "cannot load %s\n", argv[1]);
    return 1;
  }
  const int nv = sim3opt_num_vertices(src), ne = sim3opt_num_edges(src);
  std::vector<double> st(8 * (size_t)nv);
  sim3opt_get_vertices(src, st.data());

  // Setup optimizer (kitti_surf.cpp:552-558)
  g2o::SparseOptimizer optimizer;
  std::unique_ptr<g2o::BlockSolverX::LinearSolverType> linearSolver =
      g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolverX::PoseMatrixType> >();
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(
      g2o::make_unique<g2o::BlockSolverX>(std::move(linearSolver)));
  optimizer.setAlgorithm(solver);

  // SET KEYFRAME VERTICES (kitti_surf.cpp:597-622)
  for (int i = 0; i < nv; ++i) {
    vio::VertexSim3Expmap* vSim3 = new vio::VertexSim3Expmap();
    vSim3->setEstimate(g2o::Sim3(&st[8 * (size_t)i]));
    if (i == 0) vSim3->setFixed(true);
    vSim3->setId(i);
    vSim3->setMarginalized(false);
    optimizer.addVertex(vSim3);
  }
  // SET LOOP + NORMAL EDGES (kitti_surf.cpp:624-670)
  for (int k = 0; k < ne; ++k) {
    int32_t a, b;
    double m[8];
    sim3opt_get_edge(src, k, &a, &b, m);
    vio::EdgeSim3* e = new vio::EdgeSim3();
    e->setVertex(1, optimizer.vertex(b));
    e->setVertex(0, optimizer.vertex(a));
    e->setMeasurement(g2o::Sim3(m));
    for (int d = 0; d < 7; ++d) e->information()[8 * d] = 1.0;  // matLambdasim = Identity (:592)
    optimizer.addEdge(e);
  }
  sim3opt_destroy(src);

  if (!optimizer.initializeOptimization()) {  // kitti_surf.cpp:674
    std::fprintf(stderr, "initializeOptimization: %s\n", optimizer.lastError());
    return 1;
  }
  const double chi0 = optimizer.activeChi2();
  const int done = optimizer.optimize(iters);  // kitti_surf.cpp:675
  std::printf("chi2 %.10g -> %.10g in %d iterations\n", chi0, optimizer.activeChi2(), done);

  // result file (kitti_surf.cpp:678-701)
  if (sim3opt_write_poses(optimizer.handle(), argv[2], nullptr) != SIM3OPT_OK) return 1;
  g2o::Sim3 S5 = static_cast<vio::VertexSim3Expmap*>(optimizer.vertex(5))->estimate();
  g2o::Sim3 Swc = S5.inverse();
  std::printf("kf 5: s %.6g  t(Swc) %.6g %.6g %.6g\n", S5.scale(), Swc.translation_ptr()[0],
              Swc.translation_ptr()[1], Swc.translation_ptr()[2]);
  return done > 0 ? 0 : 1;
}
