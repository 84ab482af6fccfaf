// kitti_io.cpp -- reference-format loaders and the graph builder of the direct Sim3 PGO.
//
// Host C++ for the callers either side of the hot path:
//   LoadKFIndices / LoadKFPoses ......... kitti_surf.cpp:232-292
//   LoadLoopConstraints (Sim3) .......... kitti_surf.cpp:145-205
//   roteu2ro ............................ kittiDetector.h:225-243
//   graph of testDirectSim3Optimization . kitti_surf.cpp:575-670
//   pose writer ......................... kitti_surf.cpp:678-701
// Built on the public C-ABI only (sim3opt_add_vertex / sim3opt_add_edge).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/sim3opt.h"
#include "sim3_math.hpp"

namespace {

using sim3::Sim3;

void euler_rpy_to_R(double r, double p, double y, double R[9]) {  // Rz(yaw) Ry(pitch) Rx(roll)
  const double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p);
  const double ch = std::cos(y), sh = std::sin(y);
  R[0] = cp * ch; R[1] = sp * sr * ch - cr * sh; R[2] = cr * sp * ch + sh * sr;
  R[3] = cp * sh; R[4] = sr * sp * sh + cr * ch; R[5] = cr * sp * sh - sr * ch;
  R[6] = -sp;     R[7] = sr * cp;                R[8] = cr * cp;
}

bool read_lines(const std::string& path, std::vector<std::string>& out) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) return false;
  std::string ln;
  while (std::getline(f, ln)) {
    if (!ln.empty() && ln.back() == '\r') ln.pop_back();
    out.push_back(ln);
  }
  return true;
}

bool blank(const std::string& s) { return s.find_first_not_of(" \t") == std::string::npos; }

void to_array(const Sim3& s, double a[8]) {
  a[0] = s.q[0]; a[1] = s.q[1]; a[2] = s.q[2]; a[3] = s.q[3];
  a[4] = s.t[0]; a[5] = s.t[1]; a[6] = s.t[2]; a[7] = s.s;
}

}  // namespace

extern "C" int sim3opt_load_kitti_direct(sim3opt_graph* g, const char* dir, int32_t use_one) {
  if (!g || !dir) return SIM3OPT_ERR_ARG;
  const std::string d(dir);
  // cc.txt: keyframe id = line number, image id = value (kitti_surf.cpp:232-254)
  std::vector<int> cc;
  {
    std::ifstream f((d + "/cc.txt").c_str());
    if (!f.is_open()) return SIM3OPT_ERR_IO;
    int v;
    while (f >> v) cc.push_back(v);
  }
  if (cc.empty()) return SIM3OPT_ERR_IO;
  // framePoses: 2 header lines, "id, time, r, p, y, x, y, z" of T_c2w; keep ids in cc (:255-292)
  std::vector<std::string> lines;
  if (!read_lines(d + "/framePoses.txt", lines) && !read_lines(d + "/framePoses_kf.txt", lines))
    return SIM3OPT_ERR_IO;
  std::vector<Sim3> Siw(cc.size());
  size_t it = 0;
  for (size_t ln = 2; ln < lines.size() && it < cc.size(); ++ln) {
    if (blank(lines[ln])) continue;
    int id;
    double tm, v[6];
    if (std::sscanf(lines[ln].c_str(), "%d, %lf, %lf, %lf, %lf, %lf, %lf, %lf", &id, &tm, &v[0],
                    &v[1], &v[2], &v[3], &v[4], &v[5]) != 8)
      return SIM3OPT_ERR_IO;
    if (id != cc[it]) continue;
    double R[9], qc2w[4], Rw2c[9];
    euler_rpy_to_R(v[0], v[1], v[2], R);
    // Sophus::SE3d(Rc2w, t).inverse(): conjugate quaternion, t' = R^-1 (-t)      (:283-285)
    sim3::quat_from_R(R, qc2w);
    Sim3 s;
    const double qw2c[4] = {-qc2w[0], -qc2w[1], -qc2w[2], qc2w[3]};
    const double nt[3] = {-v[3], -v[4], -v[5]};
    sim3::quat_rot(qw2c, nt, s.t);
    // g2o::Sim3(Rcw, tcw, 1.0) with Rcw = Tw2c.rotationMatrix()                  (:606-608)
    sim3::R_from_quat(qw2c, Rw2c);
    sim3::quat_from_R(Rw2c, s.q);
    s.s = 1.0;
    Siw[it++] = s;
  }
  if (it != cc.size()) return SIM3OPT_ERR_IO;  // reference: assert(it == vpKFs.end())
  // loopConstraints: 5 header lines, 4 lines per loop; ids from line 1, values from line 4 (:145-205)
  lines.clear();
  if (!read_lines(d + "/loopConstraints.txt", lines)) return SIM3OPT_ERR_IO;
  struct Loop { int f1, f2; Sim3 C; };
  std::vector<Loop> loops;
  {
    std::vector<std::string> rec;
    for (size_t ln = 5; ln < lines.size(); ++ln)
      if (!blank(lines[ln])) rec.push_back(lines[ln]);
    for (size_t k = 0; k + 3 < rec.size(); k += 4) {
      unsigned f1, f2;
      double gt[6], v[6], sf2s;
      int matches;
      if (std::sscanf(rec[k].c_str(), "%u %u %lf %lf %lf %lf %lf %lf", &f1, &f2, &gt[0], &gt[1],
                      &gt[2], &gt[3], &gt[4], &gt[5]) != 8)
        return SIM3OPT_ERR_IO;
      if (std::sscanf(rec[k + 3].c_str(), "%d %lf %lf %lf %lf %lf %lf %lf", &matches, &sf2s, &v[0],
                      &v[1], &v[2], &v[3], &v[4], &v[5]) != 8)
        return SIM3OPT_ERR_IO;
      if (v[0] == 0 || v[1] == 0 || v[2] == 0) return SIM3OPT_ERR_IO;  // reference asserts (:190)
      double R[9];
      euler_rpy_to_R(v[0], v[1], v[2], R);
      Loop L;
      L.f1 = (int)f1; L.f2 = (int)f2;
      sim3::quat_from_R(R, L.C.q);
      L.C.t[0] = v[3]; L.C.t[1] = v[4]; L.C.t[2] = v[5];
      L.C.s = sf2s;
      loops.push_back(L);
    }
  }
  if (loops.empty()) return SIM3OPT_ERR_IO;
  if (use_one) loops.resize(1);  // bUseOneContraint (:568-573)
  std::map<int, int> frame2kf;   // :575-590
  for (size_t k = 0; k < cc.size(); ++k) frame2kf[cc[k]] = (int)k;
  double a[8];
  for (size_t k = 0; k < cc.size(); ++k) {  // vertices: id = keyframe id, vertex 0 fixed (:597-622)
    to_array(Siw[k], a);
    const int rc = sim3opt_add_vertex(g, (int32_t)k, a, k == 0);
    if (rc != SIM3OPT_OK) return rc;
  }
  for (const Loop& L : loops) {  // loop edges: setVertex(0, id1), setVertex(1, id2) (:624-640)
    auto i1 = frame2kf.find(L.f1), i2 = frame2kf.find(L.f2);
    if (i1 == frame2kf.end() || i2 == frame2kf.end()) return SIM3OPT_ERR_IO;
    to_array(L.C, a);
    const int rc = sim3opt_add_edge(g, i1->second, i2->second, a, nullptr, SIM3OPT_KERNEL_NONE, 0.0);
    if (rc != SIM3OPT_OK) return rc;
  }
  for (size_t i = 1; i < cc.size(); ++i) {  // odometry: Sji = Sjw * Swi, v0 = i, v1 = i-1 (:649-670)
    const Sim3 Sji = sim3::mul(Siw[i - 1], sim3::inverse(Siw[i]));
    to_array(Sji, a);
    const int rc = sim3opt_add_edge(g, (int32_t)i, (int32_t)(i - 1), a, nullptr, SIM3OPT_KERNEL_NONE, 0.0);
    if (rc != SIM3OPT_OK) return rc;
  }
  return SIM3OPT_OK;
}

// Consistency graph of the loader's conventions against numbers the reference itself wrote:
// line 1 of every loopConstraints.txt record is DCM2Euler(Pw2c[f2] Pw2c[f1]^-1) and its translation,
// computed by the reference's detector from the KITTI ground truth (kittiDetector.h:1051-1060, 8
// decimals; ReadCameraPose :599-624 inverts the 3x4 Pc2w row).  With the ground-truth poses as
// vertices S_iw = (Rw2c, tw2c, 1) and those records as edges (v0 = frame 1, v1 = frame 2), every
// residual log(C S_v0 S_v1^-1) vanishes to the file's precision -- if the Euler convention
// (roteu2ro, kittiDetector.h:225-243), compose, inverse and the edge orientation are the reference's.
static int load_kitti_gt_loops_body(sim3opt_graph* g, const char* dir);
extern "C" int sim3opt_load_kitti_gt_loops(sim3opt_graph* g, const char* dir) {
  if (!g || !dir) return SIM3OPT_ERR_ARG;
  try {  // (std::map / std::string / istringstream allocate: nothing may cross the C boundary)
    return load_kitti_gt_loops_body(g, dir);
  } catch (...) {
    return SIM3OPT_ERR_IO;
  }
}
static int load_kitti_gt_loops_body(sim3opt_graph* g, const char* dir) {
  const std::string d(dir);
  // ground truth: "image_id r00 r01 r02 tx r10 ... tz" rows of the keyframes (gt_kf.txt, one comment
  // line), or the full 00.txt (row number = image id, kitti_surf.cpp:1164-1190)
  std::map<int, int> frame2v;
  std::vector<std::string> lines;
  const bool kf_file = read_lines(d + "/gt_kf.txt", lines);
  if (!kf_file && !read_lines(d + "/00.txt", lines)) return SIM3OPT_ERR_IO;
  int row = 0;
  for (const std::string& ln : lines) {
    if (blank(ln) || ln[0] == '%') continue;
    std::istringstream is(ln);
    int id = row;
    if (kf_file) is >> id;
    double P[12];
    for (double& x : P) is >> x;
    if (!is) return SIM3OPT_ERR_IO;
    // Pw2c = Pc2w^-1 as a MATRIX inverse (ReadCameraPose's cv::Mat::inv(), kittiDetector.h:622): the
    // file's 7-digit rotations are orthonormal to 1e-7 only
    const double* M = P;  // rows (M[0..2], M[4..6], M[8..10]), translation M[3], M[7], M[11]
    const double c00 = M[5] * M[10] - M[6] * M[9], c01 = M[6] * M[8] - M[4] * M[10], c02 = M[4] * M[9] - M[5] * M[8];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    if (!(std::fabs(det) > 1e-12)) return SIM3OPT_ERR_IO;
    const double Rw2c[9] = {c00 / det, (M[2] * M[9] - M[1] * M[10]) / det, (M[1] * M[6] - M[2] * M[5]) / det,
                            c01 / det, (M[0] * M[10] - M[2] * M[8]) / det, (M[2] * M[4] - M[0] * M[6]) / det,
                            c02 / det, (M[1] * M[8] - M[0] * M[9]) / det, (M[0] * M[5] - M[1] * M[4]) / det};
    Sim3 s;
    sim3::quat_from_R(Rw2c, s.q);
    for (int i = 0; i < 3; ++i) s.t[i] = -(Rw2c[3 * i] * P[3] + Rw2c[3 * i + 1] * P[7] + Rw2c[3 * i + 2] * P[11]);
    s.s = 1.0;
    double a[8];
    to_array(s, a);
    if (frame2v.count(id)) return SIM3OPT_ERR_IO;  // a repeated image id would orphan its first vertex
    const int vnew = (int)frame2v.size();
    frame2v[id] = vnew;
    const int rc = sim3opt_add_vertex(g, (int32_t)vnew, a, row == 0);
    if (rc != SIM3OPT_OK) return rc;
    ++row;
  }
  lines.clear();
  if (!read_lines(d + "/loopConstraints.txt", lines)) return SIM3OPT_ERR_IO;
  std::vector<std::string> rec;
  for (size_t ln = 5; ln < lines.size(); ++ln)
    if (!blank(lines[ln])) rec.push_back(lines[ln]);
  for (size_t k = 0; k + 3 < rec.size(); k += 4) {
    unsigned f1, f2;
    double gt[6];
    if (std::sscanf(rec[k].c_str(), "%u %u %lf %lf %lf %lf %lf %lf", &f1, &f2, &gt[0], &gt[1], &gt[2],
                    &gt[3], &gt[4], &gt[5]) != 8)
      return SIM3OPT_ERR_IO;
    auto i1 = frame2v.find((int)f1), i2 = frame2v.find((int)f2);
    if (i1 == frame2v.end() || i2 == frame2v.end()) return SIM3OPT_ERR_IO;
    double R[9], a[8];
    euler_rpy_to_R(gt[0], gt[1], gt[2], R);
    Sim3 C;
    sim3::quat_from_R(R, C.q);
    C.t[0] = gt[3]; C.t[1] = gt[4]; C.t[2] = gt[5];
    C.s = 1.0;
    to_array(C, a);
    const int rc = sim3opt_add_edge(g, i1->second, i2->second, a, nullptr, SIM3OPT_KERNEL_NONE, 0.0);
    if (rc != SIM3OPT_OK) return rc;
  }
  return sim3opt_num_edges(g) > 0 ? SIM3OPT_OK : SIM3OPT_ERR_IO;
}

extern "C" int sim3opt_write_poses(sim3opt_graph* g, const char* path, const int32_t* image_ids) {
  if (!g || !path) return SIM3OPT_ERR_ARG;
  const int32_t nv = sim3opt_num_vertices(g);
  std::vector<double> st(8 * (size_t)(nv > 0 ? nv : 1));
  const int rc = sim3opt_get_vertices(g, st.data());
  if (rc != SIM3OPT_OK) return rc;
  FILE* f = std::fopen(path, "w");
  if (!f) return SIM3OPT_ERR_IO;
  std::fprintf(f, "%% sim3 optimization result: kf id, sw2i, scaled tiinw, ri2w(qxyzw):\n");
  for (int32_t k = 0; k < nv; ++k) {
    Sim3 S;
    const double* a = &st[8 * (size_t)k];
    S.q[0] = a[0]; S.q[1] = a[1]; S.q[2] = a[2]; S.q[3] = a[3];
    S.t[0] = a[4]; S.t[1] = a[5]; S.t[2] = a[6]; S.s = a[7];
    const Sim3 Swi = sim3::inverse(S);  // vCorrectedSwc (:691)
    std::fprintf(f, "%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n",
                 image_ids ? image_ids[k] : k, S.s, Swi.t[0], Swi.t[1], Swi.t[2], Swi.q[0],
                 Swi.q[1], Swi.q[2], Swi.q[3]);
  }
  std::fclose(f);
  return SIM3OPT_OK;
}
