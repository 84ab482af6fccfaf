// map_io.hip -- interchange formats and map-point re-anchoring.
//
//   KeyFrame .bin reader ........ LoadComboKeyFrame, drawPTAMPoints.cpp:33-84 / kittiDetector.h:299-362
//   map-point re-anchoring ...... figureKITTIBA, drawPTAMPoints.cpp:416-429: every observation
//       (frame, point) re-expresses the point through its keyframe's OLD pose and the keyframe's
//       CORRECTED Sim3; later observations overwrite earlier ones ("many previous projection may be
//       crushed"), so the result is the transform through the point's LAST observation.
//   .g2o export ................. g2o's VERTEX_SIM3:EXPMAP / EDGE_SIM3:EXPMAP text form, so a graph
//       built here can be re-run in stock g2o (format from upstream g2o, not in the reference tree)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sim3opt.h"
#include "sim3_math.hpp"

namespace {

// cor = S_new^-1 (R_old p + t_old), one lane per point; last[k] = index of the point's last observation
__global__ __launch_bounds__(256) void k_reanchor(int n_points, const int32_t* __restrict__ last,
                                                  const int32_t* __restrict__ obs_frame,
                                                  const double* __restrict__ old_Rt,
                                                  const sim3::Sim3* __restrict__ new_states,
                                                  double* __restrict__ points) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_points) return;
  const int j = last[k];
  if (j < 0) return;  // never observed: keeps its coordinates (drawPTAMPoints.cpp:418)
  const int f = obs_frame[j];
  const double* P = old_Rt + (size_t)12 * f;
  const double p[3] = {points[3 * (size_t)k], points[3 * (size_t)k + 1], points[3 * (size_t)k + 2]};
  double rel[3];
  for (int i = 0; i < 3; ++i) rel[i] = P[3 * i] * p[0] + P[3 * i + 1] * p[1] + P[3 * i + 2] * p[2] + P[9 + i];
  const double* d = reinterpret_cast<const double*>(new_states + f);
  sim3::Sim3 S;
  S.q[0] = d[0]; S.q[1] = d[1]; S.q[2] = d[2]; S.q[3] = d[3];
  S.t[0] = d[4]; S.t[1] = d[5]; S.t[2] = d[6]; S.s = d[7];
  const sim3::Sim3 Si = sim3::inverse(S);
  double rot[3];
  sim3::quat_rot(Si.q, rel, rot);
  for (int i = 0; i < 3; ++i) points[3 * (size_t)k + i] = Si.s * rot[i] + Si.t[i];
}

}  // namespace

extern "C" int sim3opt_read_keyframe_bin(const char* path, int32_t* kf_id, double Rw2c[9],
                                         double twinc[3], int32_t* n_obs, uint32_t* point_ids,
                                         double* points_w, double* obs_uv, int32_t capacity) {
  if (!path || !n_obs) return SIM3OPT_ERR_ARG;
  FILE* f = std::fopen(path, "rb");
  if (!f) return SIM3OPT_ERR_IO;
  auto rd = [&](void* dst, size_t n) { return std::fread(dst, 1, n, f) == n; };
  int32_t id = 0, len = 0, n = -1;
  char name[256];
  double size2[2], cam[5], R[9], t[3];
  unsigned char fixed = 0;
  bool ok = rd(&id, 4) && rd(&len, 4) && len >= 0 && len < 200 && rd(name, (size_t)len) &&
            rd(size2, 16) && rd(cam, 40) && rd(R, 72) && rd(t, 24) && rd(&fixed, 1) && rd(&n, 4) &&
            n >= 0;
  if (!ok) { std::fclose(f); return SIM3OPT_ERR_IO; }
  if (kf_id) *kf_id = id;
  if (Rw2c) std::memcpy(Rw2c, R, 72);
  if (twinc) std::memcpy(twinc, t, 24);
  *n_obs = n;
  if (capacity >= n && n > 0) {
    for (int32_t k = 0; k < n; ++k) {
      uint32_t pid;
      double pw[3], cosang, uv[2];
      if (!(rd(&pid, 4) && rd(pw, 24) && rd(&cosang, 8) && rd(uv, 16))) { std::fclose(f); return SIM3OPT_ERR_IO; }
      if (point_ids) point_ids[k] = pid;
      if (points_w) std::memcpy(points_w + 3 * (size_t)k, pw, 24);
      if (obs_uv) std::memcpy(obs_uv + 2 * (size_t)k, uv, 16);
    }
  }
  std::fclose(f);
  return SIM3OPT_OK;
}

extern "C" int sim3opt_reanchor_points(int32_t n_frames, const double* old_Rt,
                                       const double* new_states, int32_t n_points, double* points,
                                       int32_t n_obs, const int32_t* obs_frame,
                                       const int32_t* obs_point, int32_t device) {
  if (n_frames < 1 || n_points < 0 || n_obs < 0 || !old_Rt || !new_states || (n_points && !points) ||
      (n_obs && (!obs_frame || !obs_point)))
    return SIM3OPT_ERR_ARG;
  std::vector<int32_t> last((size_t)(n_points > 0 ? n_points : 1), -1);
  for (int32_t j = 0; j < n_obs; ++j) {  // observation order decides which keyframe a point follows
    if (obs_frame[j] < 0 || obs_frame[j] >= n_frames || obs_point[j] < 0 || obs_point[j] >= n_points)
      return SIM3OPT_ERR_ARG;
    last[obs_point[j]] = j;
  }
  if (n_points == 0) return SIM3OPT_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SIM3OPT_ERR_NO_DEVICE;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return SIM3OPT_ERR_HIP;
  int32_t *d_last = nullptr, *d_of = nullptr;
  double *d_rt = nullptr, *d_pts = nullptr;
  sim3::Sim3* d_st = nullptr;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  chk(hipMalloc((void**)&d_last, sizeof(int32_t) * (size_t)n_points));
  chk(hipMalloc((void**)&d_of, sizeof(int32_t) * (size_t)(n_obs > 0 ? n_obs : 1)));
  chk(hipMalloc((void**)&d_rt, sizeof(double) * 12 * (size_t)n_frames));
  chk(hipMalloc((void**)&d_st, sizeof(sim3::Sim3) * (size_t)n_frames));
  chk(hipMalloc((void**)&d_pts, sizeof(double) * 3 * (size_t)n_points));
  if (e == hipSuccess) {
    chk(hipMemcpy(d_last, last.data(), sizeof(int32_t) * (size_t)n_points, hipMemcpyHostToDevice));
    if (n_obs) chk(hipMemcpy(d_of, obs_frame, sizeof(int32_t) * (size_t)n_obs, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_rt, old_Rt, sizeof(double) * 12 * (size_t)n_frames, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_st, new_states, sizeof(double) * 8 * (size_t)n_frames, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_pts, points, sizeof(double) * 3 * (size_t)n_points, hipMemcpyHostToDevice));
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_reanchor, dim3((n_points + 255) / 256), dim3(256), 0, 0, n_points, d_last,
                       d_of, d_rt, d_st, d_pts);
    chk(hipGetLastError());
    chk(hipMemcpy(points, d_pts, sizeof(double) * 3 * (size_t)n_points, hipMemcpyDeviceToHost));
  }
  (void)hipFree(d_last); (void)hipFree(d_of); (void)hipFree(d_rt); (void)hipFree(d_st); (void)hipFree(d_pts);
  return e == hipSuccess ? SIM3OPT_OK : SIM3OPT_ERR_HIP;
}

extern "C" int sim3opt_write_g2o(sim3opt_graph* g, const char* path) {
  if (!g || !path) return SIM3OPT_ERR_ARG;
  const int32_t nv = sim3opt_num_vertices(g), ne = sim3opt_num_edges(g);
  std::vector<double> st(8 * (size_t)(nv > 0 ? nv : 1));
  int rc = sim3opt_get_vertices(g, st.data());
  if (rc != SIM3OPT_OK) return rc;
  FILE* f = std::fopen(path, "w");
  if (!f) return SIM3OPT_ERR_IO;
  // exact small-angle coefficient: with the as-written one log() and exp() are not inverses for
  // rotation angles between 1e-5 and 4.5e-3 rad, which would corrupt the round trip
  const sim3::Opts o{1e-5, 0, 1};
  auto put7 = [&](const double* a) {
    sim3::Sim3 S;
    S.q[0] = a[0]; S.q[1] = a[1]; S.q[2] = a[2]; S.q[3] = a[3];
    S.t[0] = a[4]; S.t[1] = a[5]; S.t[2] = a[6]; S.s = a[7];
    double lv[7];
    sim3::log(sim3::inverse(S), o, lv);  // g2o writes cam2world.log()
    for (int i = 0; i < 7; ++i) std::fprintf(f, " %.17g", lv[i]);
  };
  // ids: dense 0..nv-1 as in the reference (kitti_surf.cpp:604, :618); vertex 0 fixed (:613-616)
  for (int32_t k = 0; k < nv; ++k) {
    std::fprintf(f, "VERTEX_SIM3:EXPMAP %d", k);
    put7(&st[8 * (size_t)k]);
    std::fprintf(f, " 1 1 0 0\n");  // focal length and principal point of camera 1 (unused by EdgeSim3)
  }
  std::fprintf(f, "FIX 0\n");
  for (int32_t k = 0; k < ne; ++k) {
    int32_t a, b;
    double m[8];
    rc = sim3opt_get_edge(g, k, &a, &b, m);
    if (rc != SIM3OPT_OK) { std::fclose(f); return rc; }
    std::fprintf(f, "EDGE_SIM3:EXPMAP %d %d", a, b);
    put7(m);
    for (int i = 0; i < 7; ++i)
      for (int j = i; j < 7; ++j) std::fprintf(f, " %d", i == j ? 1 : 0);  // identity information (:592)
    std::fprintf(f, "\n");
  }
  std::fclose(f);
  return SIM3OPT_OK;
}

// BAL problem file of figureKITTIBA's hand-off to ba_demo: SaveBALFile, drawPTAMPoints.cpp:218-283
// (format of the ceres-solver BAL reader: "cams points obs", one "cam point u v" line per
// observation, then 9 numbers per camera -- angle-axis of R_w2c, t_w2c, f, k1, k2 -- and 3 per point,
// one per line, %.16g).  Like the reference, the camera frame stays right-down-forward and the
// observations keep their principal point (drawPTAMPoints.cpp:214-215, :255).  Point ids must be
// exactly 0..n_points-1 (the reference exits otherwise, :243-247; here: SIM3OPT_ERR_ARG).
extern "C" int sim3opt_write_bal(const char* path, int32_t n_cams, const double* Rw2c,
                                 const double* tw2c, const double f_k1_k2[3], int32_t n_points,
                                 const double* points, int32_t n_obs, const int32_t* obs_cam,
                                 const int32_t* obs_point, const double* obs_uv) {
  if (!path || n_cams < 1 || n_points < 1 || n_obs < 1 || !Rw2c || !tw2c || !f_k1_k2 || !points ||
      !obs_cam || !obs_point || !obs_uv)
    return SIM3OPT_ERR_ARG;
  int32_t lo = n_points, hi = -1;
  for (int32_t j = 0; j < n_obs; ++j) {
    if (obs_cam[j] < 0 || obs_cam[j] >= n_cams) return SIM3OPT_ERR_ARG;
    lo = obs_point[j] < lo ? obs_point[j] : lo;
    hi = obs_point[j] > hi ? obs_point[j] : hi;
  }
  if (lo != 0 || hi != n_points - 1) return SIM3OPT_ERR_ARG;
  FILE* f = std::fopen(path, "w");
  if (!f) return SIM3OPT_ERR_IO;
  std::fprintf(f, "%d %d %d\n", n_cams, n_points, n_obs);
  for (int32_t j = 0; j < n_obs; ++j)
    std::fprintf(f, "%d %d %g %g\n", obs_cam[j], obs_point[j], obs_uv[2 * (size_t)j],
                 obs_uv[2 * (size_t)j + 1]);
  for (int32_t c = 0; c < n_cams; ++c) {
    double q[4];  // xyzw, w >= 0 (rotro2qr, drawPTAMPoints.cpp:204-211)
    sim3::quat_from_R(Rw2c + 9 * (size_t)c, q);
    if (q[3] < 0)
      for (double& v : q) v = -v;
    // quaternion -> angle-axis (ceres QuaternionToAngleAxis, called at drawPTAMPoints.cpp:256)
    const double s2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    double k = 2.0;
    if (s2 > 0.0) {
      const double sn = std::sqrt(s2);
      k = 2.0 * std::atan2(sn, q[3]) / sn;
    }
    const double cam[9] = {k * q[0], k * q[1], k * q[2], tw2c[3 * (size_t)c], tw2c[3 * (size_t)c + 1],
                           tw2c[3 * (size_t)c + 2], f_k1_k2[0], f_k1_k2[1], f_k1_k2[2]};
    for (double v : cam) std::fprintf(f, "%.16g\n", v);
  }
  for (size_t k = 0; k < 3 * (size_t)n_points; ++k) std::fprintf(f, "%.16g\n", points[k]);
  return std::fclose(f) == 0 ? SIM3OPT_OK : SIM3OPT_ERR_IO;
}
