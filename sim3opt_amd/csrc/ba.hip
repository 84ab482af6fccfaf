// ba.hip -- the reference's `ba_demo` (bal_example.cpp:44-243) on the GPU: Levenberg-Marquardt bundle
// adjustment of SE(3) cameras and 3-D points over pinhole observations with a Huber kernel.
//
// What it replaces (all third-party g2o code; restated in oracle/ba_oracle.py, [upstream-recall]):
//   g2o::VertexSE3Expmap (T_w2c, update T <- exp([omega, upsilon]) T)       bal_example.cpp:112-118, :160-175
//   g2o::VertexSBAPointXYZ, setMarginalized(true)                            :120-130
//   g2o::EdgeProjectXYZ2UV + CameraParameters(f, pp, 0), information I/s^2    :90-97, :134-158
//   g2o::RobustKernelHuber(2.5)                                              :149-153
//   BlockSolver_6_3 (Schur complement on the points) + LinearSolverEigen     :76-88
//   OptimizationAlgorithmLevenberg, optimize(maxIterations)                  :86-88, :207
//
// Device pipeline of one LM iteration (all FP64, fixed summation orders, no atomics):
//   k_ba_obs      thread / observation: e = uv - K (R p + t), analytic Jacobians, Huber weight;
//                 stores A = sqrt(w) J_cam (2x6), B = sqrt(w) J_point (2x3), es = sqrt(w) e
//   k_ba_points   thread / point: H_pp = lambda I + sum B^T B over its observations (list order),
//                 b_p = -sum B^T es, H_pp^-1
//   k_ba_obs2     thread / observation: Z = (A^T B) H_pp^-1 (6x3)
//   k_ba_reduced  wavefront / block (i,j) of the reduced camera system: S_ij = [i=j](lambda I + sum
//                 A^T A) - sum over the listed observation pairs Z_o1 (A^T B)_o2^T (lane = pair, fixed-order
//                 butterfly sum); the diagonal block's wavefront also forms g_i = b_c,i - sum Z_o b_p(o)
//   k_ldl*        the reduced camera system S dx_c = g solved exactly: the level-scheduled block
//                 Cholesky of direct.hpp / direct_kernels.hpp (written for the 7x7 blocks of the
//                 Sim(3) graphs; the 6x6 camera blocks are stored padded to 7x7 with a unit diagonal
//                 entry, which leaves the factorisation of the 6x6 part untouched)
//   k_ba_pcg      fallback when a factorisation would be too large: block-Jacobi PCG in ONE workgroup
//   k_ba_backsub  thread / point: dx_p = H_pp^-1 b_p - sum Z_o^T dx_c(cam(o))
//   k_ba_update   exp-map update of the cameras, additive update of the points (backup kept)
//   k_ba_chi2     thread / observation: robustified chi2, fixed-order block sums
// The host runs g2o's lambda policy on three scalars per trial, as in engine.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sim3opt.h"
#include "devmem.hpp"
#include "direct.hpp"

namespace sim3opt_bundle {

#define BA_HIPCHK(call)                                                     \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      err = std::string(#call) + ": " + hipGetErrorString(e_);              \
      return SIM3OPT_ERR_HIP;                                               \
    }                                                                       \
  } while (0)

constexpr int WG = 256;

struct Cam {  // T_w2c: unit quaternion (x y z w) and translation; 8th double pads to 64 bytes
  double q[4], t[3], pad;
};

struct Scal {
  double chi2, scale, maxdiag;
  int32_t pcg_iters, fail;  // fail: PCG breakdown or a non-positive pivot of the factorisation
  double pcg_rel;
};

// the exact block Cholesky kernels, instantiated for this translation unit
using DevScalars = Scal;
using sim3opt::DirectPlan;
#include "direct_kernels.hpp"

__device__ __forceinline__ void quat_to_R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// Eigen's Quaternion(Matrix3) (trace branch, else the largest diagonal entry)
__device__ __forceinline__ void R_to_quat(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double k = sqrt(tr + 1.0);
    q[3] = 0.5 * k; k = 0.5 / k;
    q[0] = (R[7] - R[5]) * k; q[1] = (R[2] - R[6]) * k; q[2] = (R[3] - R[1]) * k;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, l = (j + 1) % 3;
    double k = sqrt(R[4 * i] - R[4 * j] - R[4 * l] + 1.0);
    q[i] = 0.5 * k; k = 0.5 / k;
    q[3] = (R[3 * l + j] - R[3 * j + l]) * k;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * k;
    q[l] = (R[3 * l + i] + R[3 * i + l]) * k;
  }
}

struct ObsArgs {
  int32_t n_obs;
  const int32_t* oc;
  const int32_t* op;
  const double* uv;
  const Cam* cams;
  const double* pts;
  double f, cx, cy, omega, huber;
};

// residual of one observation: e = uv - K (R p + t); X = camera-frame point
__device__ __forceinline__ void ba_residual(const ObsArgs& A, int o, double R[9], double X[3], double e[2]) {
  const Cam c = A.cams[A.oc[o]];
  const double* p = A.pts + (size_t)3 * A.op[o];
  quat_to_R(c.q, R);
#pragma unroll
  for (int i = 0; i < 3; ++i) X[i] = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2] + c.t[i];
  e[0] = A.uv[2 * (size_t)o] - (A.f * X[0] / X[2] + A.cx);
  e[1] = A.uv[2 * (size_t)o + 1] - (A.f * X[1] / X[2] + A.cy);
}

// g2o RobustKernelHuber on e2 = e^T Omega e
__device__ __forceinline__ void ba_huber(double e2, double delta, double& rho, double& w) {
  if (delta <= 0.0 || e2 <= delta * delta) {
    rho = e2;
    w = 1.0;
  } else {
    const double sq = sqrt(e2);
    rho = 2 * sq * delta - delta * delta;
    w = delta / sq;
  }
}

// per observation: 20 doubles [A (2x6 row-major), B (2x3 row-major), es (2)], all scaled by sqrt(w Omega)
__global__ __launch_bounds__(WG) void k_ba_obs(ObsArgs A, double* __restrict__ lin) {
  const int o = blockIdx.x * WG + threadIdx.x;
  if (o >= A.n_obs) return;
  double R[9], X[3], e[2];
  ba_residual(A, o, R, X, e);
  const double x = X[0], y = X[1], z = X[2], f = A.f, z2 = z * z;
  double rho, w;
  ba_huber(A.omega * (e[0] * e[0] + e[1] * e[1]), A.huber, rho, w);
  const double sw = sqrt(w * A.omega);
  // EdgeProjectXYZ2UV::linearizeOplus (analytic): J_cam over [omega, upsilon]
  double Jc[12] = {x * y / z2 * f, -(1 + x * x / z2) * f, y / z * f, -1.0 / z * f, 0.0, x / z2 * f,
                   (1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0.0, -1.0 / z * f, y / z2 * f};
  // J_point = -1/z [[f, 0, -f x/z], [0, f, -f y/z]] R
  const double t0[3] = {f, 0.0, -x / z * f}, t1[3] = {0.0, f, -y / z * f};
  double Jp[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Jp[c] = -(t0[0] * R[c] + t0[1] * R[3 + c] + t0[2] * R[6 + c]) / z;
    Jp[3 + c] = -(t1[0] * R[c] + t1[1] * R[3 + c] + t1[2] * R[6 + c]) / z;
  }
  double* d = lin + (size_t)20 * o;
#pragma unroll
  for (int i = 0; i < 12; ++i) d[i] = sw * Jc[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[12 + i] = sw * Jp[i];
  d[18] = sw * e[0];
  d[19] = sw * e[1];
}

// per point: Hinv (9, symmetric), bp (3), undamped max diagonal
__global__ __launch_bounds__(WG) void k_ba_points(int np, const int32_t* __restrict__ pptr,
                                                  const int32_t* __restrict__ pobs,
                                                  const double* __restrict__ lin, double lambda,
                                                  double* __restrict__ Hinv, double* __restrict__ bp,
                                                  double* __restrict__ pdmax) {
  const int p = blockIdx.x * WG + threadIdx.x;
  if (p >= np) return;
  double H[6] = {0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz
  double b[3] = {0, 0, 0};
  for (int k = pptr[p]; k < pptr[p + 1]; ++k) {
    const double* d = lin + (size_t)20 * pobs[k];
    const double* B = d + 12;
    H[0] += B[0] * B[0] + B[3] * B[3];
    H[1] += B[0] * B[1] + B[3] * B[4];
    H[2] += B[0] * B[2] + B[3] * B[5];
    H[3] += B[1] * B[1] + B[4] * B[4];
    H[4] += B[1] * B[2] + B[4] * B[5];
    H[5] += B[2] * B[2] + B[5] * B[5];
#pragma unroll
    for (int c = 0; c < 3; ++c) b[c] -= B[c] * d[18] + B[3 + c] * d[19];
  }
  pdmax[p] = fmax(H[0], fmax(H[3], H[5]));
  const double a = H[0] + lambda, bb = H[1], c = H[2], dd = H[3] + lambda, ee = H[4], ff = H[5] + lambda;
  // inverse of the symmetric 3x3 by cofactors
  const double c00 = dd * ff - ee * ee, c01 = c * ee - bb * ff, c02 = bb * ee - c * dd;
  const double det = a * c00 + bb * c01 + c * c02;
  const double id = 1.0 / det;
  double* Hi = Hinv + (size_t)9 * p;
  Hi[0] = c00 * id; Hi[1] = c01 * id; Hi[2] = c02 * id;
  Hi[3] = c01 * id; Hi[4] = (a * ff - c * c) * id; Hi[5] = (bb * c - a * ee) * id;
  Hi[6] = c02 * id; Hi[7] = (bb * c - a * ee) * id; Hi[8] = (a * dd - bb * bb) * id;
  bp[3 * (size_t)p] = b[0]; bp[3 * (size_t)p + 1] = b[1]; bp[3 * (size_t)p + 2] = b[2];
}

// per observation: Z = (A^T B) Hinv_p, 6x3 row-major
__global__ __launch_bounds__(WG) void k_ba_obs2(int n_obs, const int32_t* __restrict__ op,
                                                const double* __restrict__ lin,
                                                const double* __restrict__ Hinv, double* __restrict__ Z) {
  const int o = blockIdx.x * WG + threadIdx.x;
  if (o >= n_obs) return;
  const double* d = lin + (size_t)20 * o;
  const double* Hi = Hinv + (size_t)9 * op[o];
  double* z = Z + (size_t)18 * o;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double y0 = d[r] * d[12] + d[6 + r] * d[15];
    const double y1 = d[r] * d[13] + d[6 + r] * d[16];
    const double y2 = d[r] * d[14] + d[6 + r] * d[17];
#pragma unroll
    for (int c = 0; c < 3; ++c) z[3 * r + c] = y0 * Hi[c] + y1 * Hi[3 + c] + y2 * Hi[6 + c];
  }
}

// reduced camera system: one wavefront per block k = (row i, column j), stored as a 7x7 column-major
// block (entry (r, c) at r + 7c) whose 6x6 part is S_ij and whose 7th row / column is that of the
// identity; camera vectors are 7 per camera with a zero pad.  The lanes share out the block's list of
// observation pairs (lane = pair, 64 at a time, every load independent), each keeping a private 6x6
// sum; a butterfly over the wavefront adds them in a fixed order.  The diagonal block's list holds
// every observation of the camera as (o, o): H_cc, b_c and g are formed in the same pass.
__global__ __launch_bounds__(WG) void k_ba_reduced(int nblk, const int32_t* __restrict__ brow,
                                                   const int32_t* __restrict__ bcol,
                                                   const int32_t* __restrict__ sptr,
                                                   const int32_t* __restrict__ sa,
                                                   const int32_t* __restrict__ sb,
                                                   const int32_t* __restrict__ op,
                                                   const double* __restrict__ lin,
                                                   const double* __restrict__ Z,
                                                   const double* __restrict__ bp,
                                                   const uint8_t* __restrict__ fixed, double lambda,
                                                   double* __restrict__ S, double* __restrict__ g,
                                                   double* __restrict__ bc, double* __restrict__ cdmax) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= nblk) return;
  const int i = brow[k], j = bcol[k];
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  const bool pad = r == 6 || c == 6;
  const bool dg = i == j;
  if (fixed[i] | fixed[j]) {  // setFixed(true): the camera leaves the system (identity row, zero rhs)
    if (lane < 49) S[(size_t)49 * k + lane] = (dg && r == c) ? 1.0 : 0.0;
    if (dg && lane < 7) {
      g[7 * (size_t)i + lane] = 0.0;
      bc[7 * (size_t)i + lane] = 0.0;
      cdmax[7 * (size_t)i + lane] = 0.0;
    }
    return;
  }
  double acc[36], hd[6], bcv[6], gbv[6];
#pragma unroll
  for (int q = 0; q < 36; ++q) acc[q] = 0.0;
#pragma unroll
  for (int q = 0; q < 6; ++q) hd[q] = bcv[q] = gbv[q] = 0.0;
  for (int e = sptr[k] + lane; e < sptr[k + 1]; e += 64) {
    const int o1 = sa[e], o2 = sb[e];
    const double* z = Z + (size_t)18 * o1;
    const double* d = lin + (size_t)20 * o2;
    double zz[18], dd[20];
#pragma unroll
    for (int q = 0; q < 18; ++q) zz[q] = z[q];
#pragma unroll
    for (int q = 0; q < 20; ++q) dd[q] = d[q];
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {  // Y_o2 row cc = (A^T B) row cc
      const double y0 = dd[cc] * dd[12] + dd[6 + cc] * dd[15];
      const double y1 = dd[cc] * dd[13] + dd[6 + cc] * dd[16];
      const double y2 = dd[cc] * dd[14] + dd[6 + cc] * dd[17];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) acc[6 * rr + cc] -= zz[3 * rr] * y0 + zz[3 * rr + 1] * y1 + zz[3 * rr + 2] * y2;
    }
    if (dg && o1 == o2) {
      const double* b3 = bp + (size_t)3 * op[o1];
      const double b0 = b3[0], b1 = b3[1], b2 = b3[2];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) acc[6 * rr + cc] += dd[rr] * dd[cc] + dd[6 + rr] * dd[6 + cc];
        hd[rr] += dd[rr] * dd[rr] + dd[6 + rr] * dd[6 + rr];
        bcv[rr] -= dd[rr] * dd[18] + dd[6 + rr] * dd[19];
        gbv[rr] -= zz[3 * rr] * b0 + zz[3 * rr + 1] * b1 + zz[3 * rr + 2] * b2;
      }
    }
  }
  // butterfly sums (every lane ends with the total), then lane = entry picks its value
  double mine = 0.0, vh = 0.0, vb = 0.0, vg = 0.0;
  const int idx = pad ? 0 : 6 * r + c;
#pragma unroll
  for (int q = 0; q < 36; ++q) {
    double v = acc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    mine = idx == q ? v : mine;
  }
  if (dg) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      double h = hd[q], b = bcv[q], gg = gbv[q];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        h += __shfl_xor(h, off);
        b += __shfl_xor(b, off);
        gg += __shfl_xor(gg, off);
      }
      if (lane == q) { vh = h; vb = b; vg = gg; }
    }
    if (lane < 7) {
      // undamped diagonal for lambda_0 (computeLambdaInit looks at every vertex's Hessian diagonal)
      cdmax[7 * (size_t)i + lane] = lane < 6 ? vh : 0.0;
      bc[7 * (size_t)i + lane] = lane < 6 ? vb : 0.0;
      g[7 * (size_t)i + lane] = lane < 6 ? vb + vg : 0.0;
    }
    if (r == c) mine += lambda;
  }
  if (lane < 49) S[(size_t)49 * k + lane] = pad ? ((dg && r == c) ? 1.0 : 0.0) : mine;
}

// Fallback: block-Jacobi PCG on S x = g inside ONE workgroup (rows = cameras, padded 7x7 blocks,
// block-CSR with the diagonal block first in every row).  Wavefront per block row in the SpMV, thread
// per unknown in the vector steps; dot products through LDS in a fixed order.
__global__ __launch_bounds__(1024) void k_ba_pcg(int nc, const int32_t* __restrict__ rptr,
                                                 const int32_t* __restrict__ cidx,
                                                 const double* __restrict__ S,
                                                 const double* __restrict__ g, double* __restrict__ x,
                                                 double* __restrict__ rv, double* __restrict__ zv,
                                                 double* __restrict__ pv, double* __restrict__ qv,
                                                 double* __restrict__ Dinv, int max_iter, double tol2,
                                                 Scal* sc) {
  __shared__ double red[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int n = 7 * nc;
  auto dot = [&](const double* a, const double* b) {
    double s = 0.0;
    for (int i = tid; i < n; i += blockDim.x) s += a[i] * b[i];
    red[tid] = s;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    const double v = red[0];
    __syncthreads();
    return v;
  };
  // D^-1: thread per camera, Gauss-Jordan on the diagonal block (stored row-major here)
  bool spd = true;
  for (int i = tid; i < nc; i += blockDim.x) {
    double a[7][7];
    const double* blk = S + (size_t)49 * rptr[i];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int c = 0; c < 7; ++c) a[r][c] = blk[r + 7 * c];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (!(a[k][k] > 0.0)) spd = false;
      const double d = 1.0 / a[k][k];
#pragma unroll
      for (int jj = 0; jj < 7; ++jj)
        if (jj != k) a[k][jj] *= d;
#pragma unroll
      for (int ii = 0; ii < 7; ++ii)
        if (ii != k) {
          const double f = a[ii][k];
#pragma unroll
          for (int jj = 0; jj < 7; ++jj)
            if (jj != k) a[ii][jj] -= f * a[k][jj];
          a[ii][k] = -f * d;
        }
      a[k][k] = d;
    }
    double* dst = Dinv + (size_t)49 * i;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int c = 0; c < 7; ++c) dst[7 * r + c] = a[r][c];
  }
  if (!spd) sc->fail = 1;
  for (int i = tid; i < n; i += blockDim.x) {
    x[i] = 0.0;
    rv[i] = g[i];
  }
  __syncthreads();
  auto precond = [&]() {  // z = D^-1 r
    for (int i = tid; i < n; i += blockDim.x) {
      const int cam = i / 7, rr = i % 7;
      const double* d = Dinv + (size_t)49 * cam + 7 * rr;
      const double* r7 = rv + (size_t)7 * cam;
      zv[i] = d[0] * r7[0] + d[1] * r7[1] + d[2] * r7[2] + d[3] * r7[3] + d[4] * r7[4] + d[5] * r7[5] + d[6] * r7[6];
    }
    __syncthreads();
  };
  precond();
  for (int i = tid; i < n; i += blockDim.x) pv[i] = zv[i];
  __syncthreads();
  double rz = dot(rv, zv);
  const double rz0 = rz;
  int it = 0;
  bool fail = false;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  while (it < max_iter && rz > tol2 * rz0 && rz > 0.0) {
    // q = S p : wavefront per block row, lane = entry (r, c) of the column-major block
    for (int row = wave; row < nc; row += nw) {
      double acc = 0.0;
      for (int k = rptr[row]; k < rptr[row + 1]; ++k)
        acc += S[(size_t)49 * k + l49] * pv[(size_t)7 * cidx[k] + c];
      const double s = ldl_sum_over_c(lane < 49 ? acc : 0.0, r);
      if (lane < 7) qv[(size_t)7 * row + lane] = s;
    }
    __syncthreads();
    const double pq = dot(pv, qv);
    if (!(pq > 0.0) || !(pq < DBL_MAX)) { fail = true; break; }
    const double alpha = rz / pq;
    for (int i = tid; i < n; i += blockDim.x) {
      x[i] += alpha * pv[i];
      rv[i] -= alpha * qv[i];
    }
    __syncthreads();
    precond();
    const double rzn = dot(rv, zv);
    const double beta = rzn / rz;
    for (int i = tid; i < n; i += blockDim.x) pv[i] = zv[i] + beta * pv[i];
    __syncthreads();
    rz = rzn;
    ++it;
  }
  if (tid == 0) {
    sc->pcg_iters = it;
    if (fail || !(rz >= 0.0)) sc->fail = 1;
    sc->pcg_rel = rz0 > 0 ? sqrt(fabs(rz) / rz0) : 0.0;
  }
}

// per point: dx_p = Hinv b_p - sum_o Z_o^T dx_c(cam(o))
__global__ __launch_bounds__(WG) void k_ba_backsub(int np, const int32_t* __restrict__ pptr,
                                                   const int32_t* __restrict__ pobs,
                                                   const int32_t* __restrict__ oc,
                                                   const double* __restrict__ Hinv,
                                                   const double* __restrict__ bp,
                                                   const double* __restrict__ Z,
                                                   const double* __restrict__ xc, double* __restrict__ xp) {
  const int p = blockIdx.x * WG + threadIdx.x;
  if (p >= np) return;
  const double* Hi = Hinv + (size_t)9 * p;
  const double* b = bp + (size_t)3 * p;
  double d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = Hi[3 * c] * b[0] + Hi[3 * c + 1] * b[1] + Hi[3 * c + 2] * b[2];
  for (int k = pptr[p]; k < pptr[p + 1]; ++k) {
    const int o = pobs[k];
    const double* z = Z + (size_t)18 * o;
    const double* x6 = xc + (size_t)7 * oc[o];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += z[3 * r + c] * x6[r];
      d[c] -= s;
    }
  }
  xp[3 * (size_t)p] = d[0]; xp[3 * (size_t)p + 1] = d[1]; xp[3 * (size_t)p + 2] = d[2];
}

// VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp([omega, upsilon]) T; points: p += dx
__global__ __launch_bounds__(WG) void k_ba_update(int nc, int np, const double* __restrict__ xc,
                                                  const double* __restrict__ xp,
                                                  const uint8_t* __restrict__ fixed, Cam* cams, double* pts,
                                                  const Scal* sc) {
  if (sc->fail) return;  // the host rejects the trial
  const int t = blockIdx.x * WG + threadIdx.x;
  if (t < nc && !fixed[t]) {
    const double* u = xc + (size_t)7 * t;
    const double th = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double Om[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
    double Om2[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Om2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    double R[9], V[9];
    if (th < 1e-5) {  // se3quat.h: R = I + Omega + Omega^2, V = R
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = Om[i] + Om2[i];
      R[0] += 1; R[4] += 1; R[8] += 1;
#pragma unroll
      for (int i = 0; i < 9; ++i) V[i] = R[i];
    } else {
      const double a = sin(th) / th, b = (1 - cos(th)) / (th * th), c = (th - sin(th)) / (th * th * th);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        R[i] = a * Om[i] + b * Om2[i];
        V[i] = b * Om[i] + c * Om2[i];
      }
      R[0] += 1; R[4] += 1; R[8] += 1;
      V[0] += 1; V[4] += 1; V[8] += 1;
    }
    Cam cm = cams[t];
    double Rc[9], Rn[9];
    quat_to_R(cm.q, Rc);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rn[3 * i + j] = R[3 * i] * Rc[j] + R[3 * i + 1] * Rc[3 + j] + R[3 * i + 2] * Rc[6 + j];
    double tn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      tn[i] = R[3 * i] * cm.t[0] + R[3 * i + 1] * cm.t[1] + R[3 * i + 2] * cm.t[2] +
              V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    double q[4];
    R_to_quat(Rn, q);
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) cm.q[i] = q[i] / nq;
#pragma unroll
    for (int i = 0; i < 3; ++i) cm.t[i] = tn[i];
    cams[t] = cm;
  }
  if (t < 3 * np) pts[t] += xp[t];
}

// robustified chi2 (fixed-order block partials) and, with x given, the scale term x.(lambda x + b)
__global__ __launch_bounds__(WG) void k_ba_chi2(ObsArgs A, double* __restrict__ partials) {
  __shared__ double sh[WG];
  double acc = 0.0;
  for (int o = blockIdx.x * WG + threadIdx.x; o < A.n_obs; o += gridDim.x * WG) {
    double R[9], X[3], e[2], rho, w;
    ba_residual(A, o, R, X, e);
    ba_huber(A.omega * (e[0] * e[0] + e[1] * e[1]), A.huber, rho, w);
    acc += rho;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(WG) void k_ba_scale(int n, const double* __restrict__ x,
                                                 const double* __restrict__ b, double lambda,
                                                 double* __restrict__ partials) {
  __shared__ double sh[WG];
  double acc = 0.0;
  for (int i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) acc += x[i] * (lambda * x[i] + b[i]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// out[0] = sum of pa (chi2) ; out[1] = sum of pb + sum of pc (scale) ; out[2] = max of the two diag arrays
__global__ __launch_bounds__(WG) void k_ba_final(const double* __restrict__ pa, int na,
                                                 const double* __restrict__ pb, int nb,
                                                 const double* __restrict__ pc, int ncn,
                                                 const double* __restrict__ d1, int n1,
                                                 const double* __restrict__ d2, int n2, Scal* sc) {
  __shared__ double sh[WG];
  auto sum = [&](const double* p, int n) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += WG) a += p[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = WG / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    const double v = sh[0];
    __syncthreads();
    return v;
  };
  if (pa) {
    const double v = sum(pa, na);
    if (threadIdx.x == 0) sc->chi2 = v;
  }
  if (pb) {
    const double v = sum(pb, nb) + sum(pc, ncn);
    if (threadIdx.x == 0) sc->scale = v;
  }
  if (d1) {
    double m = 0.0;
    for (int i = threadIdx.x; i < n1; i += WG) m = fmax(m, d1[i]);
    for (int i = threadIdx.x; i < n2; i += WG) m = fmax(m, d2[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int off = WG / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + off]);
      __syncthreads();
    }
    if (threadIdx.x == 0) sc->maxdiag = sh[0];
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Problem {
  std::vector<Cam> cams;
  std::vector<double> pts;  // 3 per point
  std::vector<int32_t> oc, op;
  std::vector<uint8_t> cam_fixed;
  std::vector<double> uv;
  double f = 718.856, cx = 607.1928, cy = 185.2157;  // kitti_surf.cpp:52-57, bal_example.cpp:90-91
  sim3opt_ba_options opt;
  std::vector<sim3opt_iter_stats> stats;
  std::string err;
  // device
  bool ready = false;
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  Cam *d_cams = nullptr, *d_cams_bk = nullptr;
  double *d_pts = nullptr, *d_pts_bk = nullptr, *d_uv = nullptr, *d_lin = nullptr, *d_Z = nullptr;
  double *d_Hinv = nullptr, *d_bp = nullptr, *d_pdmax = nullptr, *d_cdmax = nullptr;
  double *d_S = nullptr, *d_g = nullptr, *d_bc = nullptr, *d_xc = nullptr, *d_xp = nullptr;
  double *d_r = nullptr, *d_z = nullptr, *d_p = nullptr, *d_q = nullptr, *d_Dinv = nullptr;
  double *d_pa = nullptr, *d_pb = nullptr, *d_pc = nullptr;
  int32_t *d_oc = nullptr, *d_op = nullptr, *d_pptr = nullptr, *d_pobs = nullptr, *d_cptr = nullptr,
          *d_cobs = nullptr, *d_brow = nullptr, *d_bcol = nullptr, *d_sptr = nullptr, *d_sa = nullptr,
          *d_sb = nullptr, *d_rptr = nullptr;
  uint8_t* d_fixed = nullptr;
  Scal *d_sc = nullptr, *h_sc = nullptr;
  int32_t nblk = 0;
  int grid_chi = 1;
  DirectPlan dplan;  // exact factorisation of the reduced camera system (empty: PCG fallback)
  LdlArgs ldl{};
  bool use_direct = false;
  int ldl_wg_sub = LDL_WG_TOP;

  ~Problem() { release(); }
  void release() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : owned)
      if (p) sim3opt::dev_free(p);  // (the library's block cache: an out-of-memory hipMalloc elsewhere flushes it)
    owned.clear();
    if (h_sc) (void)hipHostFree(h_sc);
    h_sc = nullptr;
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
    ready = false;
  }
  int nc() const { return (int)cams.size(); }
  int np() const { return (int)(pts.size() / 3); }
  int no() const { return (int)oc.size(); }

  template <typename T>
  int up(T*& d, const std::vector<T>& h) {
    BA_HIPCHK(sim3opt::dev_malloc((void**)&d, sizeof(T) * std::max<size_t>(h.size(), 1)));
    owned.push_back(d);
    if (!h.empty()) BA_HIPCHK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return SIM3OPT_OK;
  }
  int alloc(double*& d, size_t n) {
    BA_HIPCHK(sim3opt::dev_malloc((void**)&d, sizeof(double) * std::max<size_t>(n, 1)));
    owned.push_back(d);
    BA_HIPCHK(hipMemset(d, 0, sizeof(double) * std::max<size_t>(n, 1)));
    return SIM3OPT_OK;
  }

  int initialize() {
    release();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      err = "no usable HIP device (libsim3opt has no CPU fallback)";
      return SIM3OPT_ERR_NO_DEVICE;
    }
    if (opt.device >= 0) {
      if (opt.device >= ndev) { err = "device ordinal out of range"; return SIM3OPT_ERR_ARG; }
      BA_HIPCHK(hipSetDevice(opt.device));
    }
    const int NC = nc(), NP = np(), NO = no();
    if (NC < 1 || NP < 1 || NO < 1) { err = "empty problem"; return SIM3OPT_ERR_STATE; }
    // observation lists per point / per camera, ascending observation index
    std::vector<int32_t> pptr(NP + 1, 0), cptr(NC + 1, 0), pobs(NO), cobs(NO);
    for (int o = 0; o < NO; ++o) { ++pptr[op[o] + 1]; ++cptr[oc[o] + 1]; }
    for (int p = 0; p < NP; ++p) pptr[p + 1] += pptr[p];
    for (int c = 0; c < NC; ++c) cptr[c + 1] += cptr[c];
    {
      std::vector<int32_t> fp(pptr.begin(), pptr.end() - 1), fc(cptr.begin(), cptr.end() - 1);
      for (int o = 0; o < NO; ++o) { pobs[fp[op[o]]++] = o; cobs[fc[oc[o]]++] = o; }
    }
    // reduced camera system: blocks (i, j) for cameras sharing a point (and every diagonal), with
    // the observation pairs behind each block in (point, o1, o2) order
    std::map<std::pair<int32_t, int32_t>, std::vector<std::pair<int32_t, int32_t>>> blocks;
    for (int c = 0; c < NC; ++c) blocks[{c, c}];
    for (int p = 0; p < NP; ++p)
      for (int a = pptr[p]; a < pptr[p + 1]; ++a)
        for (int b = pptr[p]; b < pptr[p + 1]; ++b)
          blocks[{oc[pobs[a]], oc[pobs[b]]}].push_back({pobs[a], pobs[b]});
    // block-CSR, diagonal block first in every row
    std::vector<int32_t> rptr(NC + 1, 0), brow, bcol, sptr(1, 0), sa, sb;
    for (int pass = 0; pass < 1; ++pass) {
      int32_t cur = -1;
      std::vector<std::pair<std::pair<int32_t, int32_t>, const std::vector<std::pair<int32_t, int32_t>>*>> rowblk;
      auto flush = [&]() {
        if (cur < 0) return;
        // diagonal first
        std::stable_sort(rowblk.begin(), rowblk.end(), [&](const auto& x, const auto& y) {
          const bool dx = x.first.second == cur, dy = y.first.second == cur;
          if (dx != dy) return dx;
          return x.first.second < y.first.second;
        });
        for (auto& rb : rowblk) {
          brow.push_back(rb.first.first);
          bcol.push_back(rb.first.second);
          for (auto& pr : *rb.second) { sa.push_back(pr.first); sb.push_back(pr.second); }
          sptr.push_back((int32_t)sa.size());
        }
        rptr[cur + 1] = (int32_t)brow.size();
        rowblk.clear();
      };
      for (auto& kv : blocks) {
        if (kv.first.first != cur) { flush(); cur = kv.first.first; }
        rowblk.push_back({kv.first, &kv.second});
      }
      flush();
    }
    for (int c = 0; c < NC; ++c) rptr[c + 1] = std::max(rptr[c + 1], rptr[c]);
    nblk = (int32_t)brow.size();
    BA_HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    int rc;
#define BCHK(call) do { rc = (call); if (rc) return rc; } while (0)
    BCHK(up(d_cams, cams));
    BA_HIPCHK(sim3opt::dev_malloc((void**)&d_cams_bk, sizeof(Cam) * NC)); owned.push_back(d_cams_bk);
    BCHK(up(d_pts, pts));
    BCHK(alloc(d_pts_bk, 3 * (size_t)NP));
    BCHK(up(d_uv, uv));
    BCHK(up(d_oc, oc)); BCHK(up(d_op, op));
    BCHK(up(d_pptr, pptr)); BCHK(up(d_pobs, pobs)); BCHK(up(d_cptr, cptr)); BCHK(up(d_cobs, cobs));
    BCHK(up(d_brow, brow)); BCHK(up(d_bcol, bcol)); BCHK(up(d_sptr, sptr)); BCHK(up(d_sa, sa)); BCHK(up(d_sb, sb));
    BCHK(up(d_rptr, rptr));
    cam_fixed.resize(NC, 0);
    BCHK(up(d_fixed, cam_fixed));
    // the reduced camera system's factorisation plan (nested dissection, level schedule: direct.hpp)
    use_direct = false;
    dplan = DirectPlan();
    if (opt.linear_solver != 0) {
      int64_t max_pairs = 8000000;
      if (const char* ev = std::getenv("SIM3OPT_BA_MAX_PAIRS")) max_pairs = std::atoll(ev);
      // bottom subtrees of up to a sixth of the cameras, 8 wavefronts each: the reduced system of a
      // camera chain is a wide band (tracks span several keyframes), its separators are several
      // columns wide and the top of the tree is expensive -- more of it goes to the parallel groups
      // than for the pose graphs (profiles/r2_ba_sweep.log)
      int32_t subtree = std::max(16, NC / 6);
      if (const char* ev = std::getenv("SIM3OPT_BA_SUBTREE")) subtree = std::atoi(ev);  // tuning knobs
      if (const char* ev = std::getenv("SIM3OPT_BA_WG_SUB")) ldl_wg_sub = std::max(64, std::min(LDL_WG_TOP, std::atoi(ev) / 64 * 64));
      std::string why;
      if (sim3opt::build_direct_plan(NC, rptr.data(), bcol.data(), max_pairs, subtree, dplan, why, ldl_wg_sub / 64)) {
        int32_t *pperm, *pcolptr, *plrow, *plcol, *psrcptr, *psrc, *ppairptr, *ppa, *ppb, *ppcol, *pgptr, *plcolp,
            *prptr, *pcells, *pbord, *pbrow, *ptpre, *ptprey;
        BCHK(up(pperm, dplan.perm)); BCHK(up(pcolptr, dplan.colptr)); BCHK(up(plrow, dplan.lrow));
        BCHK(up(plcol, dplan.lcol)); BCHK(up(psrcptr, dplan.srcptr)); BCHK(up(psrc, dplan.src));
        BCHK(up(ppairptr, dplan.pairptr)); BCHK(up(ppa, dplan.pa)); BCHK(up(ppb, dplan.pb));
        BCHK(up(ppcol, dplan.pcol)); BCHK(up(pgptr, dplan.gptr)); BCHK(up(plcolp, dplan.lcolp));
        BCHK(up(prptr, dplan.rptr)); BCHK(up(pcells, dplan.cells));
        BCHK(up(pbord, dplan.bord)); BCHK(up(pbrow, dplan.brow));
        BCHK(up(ptpre, dplan.tpre)); BCHK(up(ptprey, dplan.tprey));
        ldl.tpre = ptpre; ldl.tprey = ptprey;
        ldl.ntpre = (int32_t)dplan.tpre.size(); ldl.ntprey = (int32_t)dplan.tprey.size();
        ldl.perm = pperm; ldl.colptr = pcolptr; ldl.lrow = plrow; ldl.lcol = plcol; ldl.srcptr = psrcptr;
        ldl.src = psrc; ldl.pairptr = ppairptr; ldl.pa = ppa; ldl.pb = ppb; ldl.pcol = ppcol; ldl.gptr = pgptr;
        ldl.lcolp = plcolp; ldl.rptr = prptr; ldl.cells = pcells; ldl.bord = pbord; ldl.brow = pbrow;
        ldl.nb = NC;
        ldl.nL = (int32_t)dplan.nL;
        BCHK(alloc(ldl.Aperm, 49 * (size_t)dplan.nL)); BCHK(alloc(ldl.bp, 7 * (size_t)NC));
        BCHK(alloc(ldl.L, 49 * (size_t)dplan.nL)); BCHK(alloc(ldl.Dinv, 49 * (size_t)NC));
        BCHK(alloc(ldl.y, 7 * (size_t)NC)); BCHK(alloc(ldl.xp, 7 * (size_t)NC));
        ldl.dbg = nullptr;
        if (std::getenv("SIM3OPT_BA_TRACE")) {  // tuning aid: time stamps of the top group's levels / rounds
          double* p = nullptr;
          BCHK(alloc(p, 256));
          ldl.dbg = reinterpret_cast<long long*>(p);
        }
        ldl.lambda = 0.0;  // S carries the damping already
        use_direct = true;
        if (opt.verbose)
          std::fprintf(stderr, "sim3opt ba: exact block Cholesky of the reduced system: %d cameras, %lld blocks in L, "
                       "%lld block products, tree height %d, %d groups\n", NC, (long long)dplan.nL,
                       (long long)dplan.npairs, dplan.height, dplan.ngroups());
      } else if (opt.linear_solver == 1) {
        err = "linear_solver = 1: " + why;
        return SIM3OPT_ERR_ARG;
      } else if (opt.verbose) {
        std::fprintf(stderr, "sim3opt ba: no exact factorisation (%s): PCG\n", why.c_str());
      }
    }
    BCHK(alloc(d_lin, 20 * (size_t)NO)); BCHK(alloc(d_Z, 18 * (size_t)NO));
    BCHK(alloc(d_Hinv, 9 * (size_t)NP)); BCHK(alloc(d_bp, 3 * (size_t)NP)); BCHK(alloc(d_pdmax, NP));
    BCHK(alloc(d_cdmax, 7 * (size_t)NC));
    BCHK(alloc(d_S, 49 * (size_t)nblk)); BCHK(alloc(d_g, 7 * (size_t)NC)); BCHK(alloc(d_bc, 7 * (size_t)NC));
    BCHK(alloc(d_xc, 7 * (size_t)NC)); BCHK(alloc(d_xp, 3 * (size_t)NP));
    BCHK(alloc(d_r, 7 * (size_t)NC)); BCHK(alloc(d_z, 7 * (size_t)NC)); BCHK(alloc(d_p, 7 * (size_t)NC));
    BCHK(alloc(d_q, 7 * (size_t)NC)); BCHK(alloc(d_Dinv, 49 * (size_t)NC));
    grid_chi = std::max(1, std::min(1024, (NO + WG - 1) / WG));
    BCHK(alloc(d_pa, 1024)); BCHK(alloc(d_pb, 1024)); BCHK(alloc(d_pc, 1024));
#undef BCHK
    BA_HIPCHK(sim3opt::dev_malloc((void**)&d_sc, sizeof(Scal))); owned.push_back(d_sc);
    BA_HIPCHK(hipMemset(d_sc, 0, sizeof(Scal)));
    BA_HIPCHK(hipHostMalloc((void**)&h_sc, sizeof(Scal)));
    // the uploads and memsets above ran on the null stream, the kernels run on a non-blocking one:
    // order them (engine.hip ends its init the same way)
    BA_HIPCHK(hipDeviceSynchronize());
    ready = true;
    return SIM3OPT_OK;
  }

  ObsArgs oargs() const {
    return ObsArgs{no(), d_oc, d_op, d_uv, d_cams, d_pts, f, cx, cy,
                   1.0 / (opt.pixel_noise * opt.pixel_noise), opt.huber_delta};
  }
  int fetch() {
    BA_HIPCHK(hipMemcpyAsync(h_sc, d_sc, sizeof(Scal), hipMemcpyDeviceToHost, stream));
    BA_HIPCHK(hipStreamSynchronize(stream));
    return SIM3OPT_OK;
  }
  int chi2(double* out) {
    hipLaunchKernelGGL(k_ba_chi2, dim3(grid_chi), dim3(WG), 0, stream, oargs(), d_pa);
    hipLaunchKernelGGL(k_ba_final, dim3(1), dim3(WG), 0, stream, (const double*)d_pa, grid_chi,
                       (const double*)nullptr, 0, (const double*)nullptr, 0, (const double*)nullptr, 0,
                       (const double*)nullptr, 0, d_sc);
    BA_HIPCHK(hipGetLastError());
    int rc = fetch();
    if (rc) return rc;
    *out = h_sc->chi2;
    return SIM3OPT_OK;
  }

  int optimize(int max_iters) {
    stats.clear();
    const int NC = nc(), NP = np(), NO = no();
    const int go = (NO + WG - 1) / WG, gp = (NP + WG - 1) / WG;
    double lambda = 0.0, ni = 2.0;
    bool ok = true;
    int iters = 0;
    for (int it = 0; it < max_iters && ok; ++it) {
      sim3opt_iter_stats T{};
      double currentChi = 0.0;
      int rc = chi2(&currentChi);
      if (rc) return rc;
      T.chi2_before = currentChi;
      double tempChi = currentChi;
      hipLaunchKernelGGL(k_ba_obs, dim3(go), dim3(WG), 0, stream, oargs(), d_lin);
      double rho = 0.0;
      int qmax = 0;
      do {
        BA_HIPCHK(hipMemcpyAsync(d_cams_bk, d_cams, sizeof(Cam) * NC, hipMemcpyDeviceToDevice, stream));
        BA_HIPCHK(hipMemcpyAsync(d_pts_bk, d_pts, sizeof(double) * 3 * NP, hipMemcpyDeviceToDevice, stream));
        if (it == 0 && qmax == 0 && !(opt.user_lambda_init > 0)) {
          // computeLambdaInit: tau * max diagonal entry of the (undamped) Hessian over all vertices
          hipLaunchKernelGGL(k_ba_points, dim3(gp), dim3(WG), 0, stream, NP, d_pptr, d_pobs, d_lin, 1.0,
                             d_Hinv, d_bp, d_pdmax);
          hipLaunchKernelGGL(k_ba_obs2, dim3(go), dim3(WG), 0, stream, NO, d_op, d_lin, d_Hinv, d_Z);
          hipLaunchKernelGGL(k_ba_reduced, dim3((nblk + 3) / 4), dim3(WG), 0, stream, nblk, d_brow, d_bcol,
                             d_sptr, d_sa, d_sb, d_op, d_lin, d_Z, d_bp, d_fixed, 1.0, d_S, d_g, d_bc,
                             d_cdmax);
          hipLaunchKernelGGL(k_ba_final, dim3(1), dim3(WG), 0, stream, (const double*)nullptr, 0,
                             (const double*)nullptr, 0, (const double*)nullptr, 0, (const double*)d_pdmax, NP,
                             (const double*)d_cdmax, 7 * NC, d_sc);
          BA_HIPCHK(hipGetLastError());
          rc = fetch();
          if (rc) return rc;
          lambda = opt.tau * h_sc->maxdiag;
          ni = 2.0;
        } else if (it == 0 && qmax == 0) {
          lambda = opt.user_lambda_init;
          ni = 2.0;
        }
        BA_HIPCHK(hipMemsetAsync(&d_sc->pcg_iters, 0, 2 * sizeof(int32_t), stream));
        hipLaunchKernelGGL(k_ba_points, dim3(gp), dim3(WG), 0, stream, NP, d_pptr, d_pobs, d_lin, lambda,
                           d_Hinv, d_bp, d_pdmax);
        hipLaunchKernelGGL(k_ba_obs2, dim3(go), dim3(WG), 0, stream, NO, d_op, d_lin, d_Hinv, d_Z);
        hipLaunchKernelGGL(k_ba_reduced, dim3((nblk + 3) / 4), dim3(WG), 0, stream, nblk, d_brow, d_bcol,
                           d_sptr, d_sa, d_sb, d_op, d_lin, d_Z, d_bp, d_fixed, lambda, d_S, d_g, d_bc,
                           d_cdmax);
        if (use_direct) {
          ldl.vals = d_S; ldl.b = d_g; ldl.x = d_xc; ldl.sc = d_sc;
          hipLaunchKernelGGL(k_ldl_gather, dim3(std::max(1, std::min(1024, (ldl.nL + 3) / 4))), dim3(WG), 0, stream, ldl);
          const int ng = dplan.ngroups();
          if (ng > 1) hipLaunchKernelGGL((k_ldl<true, false>), dim3(ng - 1), dim3(ldl_wg_sub), 0, stream, ldl, 0);
          hipLaunchKernelGGL((k_ldl<true, true>), dim3(1), dim3(LDL_WG_TOP), 0, stream, ldl, ng - 1);
          if (ng > 1) hipLaunchKernelGGL((k_ldl<false, true>), dim3(ng - 1), dim3(ldl_wg_sub), 0, stream, ldl, 0);
          if (ldl.dbg) {
            long long h[256];
            BA_HIPCHK(hipStreamSynchronize(stream));
            BA_HIPCHK(hipMemcpy(h, ldl.dbg, sizeof(h), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "sim3opt ba: top group stamps [us] (level start, after A+B of each round, ..., down start, end):");
            for (long long i = 0; i < h[255] && i < 255; ++i) std::fprintf(stderr, " %.1f", (h[i] - h[0]) * 0.01);
            std::fprintf(stderr, "\n");
          }
        } else {
          hipLaunchKernelGGL(k_ba_pcg, dim3(1), dim3(1024), 0, stream, NC, d_rptr, d_bcol, d_S, d_g, d_xc, d_r,
                             d_z, d_p, d_q, d_Dinv, opt.pcg_max_iters > 0 ? opt.pcg_max_iters : 20 * NC + 100,
                             opt.pcg_rel_tol * opt.pcg_rel_tol, d_sc);
        }
        hipLaunchKernelGGL(k_ba_backsub, dim3(gp), dim3(WG), 0, stream, NP, d_pptr, d_pobs, d_oc, d_Hinv,
                           d_bp, d_Z, d_xc, d_xp);
        hipLaunchKernelGGL(k_ba_update, dim3((std::max(NC, 3 * NP) + WG - 1) / WG), dim3(WG), 0, stream, NC, NP,
                           d_xc, d_xp, d_fixed, d_cams, d_pts, (const Scal*)d_sc);
        // scale = x.(lambda x + b) over cameras and points
        const int gsc = std::max(1, std::min(1024, (7 * NC + WG - 1) / WG));
        const int gsp = std::max(1, std::min(1024, (3 * NP + WG - 1) / WG));
        hipLaunchKernelGGL(k_ba_scale, dim3(gsc), dim3(WG), 0, stream, 7 * NC, d_xc, d_bc, lambda, d_pb);
        hipLaunchKernelGGL(k_ba_scale, dim3(gsp), dim3(WG), 0, stream, 3 * NP, d_xp, d_bp, lambda, d_pc);
        hipLaunchKernelGGL(k_ba_chi2, dim3(grid_chi), dim3(WG), 0, stream, oargs(), d_pa);
        hipLaunchKernelGGL(k_ba_final, dim3(1), dim3(WG), 0, stream, (const double*)d_pa, grid_chi,
                           (const double*)d_pb, gsc, (const double*)d_pc, gsp, (const double*)nullptr, 0,
                           (const double*)nullptr, 0, d_sc);
        BA_HIPCHK(hipGetLastError());
        rc = fetch();
        if (rc) return rc;
        T.pcg_iters += h_sc->pcg_iters;
        T.pcg_rel_res = h_sc->pcg_rel;
        double scale = h_sc->scale;
        tempChi = h_sc->fail ? DBL_MAX : h_sc->chi2;
        if (h_sc->fail) scale = 0.0;
        rho = (currentChi - tempChi) / (scale + 1e-3);
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1.0 - std::pow(2 * rho - 1, 3);
          alpha = std::min(alpha, 2.0 / 3.0);
          lambda *= std::max(1.0 / 3.0, alpha);
          ni = 2.0;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2.0;
          BA_HIPCHK(hipMemcpyAsync(d_cams, d_cams_bk, sizeof(Cam) * NC, hipMemcpyDeviceToDevice, stream));
          BA_HIPCHK(hipMemcpyAsync(d_pts, d_pts_bk, sizeof(double) * 3 * NP, hipMemcpyDeviceToDevice, stream));
        }
        ++qmax;
      } while (rho < 0 && qmax < opt.max_trials);
      T.chi2_after = currentChi;
      T.lambda = lambda;
      T.rho = rho;
      T.trials = qmax;
      stats.push_back(T);
      ++iters;
      if (opt.verbose)
        std::fprintf(stderr, "ba iteration= %d\t chi2= %.9g\t lambda= %.6g\t levenbergIter= %d\t pcg= %d (rel %.1e)\n",
                     it, currentChi, lambda, qmax, T.pcg_iters, T.pcg_rel_res);
      if (qmax == opt.max_trials || rho == 0 || !std::isfinite(lambda)) ok = false;
    }
    BA_HIPCHK(hipStreamSynchronize(stream));
    BA_HIPCHK(hipMemcpy(cams.data(), d_cams, sizeof(Cam) * NC, hipMemcpyDeviceToHost));
    BA_HIPCHK(hipMemcpy(pts.data(), d_pts, sizeof(double) * 3 * NP, hipMemcpyDeviceToHost));
    return iters;
  }
};

}  // namespace sim3opt_bundle

// ------------------------------------------------------------------------------------------
// C-ABI (include/sim3opt.h, "bundle adjustment hand-off")
// ------------------------------------------------------------------------------------------
using sim3opt_bundle::Problem;
struct sim3opt_ba : Problem {};

extern "C" {

void sim3opt_ba_options_default(sim3opt_ba_options* o) {
  if (!o) return;
  o->huber_delta = 2.5;   // bal_example.cpp:151
  o->pixel_noise = 1.0;   // :62
  o->tau = 1e-5;
  o->user_lambda_init = 0.0;
  o->max_trials = 10;
  o->pcg_max_iters = 0;
  o->pcg_rel_tol = 1e-12;
  o->linear_solver = -1;
  o->device = -1;
  o->verbose = 0;
}

sim3opt_ba* sim3opt_ba_create(void) {
  sim3opt_ba* b = new (std::nothrow) sim3opt_ba();
  if (b) {
    sim3opt_ba_options_default(&b->opt);
    sim3opt::handle_count(+1);
  }
  return b;
}

void sim3opt_ba_destroy(sim3opt_ba* b) {
  if (!b) return;
  delete b;
  if (sim3opt::handle_count(-1) == 0) sim3opt::dev_cache_release();
}

const char* sim3opt_ba_last_error(const sim3opt_ba* b) { return b ? b->err.c_str() : "null problem"; }

int sim3opt_ba_set_options(sim3opt_ba* b, const sim3opt_ba_options* o) {
  if (!b || !o) return SIM3OPT_ERR_ARG;
  if (!(o->pixel_noise > 0) || o->max_trials < 1 || !(o->tau > 0) || !(o->pcg_rel_tol >= 0) || o->huber_delta < 0 ||
      o->linear_solver < -1 || o->linear_solver > 1) {
    b->err = "ba_set_options: value out of range";
    return SIM3OPT_ERR_ARG;
  }
  if (b->ready && (o->linear_solver != b->opt.linear_solver || o->device != b->opt.device))
    b->release();  // the factorisation plan / the device are chosen at the next upload
  b->opt = *o;
  return SIM3OPT_OK;
}

int sim3opt_ba_set_problem(sim3opt_ba* b, int32_t n_cams, const double* cam_qt, int32_t n_points,
                           const double* points, int32_t n_obs, const int32_t* obs_cam,
                           const int32_t* obs_point, const double* obs_uv, double focal, double cx,
                           double cy) {
  if (!b || n_cams < 1 || n_points < 1 || n_obs < 1 || !cam_qt || !points || !obs_cam || !obs_point ||
      !obs_uv || !(focal > 0)) {
    if (b) b->err = "ba_set_problem: bad argument";
    return SIM3OPT_ERR_ARG;
  }
  try {
  for (int32_t o = 0; o < n_obs; ++o)
    if (obs_cam[o] < 0 || obs_cam[o] >= n_cams || obs_point[o] < 0 || obs_point[o] >= n_points) {
      b->err = "ba_set_problem: observation index out of range";  // (the reference asserts, :140-143)
      return SIM3OPT_ERR_ARG;
    }
  for (size_t i = 0; i < 7 * (size_t)n_cams; ++i)
    if (!std::isfinite(cam_qt[i])) { b->err = "ba_set_problem: non-finite camera"; return SIM3OPT_ERR_ARG; }
  for (size_t i = 0; i < 3 * (size_t)n_points; ++i)
    if (!std::isfinite(points[i])) { b->err = "ba_set_problem: non-finite point"; return SIM3OPT_ERR_ARG; }
  for (size_t i = 0; i < 2 * (size_t)n_obs; ++i)
    if (!std::isfinite(obs_uv[i])) { b->err = "ba_set_problem: non-finite observation"; return SIM3OPT_ERR_ARG; }
  b->release();
  b->cams.resize(n_cams);
  for (int32_t c = 0; c < n_cams; ++c) {
    const double* s = cam_qt + 7 * (size_t)c;
    const double nq = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3]);
    if (!(nq > 0)) { b->err = "ba_set_problem: zero quaternion"; return SIM3OPT_ERR_ARG; }
    for (int i = 0; i < 4; ++i) b->cams[c].q[i] = s[i] / nq;
    for (int i = 0; i < 3; ++i) b->cams[c].t[i] = s[4 + i];
    b->cams[c].pad = 0.0;
  }
  b->pts.assign(points, points + 3 * (size_t)n_points);
  b->oc.assign(obs_cam, obs_cam + n_obs);
  b->op.assign(obs_point, obs_point + n_obs);
  b->uv.assign(obs_uv, obs_uv + 2 * (size_t)n_obs);
  b->f = focal; b->cx = cx; b->cy = cy;
  b->cam_fixed.assign(n_cams, 0);
  b->stats.clear();
  return SIM3OPT_OK;
  } catch (...) {  // nothing crosses the C boundary
    b->err = "ba_set_problem: out of host memory"; return SIM3OPT_ERR_ARG;
  }
}

int sim3opt_ba_set_fixed_cameras(sim3opt_ba* b, const uint8_t* fixed) {
  if (!b || !fixed) return SIM3OPT_ERR_ARG;
  try {
  if (b->nc() < 1) { b->err = "ba_set_fixed_cameras: no problem set"; return SIM3OPT_ERR_STATE; }
  b->cam_fixed.assign(fixed, fixed + b->nc());
  for (auto& f : b->cam_fixed) f = f ? 1 : 0;
  b->release();  // the next call re-uploads
  return SIM3OPT_OK;
  } catch (...) {  // nothing crosses the C boundary
    b->err = "ba_set_fixed_cameras: out of host memory"; return SIM3OPT_ERR_ARG;
  }
}

// The BAL file as ba_demo reads it (bal_example.cpp:104-189): "<cams> <points> <observations>", one
// observation per line, 9 numbers per camera (angle-axis, translation, f, k1, k2 -- the last three
// read and ignored: the reference projects with its fixed CameraParameters), 3 per point.
int sim3opt_ba_read_bal(sim3opt_ba* b, const char* path, double focal, double cx, double cy) {
  if (!b || !path) return SIM3OPT_ERR_ARG;
  FILE* f = std::fopen(path, "r");
  if (!f) { b->err = std::string("cannot open ") + path; return SIM3OPT_ERR_IO; }
  int nc = 0, np = 0, no = 0;
  if (std::fscanf(f, "%d %d %d", &nc, &np, &no) != 3 || nc < 1 || np < 1 || no < 1) {
    std::fclose(f);
    b->err = "BAL header";
    return SIM3OPT_ERR_IO;
  }
  {  // counts the file cannot hold (every number takes at least two bytes) are refused before any
     // allocation is sized from them
    const long at = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const long long bytes = std::ftell(f);
    std::fseek(f, at, SEEK_SET);
    const long long need = 2 * (4LL * no + 9LL * nc + 3LL * np);
    if (at < 0 || bytes < need) {
      std::fclose(f);
      b->err = "BAL header: counts exceed the file";
      return SIM3OPT_ERR_IO;
    }
  }
  std::vector<int32_t> oc, op;
  std::vector<double> uv, cams, pts;
  try {
    oc.resize((size_t)no); op.resize((size_t)no);
    uv.resize(2 * (size_t)no); cams.resize(7 * (size_t)nc); pts.resize(3 * (size_t)np);
  } catch (const std::exception&) {
    std::fclose(f);
    b->err = "BAL file: out of host memory";
    return SIM3OPT_ERR_ARG;
  }
  bool ok = true;
  for (size_t o = 0; o < (size_t)no && ok; ++o) ok = std::fscanf(f, "%d %d %lf %lf", &oc[o], &op[o], &uv[2 * o], &uv[2 * o + 1]) == 4;
  for (int c = 0; c < nc && ok; ++c) {
    double v[9];
    for (int j = 0; j < 9 && ok; ++j) ok = std::fscanf(f, "%lf", &v[j]) == 1;
    // ceres AngleAxisToQuaternion (called at bal_example.cpp:168)
    const double th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double k = 0.5, w = 1.0;
    if (th2 > 0.0) {
      const double th = std::sqrt(th2);
      k = std::sin(0.5 * th) / th;
      w = std::cos(0.5 * th);
    }
    double* s = &cams[7 * (size_t)c];
    s[0] = v[0] * k; s[1] = v[1] * k; s[2] = v[2] * k; s[3] = w;
    s[4] = v[3]; s[5] = v[4]; s[6] = v[5];
  }
  for (size_t i = 0; i < pts.size() && ok; ++i) ok = std::fscanf(f, "%lf", &pts[i]) == 1;
  std::fclose(f);
  if (!ok) { b->err = "BAL file truncated or malformed"; return SIM3OPT_ERR_IO; }
  return sim3opt_ba_set_problem(b, nc, cams.data(), np, pts.data(), no, oc.data(), op.data(), uv.data(),
                                focal, cx, cy);
}

int sim3opt_ba_dims(const sim3opt_ba* b, int32_t* n_cams, int32_t* n_points, int32_t* n_obs) {
  if (!b) return SIM3OPT_ERR_ARG;
  if (n_cams) *n_cams = b->nc();
  if (n_points) *n_points = b->np();
  if (n_obs) *n_obs = b->no();
  return SIM3OPT_OK;
}

int sim3opt_ba_chi2(sim3opt_ba* b, double* chi2) {
  if (!b || !chi2) return SIM3OPT_ERR_ARG;
  try {
  if (!b->ready) { int rc = b->initialize(); if (rc) return rc; }
  return b->chi2(chi2);
  } catch (...) {  // nothing crosses the C boundary
    b->err = "ba_chi2: out of host memory or internal error"; return SIM3OPT_ERR_ARG;
  }
}

int sim3opt_ba_optimize(sim3opt_ba* b, int32_t max_iters) {
  if (!b) return 0;
  try {
  b->err.clear();
  if (max_iters < 1 || b->no() < 1) return -1;
  if (!b->ready) { int rc = b->initialize(); if (rc) return 0; }
  const int n = b->optimize(max_iters);
  return n < 0 ? 0 : n;
  } catch (...) {  // nothing crosses the C boundary
    b->err = "ba_optimize: out of host memory or internal error"; return 0;
  }
}

int sim3opt_ba_get_cameras(const sim3opt_ba* b, double* cam_qt) {
  if (!b || !cam_qt) return SIM3OPT_ERR_ARG;
  for (int c = 0; c < b->nc(); ++c) {
    for (int i = 0; i < 4; ++i) cam_qt[7 * (size_t)c + i] = b->cams[c].q[i];
    for (int i = 0; i < 3; ++i) cam_qt[7 * (size_t)c + 4 + i] = b->cams[c].t[i];
  }
  return SIM3OPT_OK;
}

int sim3opt_ba_get_points(const sim3opt_ba* b, double* points) {
  if (!b || !points) return SIM3OPT_ERR_ARG;
  std::memcpy(points, b->pts.data(), sizeof(double) * b->pts.size());
  return SIM3OPT_OK;
}

int32_t sim3opt_ba_num_iterations(const sim3opt_ba* b) { return b ? (int32_t)b->stats.size() : 0; }

int sim3opt_ba_get_stats(const sim3opt_ba* b, int32_t iter, sim3opt_iter_stats* out) {
  if (!b || !out || iter < 0 || iter >= (int32_t)b->stats.size()) return SIM3OPT_ERR_ARG;
  *out = b->stats[iter];
  return SIM3OPT_OK;
}

// "% SE3 optimization result: kf id, tcinw, rc2w(qxyzw)" rows (bal_example.cpp:223-238), 17 digits
int sim3opt_ba_write_poses(const sim3opt_ba* b, const char* path) {
  if (!b || !path) return SIM3OPT_ERR_ARG;
  FILE* f = std::fopen(path, "w");
  if (!f) return SIM3OPT_ERR_IO;
  std::fprintf(f, "%% SE3 optimization result: kf id, tcinw, rc2w(qxyzw):\n");
  for (int c = 0; c < b->nc(); ++c) {
    const double* q = b->cams[c].q;
    const double* t = b->cams[c].t;
    const double qc[4] = {-q[0], -q[1], -q[2], q[3]};
    // tcinw = -R^T t
    const double ux = 2 * (qc[1] * t[2] - qc[2] * t[1]), uy = 2 * (qc[2] * t[0] - qc[0] * t[2]),
                 uz = 2 * (qc[0] * t[1] - qc[1] * t[0]);
    const double r0 = t[0] + qc[3] * ux + (qc[1] * uz - qc[2] * uy);
    const double r1 = t[1] + qc[3] * uy + (qc[2] * ux - qc[0] * uz);
    const double r2 = t[2] + qc[3] * uz + (qc[0] * uy - qc[1] * ux);
    std::fprintf(f, "%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c, -r0, -r1, -r2, qc[0], qc[1], qc[2], qc[3]);
  }
  return std::fclose(f) == 0 ? SIM3OPT_OK : SIM3OPT_ERR_IO;
}

}  // extern "C"
