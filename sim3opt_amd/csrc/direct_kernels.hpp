// direct_kernels.hpp -- device side of the exact sparse block Cholesky (included by engine.hip after
// DevScalars; plan from direct.cpp).  Stands in for LinearSolverEigen::solve (SimplicialLDLT,
// kitti_surf.cpp:553-554) where the step has to be exact: (H + lambda I) = L L^T on the 7x7 block
// pattern, then L y = b and L^T x = y, all in elimination order, result scattered back.
//
// Left-looking by levels of the elimination tree; a level is a contiguous range of columns and of
// stored blocks (direct.hpp).  One workgroup owns one group of the schedule and walks its levels
// with workgroup barriers; within a level
//   phase A  wavefront per stored block (i,j): sum of its H blocks (+ lambda on the diagonal) minus
//            the listed products L(i,k) L(j,k)^T, in list order; the diagonal block also gathers
//            b_j - sum_k L(j,k) y_k (the forward solve rides along)
//   phase B  wavefront per column: 7x7 Cholesky of the diagonal block, its inverse, y_j
//   phase C  wavefront per off-diagonal block: L(i,j) = raw(i,j) L(j,j)^-T
// and the backward solve walks the levels downwards, a wavefront per column.
// Lane l of a wavefront holds entry l49 = l mod 49 of a column-major 7x7 block (lanes 49..63 mirror
// lanes 0..14: every lane issues a valid load); 7x7x7 products go through the LDS crossbar
// (ds_bpermute).  No atomics, fixed summation order: bit-reproducible.
#pragma once
// (included inside namespace sim3opt)

struct LdlArgs {
  const int32_t* perm;
  const int32_t* colptr;
  const int32_t* lrow;
  const int32_t* lcol;
  const int32_t* srcptr;
  const int32_t* src;
  const int32_t* pairptr;
  const int32_t* pa;
  const int32_t* pb;
  const int32_t* gptr;
  const int32_t* lcolp;
  const double* vals;  // block-CSR values of H (column-major 7x7)
  const double* b;     // right-hand side, block rows of H
  double* L;           // nL x 49
  double* Dinv;        // nb x 49: L(j,j)^-1 (lower triangular, column-major)
  double* y;           // 7 nb, elimination order
  double* xp;          // 7 nb, elimination order
  double* x;           // 7 nb, block rows of H (the result)
  double lambda;
  DevScalars* sc;
};

constexpr int LDL_WG_TOP = 1024;  // the top of the tree: one workgroup of 16 wavefronts
constexpr int LDL_WG_SUB = 256;   // bottom subtrees: one workgroup of 4 wavefronts each

// sum over the 7 lanes that share this lane's column index c (lanes 7c .. 7c+6)
__device__ __forceinline__ double ldl_sum_over_r(double v, int c49) {
  double s = 0.0;
#pragma unroll
  for (int rr = 0; rr < 7; ++rr) s += __shfl(v, 7 * c49 + rr);
  return s;
}
// sum over the 7 lanes that share this lane's row index r (lanes r, r+7, ..., r+42)
__device__ __forceinline__ double ldl_sum_over_c(double v, int r49) {
  double s = 0.0;
#pragma unroll
  for (int cc = 0; cc < 7; ++cc) s += __shfl(v, r49 + 7 * cc);
  return s;
}

__device__ __forceinline__ void ldl_phase_a(const LdlArgs& A, int s, int lane, int l49, int r, int c) {
  const int j = A.lcol[s];
  const bool diag = s == A.colptr[j];
  double acc = 0.0;
  for (int k = A.srcptr[s]; k < A.srcptr[s + 1]; ++k) acc += A.vals[(size_t)49 * A.src[k] + l49];
  if (diag && r == c) acc += A.lambda;
  double t = 0.0;
  const int k1 = A.pairptr[s + 1];
  for (int k = A.pairptr[s]; k < k1; ++k) {
    const int sa = A.pa[k], sb = A.pb[k];
    const double a = A.L[(size_t)49 * sa + l49];
    const double bt = sa == sb ? a : A.L[(size_t)49 * sb + l49];
#pragma unroll
    for (int m = 0; m < 7; ++m) acc -= __shfl(a, 7 * m + r) * __shfl(bt, 7 * m + c);
    if (diag) t += a * A.y[(size_t)7 * A.lcol[sa] + c];  // L(j,k)(r,c) y_k(c)
  }
  if (lane < 49) A.L[(size_t)49 * s + lane] = acc;
  if (diag) {
    const double rs = ldl_sum_over_c(t, r);
    if (lane < 7) A.y[(size_t)7 * j + lane] = A.b[(size_t)7 * A.perm[j] + lane] - rs;
  }
}

__device__ __forceinline__ void ldl_phase_b(const LdlArgs& A, int j, int lane, int l49, int r, int c,
                                            double* w /*98 doubles of LDS, this wavefront's*/) {
  const int s0 = A.colptr[j];
  double* wi = w + 49;
  if (lane < 49) w[lane] = A.L[(size_t)49 * s0 + lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {  // 7x7 Cholesky, lower triangle, entry (r, c) at r + 7c
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      double d = w[k + 7 * k];
#pragma unroll
      for (int m = 0; m < k; ++m) d -= w[k + 7 * m] * w[k + 7 * m];
      if (!(d > 0.0) || !(d < DBL_MAX)) { ok = false; d = 1.0; }  // not positive definite (g2o: the solve fails)
      const double lkk = sqrt(d), inv = 1.0 / lkk;
      w[k + 7 * k] = lkk;
#pragma unroll
      for (int rr = k + 1; rr < 7; ++rr) {
        double v = w[rr + 7 * k];
#pragma unroll
        for (int m = 0; m < k; ++m) v -= w[rr + 7 * m] * w[k + 7 * m];
        w[rr + 7 * k] = v * inv;
      }
#pragma unroll
      for (int rr = 0; rr < k; ++rr) w[rr + 7 * k] = 0.0;  // upper triangle
    }
    if (!ok) A.sc->fail = 1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane < 7) {  // column `lane` of the inverse of the lower-triangular factor
    const int cc = lane;
#pragma unroll
    for (int rr = 0; rr < 7; ++rr) {
      double v = rr == cc ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < rr; ++m) v -= w[rr + 7 * m] * (m >= cc ? wi[m + 7 * cc] : 0.0);
      wi[rr + 7 * cc] = rr >= cc ? v / w[rr + 7 * rr] : 0.0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double lf = w[l49], li = wi[l49];
  if (lane < 49) {
    A.L[(size_t)49 * s0 + lane] = lf;
    A.Dinv[(size_t)49 * j + lane] = li;
  }
  // y_j = L(j,j)^-1 (b_j - sum_k L(j,k) y_k)
  const double yr = ldl_sum_over_c(li * A.y[(size_t)7 * j + c], r);
  if (lane < 7) A.y[(size_t)7 * j + lane] = yr;
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void ldl_phase_c(const LdlArgs& A, int s, int lane, int l49, int r, int c) {
  const int j = A.lcol[s];
  const double raw = A.L[(size_t)49 * s + l49];
  const double li = A.Dinv[(size_t)49 * j + l49];
  double xv = 0.0;
#pragma unroll
  for (int m = 0; m < 7; ++m) xv += __shfl(raw, 7 * m + r) * __shfl(li, 7 * m + c);  // raw L(j,j)^-T
  if (lane < 49) A.L[(size_t)49 * s + lane] = xv;
}

// x_j = L(j,j)^-T (y_j - sum_{i > j} L(i,j)^T x_i)
__device__ __forceinline__ void ldl_back(const LdlArgs& A, int j, int lane, int l49, int r, int c) {
  const int s0 = A.colptr[j], s1 = A.colptr[j + 1];
  double t = 0.0;
  for (int s = s0 + 1; s < s1; ++s)
    t += A.L[(size_t)49 * s + l49] * A.xp[(size_t)7 * A.lrow[s] + r];
  const double z = A.y[(size_t)7 * j + c] - ldl_sum_over_r(t, c);  // z(c), the same in lanes (., c)
  const double li = A.Dinv[(size_t)49 * j + l49];
  const double p = __shfl(li, c + 7 * r) * z;  // Linv(c, r) z(c)
  const double xr = ldl_sum_over_c(p, r);
  if (lane < 7) {
    A.xp[(size_t)7 * j + lane] = xr;
    A.x[(size_t)7 * A.perm[j] + lane] = xr;
  }
}

template <bool UP, bool DOWN>
__global__ __launch_bounds__(LDL_WG_TOP) void k_ldl(LdlArgs A, int g0) {
  __shared__ double lds[LDL_WG_TOP / 64][98];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  const int g = g0 + blockIdx.x;
  const int lv0 = A.gptr[g], lv1 = A.gptr[g + 1];
  if (UP) {
    for (int l = lv0; l < lv1; ++l) {
      const int c0 = A.lcolp[l], c1 = A.lcolp[l + 1];
      const int sb = A.colptr[c0], se = A.colptr[c1];
      for (int s = sb + wave; s < se; s += nw) ldl_phase_a(A, s, lane, l49, r, c);
      __syncthreads();
      for (int j = c0 + wave; j < c1; j += nw) ldl_phase_b(A, j, lane, l49, r, c, lds[wave]);
      __syncthreads();
      for (int s = sb + wave; s < se; s += nw)
        if (s != A.colptr[A.lcol[s]]) ldl_phase_c(A, s, lane, l49, r, c);
      __syncthreads();
    }
  }
  if (DOWN) {
    for (int l = lv1 - 1; l >= lv0; --l) {
      const int c0 = A.lcolp[l], c1 = A.lcolp[l + 1];
      for (int j = c0 + wave; j < c1; j += nw) ldl_back(A, j, lane, l49, r, c);
      __syncthreads();
    }
  }
}
