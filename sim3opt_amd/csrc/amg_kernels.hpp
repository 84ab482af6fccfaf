// amg_kernels.hpp -- device side of the aggregation-multigrid preconditioner (included by
// engine.hip after DevScalars / load_sim3; structure from amg.cpp).
//
// Level 0 is the LM system itself; level l+1 = P_l^T (H_l) P_l (Galerkin) with
//   P_0 = block rows Ad(S_v) of the aggregate's members (the near-kernel of a pose-graph Hessian is
//         dx_v = Ad(S_v) g: a global right-multiplication leaves every residual unchanged),
//   P_l = piecewise constant (identity blocks) for l >= 1: coarse variables live in the world frame.
// The damping is carried separately: P^T (H + lambda I) P = P^T H P + lambda W, W = P^T P block
// diagonal, so the products are formed once per linearisation and a trial only refreshes the
// diagonal blocks.  Everything is fixed-order (no atomics): results are bit-reproducible.
// Blocks, P and W are column-major 7x7 (entry (r, c) at r + 7c); lane l of a wavefront handles
// entry l49 = l mod 49, so lanes 49..63 mirror lanes 0..14 (no exec-masked loads).
#pragma once
// (included inside namespace sim3opt)

// P_i = Ad(S_v), tangent order [omega, upsilon, sigma]:
//   [ R      0    0 ]
//   [ [t]x R s R  -t ]        S exp(x) S^-1 = exp(Ad_S x), S = (R, t, s)   (sim3_rv.h:199-220 algebra)
//   [ 0      0    1 ]
__global__ __launch_bounds__(WG) void k_amg_adjoint(int nb, const int32_t* __restrict__ row2v,
                                                    const Sim3* __restrict__ states,
                                                    double* __restrict__ P) {
  for (int i = blockIdx.x * WG + threadIdx.x; i < nb; i += gridDim.x * WG) {
  const Sim3 S = load_sim3(states + row2v[i]);
  double R[9];
  sim3::R_from_quat(S.q, R);
  double A[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) A[k] = 0.0;
  const double t0 = S.t[0], t1 = S.t[1], t2 = S.t[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double v0 = R[c], v1 = R[3 + c], v2 = R[6 + c];  // column c of R
    A[0 + 7 * c] = v0;
    A[1 + 7 * c] = v1;
    A[2 + 7 * c] = v2;
    A[3 + 7 * c] = t1 * v2 - t2 * v1;  // t x column
    A[4 + 7 * c] = t2 * v0 - t0 * v2;
    A[5 + 7 * c] = t0 * v1 - t1 * v0;
    A[3 + 7 * (3 + c)] = S.s * v0;
    A[4 + 7 * (3 + c)] = S.s * v1;
    A[5 + 7 * (3 + c)] = S.s * v2;
  }
  A[3 + 42] = -t0;
  A[4 + 42] = -t1;
  A[5 + 42] = -t2;
  A[6 + 42] = 1.0;
  double* dst = P + (size_t)49 * i;
#pragma unroll
  for (int k = 0; k < 49; ++k) dst[k] = A[k];
  }
}

// Coarse block cb = sum over its fine blocks k (row i, column j) of P_i^T A_k P_j, in list order.
// One wavefront per coarse block; the two 7x7x7 products run on the LDS crossbar (ds_bpermute).
template <bool HASP>
__global__ __launch_bounds__(WG) void k_amg_galerkin(int ncb, const int32_t* __restrict__ gptr,
                                                     const int32_t* __restrict__ gblk,
                                                     const int32_t* __restrict__ grow,
                                                     const int32_t* __restrict__ colidx_f,
                                                     const double* __restrict__ vals_f,
                                                     const double* __restrict__ P,
                                                     double* __restrict__ vals_c) {
  const int lane = threadIdx.x & 63;
  const int cb = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (cb >= ncb) return;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  double acc = 0.0;
  const int e0 = gptr[cb], e1 = gptr[cb + 1];
  for (int e = e0; e < e1; ++e) {
    const int k = gblk[e];
    const double a = vals_f[(size_t)49 * k + l49];
    if (HASP) {
      const int i = grow[e], j = colidx_f[k];
      const double pi = P[(size_t)49 * i + l49];
      const double pj = P[(size_t)49 * j + l49];
      double t = 0.0;  // T = A P_j
#pragma unroll
      for (int m = 0; m < 7; ++m) t += __shfl(a, r + 7 * m) * __shfl(pj, m + 7 * c);
      double o = 0.0;  // P_i^T T
#pragma unroll
      for (int m = 0; m < 7; ++m) o += __shfl(pi, m + 7 * r) * __shfl(t, m + 7 * c);
      acc += o;
    } else {
      acc += a;
    }
  }
  if (lane < 49) vals_c[(size_t)49 * cb + lane] = acc;
}

// keeps the undamped Galerkin diagonal blocks (a trial overwrites the ones inside vals)
__global__ __launch_bounds__(WG) void k_amg_copydiag(int nb, const int32_t* __restrict__ rowptr,
                                                     const double* __restrict__ vals,
                                                     double* __restrict__ diagH) {
  for (int idx = blockIdx.x * WG + threadIdx.x; idx < 49 * nb; idx += gridDim.x * WG)
    diagH[idx] = vals[(size_t)49 * rowptr[idx / 49] + idx % 49];
}

// W_c[a] = sum over members i of P_i^T P_i (FIRST) or of W_f[i]; one wavefront per aggregate
template <bool FIRST>
__global__ __launch_bounds__(WG) void k_amg_wsum(int nc, const int32_t* __restrict__ mptr,
                                                 const int32_t* __restrict__ mem,
                                                 const double* __restrict__ src,
                                                 double* __restrict__ Wc) {
  const int lane = threadIdx.x & 63;
  const int a = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (a >= nc) return;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  double acc = 0.0;
  for (int e = mptr[a]; e < mptr[a + 1]; ++e) {
    const double v = src[(size_t)49 * mem[e] + l49];
    if (FIRST) {
      double o = 0.0;
#pragma unroll
      for (int m = 0; m < 7; ++m) o += __shfl(v, m + 7 * r) * __shfl(v, m + 7 * c);
      acc += o;
    } else {
      acc += v;
    }
  }
  if (lane < 49) Wc[(size_t)49 * a + lane] = acc;
}

// r_c[a] = sum over members i of P_i^T t_f[i]; then x_c[a] = Minv_c[a] r_c[a] (first smoothing step
// of the coarse level from a zero guess).  63 lanes = 9 aggregates x 7 entries.
template <bool HASP>
__global__ __launch_bounds__(WG) void k_amg_restrict(int nc, const int32_t* __restrict__ mptr,
                                                     const int32_t* __restrict__ mem,
                                                     const double* __restrict__ P,
                                                     const double* __restrict__ t_f,
                                                     double* __restrict__ r_c,
                                                     const double* __restrict__ Minv_c,
                                                     double* __restrict__ x_c,
                                                     const DevScalars* __restrict__ sc, int row_lo,
                                                     int row_hi) {
  if (sc && sc->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int a0 = (blockIdx.x * 4 + wave) * 9; a0 < nc; a0 += gridDim.x * 36) {
    const int a = a0 + sub;
    const bool act = lane < 63 && a < nc;
    const int e0 = act ? mptr[a] : 0, e1 = act ? mptr[a + 1] : 0;
    double acc = 0.0;
    for (int e = e0; __any(e < e1); ++e) {
      const int i = e < e1 ? mem[e] : 0;
      // multi-GPU: a rank sums the members it owns; the partial sums are all-reduced afterwards
      const bool on = e < e1 && i >= row_lo && i < row_hi;
      const double tv = on ? t_f[(size_t)7 * i + rr] : 0.0;
      if (HASP) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
          const double tm = __shfl(tv, base + m);
          if (on) acc += P[(size_t)49 * i + m + 7 * rr] * tm;
        }
      } else {
        acc += tv;
      }
    }
    if (act) r_c[(size_t)7 * a + rr] = acc;
    if (Minv_c) {
      double xv = 0.0;
#pragma unroll
      for (int cc = 0; cc < 7; ++cc) {
        const double rc = __shfl(acc, base + cc);
        if (act) xv += Minv_c[(size_t)49 * a + 7 * rr + cc] * rc;
      }
      if (act) x_c[(size_t)7 * a + rr] = xv;
    }
  }
}

// x = Minv r, 63 lanes = 9 block rows x 7 (multi-GPU: after the all-reduce of the restricted residual)
__global__ __launch_bounds__(WG) void k_amg_bjapply(int nb, const double* __restrict__ Minv,
                                                    const double* __restrict__ r,
                                                    double* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = (blockIdx.x * 4 + wave) * 9; row0 < nb; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < nb;
    const double rv = act ? r[(size_t)7 * row + rr] : 0.0;
    double xv = 0.0;
#pragma unroll
    for (int cc = 0; cc < 7; ++cc) {
      const double rc = __shfl(rv, base + cc);
      if (act) xv += Minv[(size_t)49 * row + 7 * rr + cc] * rc;
    }
    if (act) x[(size_t)7 * row + rr] = xv;
  }
}

// x_out[i] = x_in[i] + P_i x_c[agg[i]]   (x_out may be x_in)
template <bool HASP>
__global__ __launch_bounds__(WG) void k_amg_prolong(int nb, const int32_t* __restrict__ agg,
                                                    const double* __restrict__ P,
                                                    const double* __restrict__ x_c,
                                                    const double* x_in, double* x_out,
                                                    const DevScalars* __restrict__ sc) {
  if (sc && sc->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = (blockIdx.x * 4 + wave) * 9; row0 < nb; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < nb;
    const double xc = act ? x_c[(size_t)7 * agg[row] + rr] : 0.0;
    double add = xc;
    if (HASP) {
      add = 0.0;
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        const double xm = __shfl(xc, base + m);
        if (act) add += P[(size_t)49 * row + rr + 7 * m] * xm;
      }
    }
    if (act) x_out[(size_t)7 * row + rr] = x_in[(size_t)7 * row + rr] + add;
  }
}

// Coarsest level: dense copy of its block-CSR matrix (unique columns per row, damping already in
// the diagonal blocks) inverted in place by Gauss-Jordan without pivoting (SPD: positive pivots).
// One workgroup; n = 7 nb <= 448.  Pivot row and column go through LDS: two barriers per pivot.
constexpr int AMG_DENSE_WG = 1024;
constexpr int AMG_DENSE_MAX_N = 448;
__global__ __launch_bounds__(AMG_DENSE_WG) void k_amg_dense_invert(int nb,
                                                                   const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ colidx,
                                                                   const double* __restrict__ vals,
                                                                   double* __restrict__ Ainv,
                                                                   DevScalars* sc) {
  __shared__ double rowk[AMG_DENSE_MAX_N], colk[AMG_DENSE_MAX_N];
  __shared__ int bad;
  const int n = 7 * nb, tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int idx = tid; idx < n * n; idx += AMG_DENSE_WG) Ainv[idx] = 0.0;
  __syncthreads();
  for (int i = 0; i < nb; ++i) {
    const int k0 = rowptr[i], cnt = (rowptr[i + 1] - k0) * 49;
    for (int t = tid; t < cnt; t += AMG_DENSE_WG) {
      const int k = k0 + t / 49, e = t % 49;
      Ainv[(size_t)(7 * i + e % 7) * n + 7 * colidx[k] + e / 7] = vals[(size_t)49 * k + e];
    }
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const double piv = Ainv[(size_t)k * n + k];
    if (tid == 0 && !(piv > 0.0)) bad = 1;
    const double d = 1.0 / piv;
    for (int j = tid; j < n; j += AMG_DENSE_WG) {
      rowk[j] = Ainv[(size_t)k * n + j] * d;
      colk[j] = Ainv[(size_t)j * n + k];
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += AMG_DENSE_WG) {
      const int i = idx / n, j = idx - i * n;
      double v;
      if (i == k) v = j == k ? d : rowk[j];
      else if (j == k) v = -colk[i] * d;
      else v = Ainv[idx] - colk[i] * rowk[j];
      Ainv[idx] = v;
    }
    __syncthreads();
  }
  if (tid == 0 && bad) sc->fail = 1;
}

// x = Ainv r on the coarsest level (Ainv symmetric: column reads are coalesced)
__global__ __launch_bounds__(512) void k_amg_dense_apply(int n, const double* __restrict__ Ainv,
                                                         const double* __restrict__ r,
                                                         double* __restrict__ x,
                                                         const DevScalars* __restrict__ sc) {
  __shared__ double rs[AMG_DENSE_MAX_N];
  if (sc && sc->done) return;
  const int tid = threadIdx.x;
  if (tid < n) rs[tid] = r[tid];
  __syncthreads();
  if (tid >= n) return;
  double acc = 0.0;
  for (int j = 0; j < n; ++j) acc += Ainv[(size_t)j * n + tid] * rs[j];
  x[tid] = acc;
}

