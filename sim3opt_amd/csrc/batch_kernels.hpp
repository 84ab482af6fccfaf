// batch_kernels.hpp -- the PCG loop for K right-hand sides at once (included by engine_batch.hip only, inside
// namespace sim3opt, after spmv_kernel.hpp): the rejected trials of one LM iteration solve
// (H + lambda_k I) x_k = b for a KNOWN sequence lambda_k (g2o's OptimizationAlgorithmLevenberg: lambda *= nu,
// nu *= 2 after every rejection), so after the first rejection the next K systems are solved together --
// ONE pass over the matrix blocks for K vectors, K vectors per coarse launch -- and the trials are then
// evaluated in g2o's order.  What LinearSolverEigen does K times in a row (kitti_surf.cpp:553-554, 675).
// Every kernel here performs, per system, the operations of its one-system counterpart in the same order:
// the K solutions are bit for bit those of K sequential solves (asserted in tests/test_gpu_parity.py).
// Vectors of system s live at base + s * stride (BatchStrides, spmv_kernel.hpp, where the K-system SpMV lives:
// the one-system kernel is its K = 1 instantiation).
#pragma once

// ------------------------------------------------------------------------------------------
// PCG vector kernels for K systems (k_pcg_init / k_pcg_step / k_final_sum2 per system; b is shared)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(WG) void k_final_sum2_k(const double* __restrict__ pa, const double* __restrict__ pb,
                                                     int n, int pstride, DevScalars* __restrict__ sc) {
  __shared__ double sh[4];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    const double a = sum_partials(pa + (size_t)s * pstride, n, sh);
    const double b = sum_partials(pb + (size_t)s * pstride, n, sh);
    if (threadIdx.x == 0) {
      sc[s].tmp_pq = a;
      sc[s].tmp_rz = b;
    }
  }
}

template <int K>
__global__ __launch_bounds__(WG) void k_pcg_init_k(int r0, int r1, const double* __restrict__ b,
                                                   const double* __restrict__ Minv, double* __restrict__ x,
                                                   double* __restrict__ r, double* __restrict__ z,
                                                   double* __restrict__ p, double* __restrict__ sv,
                                                   BatchStrides bs) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = r0 + (blockIdx.x * 4 + wave) * 9; row0 < r1; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < r1;
    const size_t j = (size_t)7 * row + rr;
    const double rv = act ? b[j] : 0.0;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const size_t o = (size_t)s * bs.vec;
      if (act) {
        x[o + j] = 0.0;
        r[o + j] = rv;
        p[o + j] = 0.0;
        sv[o + j] = 0.0;
      }
      double zv = 0.0;
#pragma unroll
      for (int cc = 0; cc < 7; ++cc) {
        const double rc = __shfl(rv, base + cc);
        if (act) zv += Minv[(size_t)s * bs.minv + (size_t)49 * row + 7 * rr + cc] * rc;
      }
      if (act) z[o + j] = zv;
    }
  }
}

// k_pcg_step for K systems: every system has its own scalars (sc[s]); a finished system is left alone.
// it >= 0: the launch number (0 = first iteration of every system); the systems run in lock-step, a system
// that converges earlier just stops being updated.
template <int K>
__global__ __launch_bounds__(WG) void k_pcg_step_k(int r0, int r1, int par, int it,
                                                   const double* __restrict__ Minv, const double* zin,
                                                   double* zout, const double* __restrict__ w,
                                                   double* __restrict__ p, double* __restrict__ sv,
                                                   double* __restrict__ x, double* __restrict__ r,
                                                   DevScalars* sc, BatchStrides bs) {
  double alpha[K], beta[K];
  bool live[K];
  const bool commit = blockIdx.x == 0 && threadIdx.x == 0;
  const bool first = it == 0;
  bool any = false;
#pragma unroll
  for (int s = 0; s < K; ++s) {
    live[s] = false;
    alpha[s] = beta[s] = 0.0;
    if (sc[s].done) continue;
    const double delta = sc[s].tmp_pq, gamma = sc[s].tmp_rz;
    const double gamma0 = first ? gamma : sc[s].rz0;
    if (!(gamma == gamma) || gamma < 0.0 || gamma <= sc[s].tol2 * gamma0 || (first && gamma == 0.0)) {
      if (commit) {
        if (!(gamma == gamma) || gamma < 0.0) sc[s].fail = 1;
        if (first) sc[s].rz0 = gamma;
        sc[s].rz[par ^ 1] = gamma;
        sc[s].gam_last = gamma;
        sc[s].done = 1;
      }
      continue;
    }
    beta[s] = first ? 0.0 : gamma / sc[s].rz[par];
    const double denom = first ? delta : delta - beta[s] * gamma / sc[s].alpha[par];
    if (!(denom > 0.0) || !(denom < DBL_MAX)) {
      if (commit) {
        sc[s].fail = 1;
        sc[s].done = 1;
      }
      continue;
    }
    alpha[s] = gamma / denom;
    live[s] = true;
    any = true;
  }
  if (!any) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = r0 + (blockIdx.x * 4 + wave) * 9; row0 < r1; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < r1;
    const size_t j = (size_t)7 * row + rr;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      if (!live[s]) continue;  // (uniform over the grid)
      const size_t o = (size_t)s * bs.vec;
      double rv = 0.0;
      if (act) {
        const double pn = zin[o + j] + beta[s] * p[o + j];
        const double sn = w[o + j] + beta[s] * sv[o + j];
        p[o + j] = pn;
        sv[o + j] = sn;
        x[o + j] += alpha[s] * pn;
        rv = r[o + j] - alpha[s] * sn;
        r[o + j] = rv;
      }
      double zv = 0.0;
#pragma unroll
      for (int cc = 0; cc < 7; ++cc) {
        const double rc = __shfl(rv, base + cc);
        if (act) zv += Minv[(size_t)s * bs.minv + (size_t)49 * row + 7 * rr + cc] * rc;
      }
      if (act) zout[o + j] = zv;
    }
  }
  if (commit) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      if (!live[s]) continue;
      const double gamma = sc[s].tmp_rz;
      if (first) sc[s].rz0 = gamma;
      sc[s].rz[par ^ 1] = gamma;
      sc[s].gam_last = gamma;
      sc[s].alpha[par ^ 1] = alpha[s];
      const int itn = it + 1;
      sc[s].iter = itn;
      if (itn >= sc[s].max_iter) sc[s].stop = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// multigrid transfer kernels for K systems (k_amg_restrict0 / k_amg_restrict / k_amg_prolong<true> /
// k_amg_dense_apply per system; P and the aggregation are shared)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(WG) void k_amg_restrict0_k(int nc, const int32_t* __restrict__ mptr,
                                                        const int32_t* __restrict__ mem,
                                                        const double* __restrict__ P,
                                                        const double* __restrict__ t_f, double* __restrict__ r_c,
                                                        const double* __restrict__ Minv_c, double* __restrict__ x_c,
                                                        const DevScalars* __restrict__ sc, int64_t vs_f, int64_t vs_c,
                                                        int64_t ms_c) {
  if (sc) {
    bool all_done = true;
#pragma unroll
    for (int s = 0; s < K; ++s) all_done = all_done && sc[s].done;
    if (all_done) return;
  }
  const int lane = threadIdx.x & 63;
  const int a = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (a >= nc) return;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int m = l49 % 7, c = l49 / 7;
  double acc[K];
#pragma unroll
  for (int s = 0; s < K; ++s) acc[s] = 0.0;
  const int e0 = mptr[a], e1 = mptr[a + 1];
  for (int e = e0; e < e1; ++e) {
    const int i = mem[e];
    const double pv = P[(size_t)49 * i + l49];
#pragma unroll
    for (int s = 0; s < K; ++s) acc[s] += pv * t_f[(size_t)s * vs_f + (size_t)7 * i + m];
  }
#pragma unroll
  for (int s = 0; s < K; ++s) {
    double rc = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) rc += __shfl(acc[s], 7 * c + q);
    if (lane < 49 && m == 0) r_c[(size_t)s * vs_c + (size_t)7 * a + c] = rc;
    if (Minv_c) {
      const double pr = Minv_c[(size_t)s * ms_c + (size_t)49 * a + l49] * rc;
      double xv = pr;
#pragma unroll
      for (int q = 1; q < 7; ++q) xv += __shfl(pr, m + 7 * ((c + q) % 7));
      if (lane < 7) x_c[(size_t)s * vs_c + (size_t)7 * a + lane] = xv;
    }
  }
}

template <int K>
__global__ __launch_bounds__(WG) void k_amg_restrict_k(int nc, const int32_t* __restrict__ mptr,
                                                       const int32_t* __restrict__ mem,
                                                       const double* __restrict__ t_f, double* __restrict__ r_c,
                                                       const double* __restrict__ Minv_c, double* __restrict__ x_c,
                                                       int64_t vs_f, int64_t vs_c, int64_t ms_c) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  if (gridDim.y > 1) {  // every slice of the grid takes K of the systems
    const size_t s0 = (size_t)blockIdx.y * K;
    t_f += s0 * vs_f;
    r_c += s0 * vs_c;
    x_c += s0 * vs_c;
    if (Minv_c) Minv_c += s0 * ms_c;
  }
  for (int a0 = (blockIdx.x * 4 + wave) * 9; a0 < nc; a0 += gridDim.x * 36) {
    const int a = a0 + sub;
    const bool act = lane < 63 && a < nc;
    const int e0 = act ? mptr[a] : 0, e1 = act ? mptr[a + 1] : 0;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const double* tf = t_f + (size_t)s * vs_f;
      double acc = 0.0;
      int e = e0;
      for (; e + 3 < e1; e += 4) {  // four members in flight (the one-system kernel's order of summation)
        const int i0 = mem[e], i1 = mem[e + 1], i2 = mem[e + 2], i3 = mem[e + 3];
        const double v0 = tf[(size_t)7 * i0 + rr], v1 = tf[(size_t)7 * i1 + rr];
        const double v2 = tf[(size_t)7 * i2 + rr], v3 = tf[(size_t)7 * i3 + rr];
        acc += (v0 + v1) + (v2 + v3);
      }
      for (; e < e1; ++e) acc += tf[(size_t)7 * mem[e] + rr];
      if (act) r_c[(size_t)s * vs_c + (size_t)7 * a + rr] = acc;
      if (Minv_c) {
        double xv = 0.0;
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) {
          const double rc = __shfl(acc, base + cc);
          if (act) xv += Minv_c[(size_t)s * ms_c + (size_t)49 * a + 7 * rr + cc] * rc;
        }
        if (act) x_c[(size_t)s * vs_c + (size_t)7 * a + rr] = xv;
      }
    }
  }
}

// x_out[i] = x_in[i] + scale P_i x_c[agg[i]] (level 0), K systems
template <int K>
__global__ __launch_bounds__(WG) void k_amg_prolong0_k(int nb, const int32_t* __restrict__ agg,
                                                       const double* __restrict__ P, const double* __restrict__ x_c,
                                                       const double* x_in, double* x_out,
                                                       const DevScalars* __restrict__ sc, double scale, int64_t vs_f,
                                                       int64_t vs_c) {
  if (sc) {
    bool all_done = true;
#pragma unroll
    for (int s = 0; s < K; ++s) all_done = all_done && sc[s].done;
    if (all_done) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = (blockIdx.x * 4 + wave) * 9; row0 < nb; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < nb;
    const int ag = act ? agg[row] : 0;
    double pm[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) pm[m] = act ? P[(size_t)49 * row + rr + 7 * m] : 0.0;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const double xc = act ? x_c[(size_t)s * vs_c + (size_t)7 * ag + rr] : 0.0;
      double add = 0.0;
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        const double xm = __shfl(xc, base + m);
        if (act) add += pm[m] * xm;
      }
      if (act) x_out[(size_t)s * vs_f + (size_t)7 * row + rr] = x_in[(size_t)s * vs_f + (size_t)7 * row + rr] + scale * add;
    }
  }
}

// x_s = Ainv_s r_s on the coarsest level (every system has its own dense inverse: its damping is inside)
template <int K>
__global__ __launch_bounds__(WG) void k_amg_dense_apply_k(int n, const double* __restrict__ Ainv,
                                                          const double* __restrict__ r, double* __restrict__ x,
                                                          int64_t as, int64_t vs) {
  const int lane = threadIdx.x & 63;
  if (gridDim.y > 1) {  // every slice of the grid takes K of the systems
    const size_t s0 = (size_t)blockIdx.y * K;
    Ainv += s0 * as;
    r += s0 * vs;
    x += s0 * vs;
  }
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const double* row = Ainv + (size_t)s * as + (size_t)i * n;
      const double* rs = r + (size_t)s * vs;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int j = lane;
      for (; j + 192 < n; j += 256) {
        const double m0 = row[j], m1 = row[j + 64], m2 = row[j + 128], m3 = row[j + 192];
        a0 += m0 * rs[j]; a1 += m1 * rs[j + 64]; a2 += m2 * rs[j + 128]; a3 += m3 * rs[j + 192];
      }
      for (; j < n; j += 64) a0 += row[j] * rs[j];
      double acc = (a0 + a1) + (a2 + a3);
      acc = wave_sum(acc);
      if (lane == 0) x[(size_t)s * vs + i] = acc;
    }
  }
}
