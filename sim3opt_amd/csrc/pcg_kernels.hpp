// pcg_kernels.hpp -- device side of the linear solve, non-template part (included by engine_pcg.hip ONLY,
// inside namespace sim3opt): the single-reduction PCG step, the block-Jacobi inverses, the chain-segment
// preconditioner, the halo exchange of the row-partitioned path, fixed-order two-value sums and the HBM read
// calibration kernels.  The block-CSR SpMV itself is a template: spmv_kernel.hpp.  Stands in for
// LinearSolverEigen::solve (kitti_surf.cpp:553-554) on graphs where a factorisation is not cheap;
// SURVEY.md 8(a) row a9.
#pragma once
// ------------------------------------------------------------------------------------------
// PCG kernels.  Vector kernels map 63 lanes of a wavefront onto 9 block rows x 7 so a block
// row's 7 entries sit in one wavefront (z = Minv r by shuffles) and addresses stay contiguous.
// ------------------------------------------------------------------------------------------
// two sums in one launch (multi-GPU PCG: [w.z, r.z] land in adjacent doubles for one all-reduce)
__global__ __launch_bounds__(WG) void k_final_sum2(const double* __restrict__ pa,
                                                   const double* __restrict__ pb, int n,
                                                   double* __restrict__ out2) {
  __shared__ double sh[4];
  const double a = sum_partials(pa, n, sh);
  const double b = sum_partials(pb, n, sh);
  if (threadIdx.x == 0) {
    out2[0] = a;
    out2[1] = b;
  }
}

// multi-GPU: the breakdown flag is rank-local (a non-SPD block on one rank's rows); the ranks agree
// on it through a max all-reduce of tmp_pq so that they keep taking the same branches
__global__ void k_fail_to_double(DevScalars* sc) { sc->tmp_pq = sc->fail ? 1.0 : 0.0; }
__global__ void k_double_to_fail(DevScalars* sc) { if (sc->tmp_pq > 0.0) sc->fail = 1; }

// ------------------------------------------------------------------------------------------
// block-Jacobi preconditioner / smoother: Minv = omega (D + lambda W)^-1, one lane per block row
// (Gauss-Jordan without pivoting; positive pivots <=> SPD block).  Level 0 of the system:
// D = the row's diagonal block, W = I.  Coarse multigrid levels (diagH, W given): D = the undamped
// Galerkin diagonal block, W = P^T P; the damped block is also stored back into vals.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_jacobi(int r0, int r1, const int32_t* __restrict__ rowptr,
                                               double* __restrict__ vals, double lambda,
                                               double* __restrict__ Minv, DevScalars* sc,
                                               double omega, const double* __restrict__ diagH,
                                               const double* __restrict__ W,
                                               float* __restrict__ vals32 = nullptr,
                                               double* __restrict__ diag64_out = nullptr,
                                               float* __restrict__ diag32_out = nullptr) {
  const int row = r0 + blockIdx.x * WG + threadIdx.x;
  if (row >= r1) return;
  double a[7][7];
  double* blk = vals + (size_t)49 * rowptr[row];
  const int64_t kd = rowptr[row];  // the row's diagonal block
  const double* src = diagH ? diagH + (size_t)49 * row : blk;
#pragma unroll
  for (int c = 0; c < 7; ++c)
#pragma unroll
    for (int r = 0; r < 7; ++r) a[r][c] = src[7 * c + r];
  if (W) {
    const double* w = W + (size_t)49 * row;
#pragma unroll
    for (int c = 0; c < 7; ++c)
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        a[r][c] += lambda * w[7 * c + r];
        if (diag64_out || diag32_out) {
          // (one of several systems solved together, batch_kernels.hpp: the damped block goes to the system's
          // own arrays, [row][49] column-major; the shared level arrays stay as they are)
          if (diag64_out) diag64_out[(size_t)49 * row + 7 * c + r] = a[r][c];
          if (diag32_out) diag32_out[(size_t)49 * row + 7 * c + r] = (float)a[r][c];
        } else {
          blk[7 * c + r] = a[r][c];
          if (vals32) vals32[f32_pair_index(kd, 7 * c + r)] = (float)a[r][c];
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i][i] += lambda;
  }
  bool spd = true;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if (!(a[k][k] > 0.0)) spd = false;
    const double d = 1.0 / a[k][k];
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j != k) a[k][j] *= d;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i != k) {
        const double f = a[i][k];
#pragma unroll
        for (int j = 0; j < 7; ++j)
          if (j != k) a[i][j] -= f * a[k][j];
        a[i][k] = -f * d;
      }
    a[k][k] = d;
  }
  if (!spd) sc->fail = 1;
  double* dst = Minv + (size_t)49 * row;  // row-major
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int c = 0; c < 7; ++c) dst[7 * r + c] = omega * a[r][c];
}

// x = 0, r = b, z = Minv b (block-Jacobi; the chain preconditioner runs separately), p = s = 0
__global__ __launch_bounds__(WG) void k_pcg_init(int r0, int r1, const double* __restrict__ b,
                                                 const double* __restrict__ Minv,
                                                 double* __restrict__ x, double* __restrict__ r,
                                                 double* __restrict__ z, double* __restrict__ p,
                                                 double* __restrict__ sv) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = r0 + (blockIdx.x * 4 + wave) * 9; row0 < r1; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < r1;
    const size_t j = (size_t)7 * row + rr;
    const double rv = act ? b[j] : 0.0;
    if (act) {
      x[j] = 0.0;
      r[j] = rv;
      p[j] = 0.0;
      sv[j] = 0.0;
    }
    if (Minv) {
      double zv = 0.0;
#pragma unroll
      for (int cc = 0; cc < 7; ++cc) {
        const double rc = __shfl(rv, base + cc);
        if (act) zv += Minv[(size_t)49 * row + 7 * rr + cc] * rc;
      }
      if (act) z[j] = zv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Chain-segment preconditioner (option `preconditioner`): M = the block-tridiagonal part of
// H + lambda I inside segments of `seg` consecutive block rows (diagonal blocks plus the blocks
// between rows i and i-1, i.e. the odometry chain of kitti_surf.cpp:649-670), factored exactly:
//   S_i = D_i + lambda I - G_i L_i^T,  G_i = L_i S_{i-1}^-1,  L_i = H(i, i-1)   (block LDL^T)
// For chain-like graphs (KITTI-00: 770 rows, 1..118 loops) the preconditioned operator is the
// identity plus a low-rank term (14 per loop or cut link), so PCG needs tens of iterations where
// block-Jacobi needs 1e4.  It does not help loop-dominated graphs (measured).
// ------------------------------------------------------------------------------------------
// factorisation: one lane per segment (once per LM trial; sequential by nature)
__global__ __launch_bounds__(64) void k_chain_factor(int r0, int r1, int seg,
                                                     const int32_t* __restrict__ rowptr,
                                                     const double* __restrict__ vals,
                                                     const int32_t* __restrict__ sub_first,
                                                     const int32_t* __restrict__ sub_cnt,
                                                     double lambda, double* __restrict__ Sinv,
                                                     double* __restrict__ Gm, DevScalars* sc) {
  const int sidx = blockIdx.x * 64 + threadIdx.x;
  const long long start = (long long)r0 + (long long)sidx * seg;
  if (start >= r1) return;
  const int row_end = (int)(start + seg < r1 ? start + seg : r1);
  double P[7][7];  // S_{i-1}^-1
  bool spd = true;
  for (int i = (int)start; i < row_end; ++i) {
    double a[7][7], G[7][7];
    const double* d = vals + (size_t)49 * rowptr[i];
    for (int c = 0; c < 7; ++c)
      for (int r = 0; r < 7; ++r) a[r][c] = d[7 * c + r];
    for (int k = 0; k < 7; ++k) a[k][k] += lambda;
    const int nsub = i > start ? sub_cnt[i] : 0;
    if (nsub > 0) {
      double Lm[7][7];
      for (int r = 0; r < 7; ++r)
        for (int c = 0; c < 7; ++c) Lm[r][c] = 0.0;
      for (int t = 0; t < nsub; ++t) {  // parallel edges between i and i-1 keep separate blocks
        const double* l = vals + (size_t)49 * (sub_first[i] + t);
        for (int c = 0; c < 7; ++c)
          for (int r = 0; r < 7; ++r) Lm[r][c] += l[7 * c + r];
      }
      for (int r = 0; r < 7; ++r)
        for (int c = 0; c < 7; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 7; ++k) acc += Lm[r][k] * P[k][c];
          G[r][c] = acc;
        }
      for (int r = 0; r < 7; ++r)
        for (int c = 0; c < 7; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 7; ++k) acc += G[r][k] * Lm[c][k];
          a[r][c] -= acc;
        }
    } else {
      for (int r = 0; r < 7; ++r)
        for (int c = 0; c < 7; ++c) G[r][c] = 0.0;
    }
    for (int k = 0; k < 7; ++k) {  // Gauss-Jordan, positive pivots <=> SPD
      if (!(a[k][k] > 0.0)) spd = false;
      const double dd = 1.0 / a[k][k];
      for (int j = 0; j < 7; ++j)
        if (j != k) a[k][j] *= dd;
      for (int r = 0; r < 7; ++r)
        if (r != k) {
          const double f = a[r][k];
          for (int j = 0; j < 7; ++j)
            if (j != k) a[r][j] -= f * a[k][j];
          a[r][k] = -f * dd;
        }
      a[k][k] = dd;
    }
    double* so = Sinv + (size_t)49 * i;
    double* go = Gm + (size_t)49 * i;
    for (int r = 0; r < 7; ++r)
      for (int c = 0; c < 7; ++c) {
        so[7 * r + c] = a[r][c];
        go[7 * r + c] = G[r][c];
        P[r][c] = a[r][c];
      }
  }
  if (!spd) sc->fail = 1;
}

// application z = M^-1 r and partial r.z: one wavefront per segment, lane = (row rr, column cc)
// of the 7x7 factor blocks; forward pass y_i = r_i - G_i y_{i-1} (kept in LDS), backward pass
// z_i = S_i^-1 y_i - G_{i+1}^T z_{i+1}.  Two dependent shuffles per row and direction.

__global__ __launch_bounds__(WG) void k_chain_apply(int r0, int r1, int seg,
                                                    const double* __restrict__ Sinv,
                                                    const double* __restrict__ Gm,
                                                    const double* __restrict__ r,
                                                    double* __restrict__ z,
                                                    const DevScalars* __restrict__ sc) {
  __shared__ double ybuf[4][CHAIN_SEG_MAX * 7];
  if (sc && sc->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int rr = l49 / 7, cc = l49 % 7;
  const int nseg = (r1 - r0 + seg - 1) / seg;
  for (int sidx = blockIdx.x * 4 + wave; sidx < nseg; sidx += gridDim.x * 4) {
    const int start = r0 + sidx * seg;
    const int end = start + seg < r1 ? start + seg : r1;
    double* yb = ybuf[wave];
    // forward
    double yprev_cc = 0.0;
    double g_nx = Gm[(size_t)49 * start + 7 * rr + cc], r_nx = r[(size_t)7 * start + rr];
    for (int i = start; i < end; ++i) {
      const double g = g_nx, ri = r_nx;
      if (i + 1 < end) {  // the next row's factor block and rhs are in flight during this row's shuffles
        g_nx = Gm[(size_t)49 * (i + 1) + 7 * rr + cc];
        r_nx = r[(size_t)7 * (i + 1) + rr];
      }
      const double prod = g * yprev_cc;
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) sum += __shfl(prod, 7 * rr + k);
      const double yi = ri - sum;  // y_i[rr], identical on the 7 lanes of row rr
      yprev_cc = __shfl(yi, 7 * cc);                   // y_i[cc] for the next row
      if (cc == 0 && lane < 49) yb[7 * (i - start) + rr] = yi;
    }
    __builtin_amdgcn_wave_barrier();
    // backward
    double znext_cc = 0.0;
    double s_nx = Sinv[(size_t)49 * (end - 1) + 7 * rr + cc], gt_nx = 0.0;
    for (int i = end - 1; i >= start; --i) {
      const double yc = yb[7 * (i - start) + cc];
      const double sv = s_nx, gt = gt_nx;  // S_i^-1 and G_{i+1}^T entries, prefetched
      if (i > start) {
        s_nx = Sinv[(size_t)49 * (i - 1) + 7 * rr + cc];
        gt_nx = Gm[(size_t)49 * i + 7 * cc + rr];
      }
      double prod = sv * yc;
      if (i + 1 < end) prod -= gt * znext_cc;  // G_{i+1}^T
      double zi = 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) zi += __shfl(prod, 7 * rr + k);
      znext_cc = __shfl(zi, 7 * cc);
      if (cc == 0 && lane < 49) z[(size_t)7 * i + rr] = zi;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// One PCG iteration in the single-reduction form (Chronopoulos & Gear): the SpMV launch before
// this one produced w = (H + lambda I) z and the partials of delta = w.z and gamma = r.z, so the
// iteration has ONE reduction point (one 2-double all-reduce on multi-GPU) and two launches:
//   beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)
//   p = z + beta p,  s = w + beta s (= A p),  x += alpha p,  r -= alpha s,  z = Minv r
// Workgroup 0 commits gamma / alpha for the next launch (ping-pong by parity, so no workgroup
// reads what another one writes in the same launch) and the stopping decision.
__global__ __launch_bounds__(WG) void k_pcg_step(int r0, int r1, int par, int it,
                                                 const double* __restrict__ scal,
                                                 const double* __restrict__ part_d,
                                                 const double* __restrict__ part_g, int npart,
                                                 const double* __restrict__ Minv,
                                                 const double* zin, double* zout,
                                                 const double* __restrict__ w,
                                                 double* __restrict__ p, double* __restrict__ sv,
                                                 double* __restrict__ x, double* __restrict__ r,
                                                 DevScalars* sc) {
  __shared__ double sh[4];
  if (sc->done) return;
  const double delta = scal ? scal[0] : sum_partials(part_d, npart, sh);
  const double gamma = scal ? scal[1] : sum_partials(part_g, npart, sh);
  const bool first = it == 0;  // it < 0: a captured (replayed) launch, never the first iteration
  const double gamma0 = first ? gamma : sc->rz0;
  const bool commit = blockIdx.x == 0 && threadIdx.x == 0;
  if (!(gamma == gamma) || gamma < 0.0 || gamma <= sc->tol2 * gamma0 || (first && gamma == 0.0)) {
    if (commit) {  // converged (x is final) or broken down; every workgroup sees the same gamma
      if (!(gamma == gamma) || gamma < 0.0) sc->fail = 1;
      if (first) sc->rz0 = gamma;
      sc->rz[par ^ 1] = gamma;
      sc->gam_last = gamma;
      sc->done = 1;
    }
    return;
  }
  const double beta = first ? 0.0 : gamma / sc->rz[par];
  const double denom = first ? delta : delta - beta * gamma / sc->alpha[par];
  if (!(denom > 0.0) || !(denom < DBL_MAX)) {  // not positive definite (g2o: Cholesky fails)
    if (commit) {
      sc->fail = 1;
      sc->done = 1;
    }
    return;
  }
  const double alpha = gamma / denom;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = r0 + (blockIdx.x * 4 + wave) * 9; row0 < r1; row0 += gridDim.x * 36) {
    const int row = row0 + sub;
    const bool act = lane < 63 && row < r1;
    const size_t j = (size_t)7 * row + rr;
    double rv = 0.0;
    if (act) {
      const double pn = zin[j] + beta * p[j];
      const double sn = w[j] + beta * sv[j];
      p[j] = pn;
      sv[j] = sn;
      x[j] += alpha * pn;
      rv = r[j] - alpha * sn;
      r[j] = rv;
    }
    if (Minv) {
      double zv = 0.0;
#pragma unroll
      for (int cc = 0; cc < 7; ++cc) {
        const double rc = __shfl(rv, base + cc);
        if (act) zv += Minv[(size_t)49 * row + 7 * rr + cc] * rc;
      }
      if (act) zout[j] = zv;
    }
  }
  if (commit) {
    if (first) sc->rz0 = gamma;
    sc->rz[par ^ 1] = gamma;
    sc->gam_last = gamma;
    sc->alpha[par ^ 1] = alpha;
    const int itn = (it < 0 ? sc->iter : it) + 1;  // only this thread ever writes sc->iter
    sc->iter = itn;
    if (itn >= sc->max_iter) sc->stop = 1;
  }
}

// Halo exchange of the row-partitioned path: the rows another rank reads are gathered into a send buffer
// (host list, grouped by that rank), the buffers travel as grouped send / receive pairs, and the rows received
// are written into the full-length vector.  Lane per entry.
__global__ __launch_bounds__(WG) void k_rows_gather(int n, const int32_t* __restrict__ rows,
                                                    const double* __restrict__ vec, double* __restrict__ buf) {
  const int t = blockIdx.x * WG + threadIdx.x;
  const int k = t / 7, c = t % 7;
  if (k < n) buf[(size_t)7 * k + c] = vec[(size_t)7 * rows[k] + c];
}
__global__ __launch_bounds__(WG) void k_rows_scatter(int n, const int32_t* __restrict__ rows,
                                                     const double* __restrict__ buf, double* __restrict__ vec) {
  const int t = blockIdx.x * WG + threadIdx.x;
  const int k = t / 7, c = t % 7;
  if (k < n) vec[(size_t)7 * rows[k] + c] = buf[(size_t)7 * k + c];
}

// ||r||^2 and ||b||^2 over a row range (the multigrid path verifies what its stopping test claims)
__global__ __launch_bounds__(WG) void k_norms2(int j0, int j1, const double* __restrict__ r,
                                               const double* __restrict__ b,
                                               double* __restrict__ pa, double* __restrict__ pb) {
  __shared__ double sh[4];
  double a = 0.0, c = 0.0;
  for (int j = j0 + blockIdx.x * WG + threadIdx.x; j < j1; j += gridDim.x * WG) {
    a += r[j] * r[j];
    c += b[j] * b[j];
  }
  const double sa = block_sum(a, sh);
  const double sb = block_sum(c, sh);
  if (threadIdx.x == 0) {
    pa[blockIdx.x] = sa;
    pb[blockIdx.x] = sb;
  }
}

// ------------------------------------------------------------------------------------------
// HBM read calibration (bench only): streams the block-CSR value array with different access
// shapes so the SpMV's achieved rate can be read against what this access shape can reach.
//   mode 0: 16 B per lane, all 64 lanes, contiguous          (the copy-kernel shape)
//   mode 1:  8 B per lane, all 64 lanes, contiguous
//   mode 2:  8 B per lane, 49 of 64 lanes, one 392-B block per wave-instruction (SpMV shape)
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(WG) void k_stream_read(const double* __restrict__ src, size_t n,
                                                    double* __restrict__ sink) {
  double acc = 0.0;
  const size_t tid = (size_t)blockIdx.x * WG + threadIdx.x, nth = (size_t)gridDim.x * WG;
  if (MODE == 0) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* s2 = reinterpret_cast<const d2*>(src);
    const size_t n2 = n / 2;
    for (size_t i = tid; i + 3 * nth < n2; i += 4 * nth) {
      const d2 a = __builtin_nontemporal_load(s2 + i), b = __builtin_nontemporal_load(s2 + i + nth);
      const d2 c = __builtin_nontemporal_load(s2 + i + 2 * nth), d = __builtin_nontemporal_load(s2 + i + 3 * nth);
      acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
  } else if (MODE == 1) {
    for (size_t i = tid; i + 7 * nth < n; i += 8 * nth) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += __builtin_nontemporal_load(src + i + u * nth);
    }
  } else {
    const int lane = threadIdx.x & 63;
    const int l49 = lane < 49 ? lane : lane - 49;
    const size_t wid = tid >> 6, nw = nth >> 6, nblk = n / 49;
    for (size_t k = wid * 8; k + 8 <= nblk; k += nw * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += __builtin_nontemporal_load(src + 49 * (k + u) + l49);
    }
  }
  if (acc == 123.456) sink[0] = acc;  // keep the loads alive
}


