// batch_kernels.hpp -- the PCG loop for K right-hand sides at once (included by engine_batch.hip only, inside
// namespace sim3opt, after spmv_kernel.hpp): the rejected trials of one LM iteration solve
// (H + lambda_k I) x_k = b for a KNOWN sequence lambda_k (g2o's OptimizationAlgorithmLevenberg: lambda *= nu,
// nu *= 2 after every rejection), so after the first rejection the next K systems are solved together --
// ONE pass over the matrix blocks for K vectors, K vectors per coarse launch -- and the trials are then
// evaluated in g2o's order.  What LinearSolverEigen does K times in a row (kitti_surf.cpp:553-554, 675).
// Every kernel here performs, per system, the operations of its one-system counterpart in the same order:
// the K solutions are bit for bit those of K sequential solves (asserted in tests/test_gpu_parity.py).
// Vectors of system s live at base + s * stride (BatchStrides).
#pragma once

struct BatchStrides {
  int64_t vec;   // between the systems' vectors of this level (doubles)
  int64_t minv;  // between their smoother inverses (49 doubles per row)
  int64_t xc;    // between their coarse corrections (mode 3; vectors of the next level)
  int64_t diag;  // between their damped diagonal blocks (coarse levels; 49 floats per row)
  int part;      // between their arrays of partial sums
};

// The block-CSR SpMV of spmv_kernel.hpp for K input vectors: the block stream, the column indices and the row
// bookkeeping are shared, everything that depends on the vector is an array over the systems.  Level 0 adds the
// damping as lambda_s x at the row end; on a coarse level (DIAGK) the damping sits in the diagonal block, so
// system s takes ITS diagonal block from diagk and the shared stream's diagonal block is skipped.
#ifndef SIM3OPT_BATCH_WAVES
#define SIM3OPT_BATCH_WAVES 0  // tuning: force this many wavefronts per SIMD (0: the compiler's choice)
#endif
template <int CH, bool NT, int MODE, typename VT, int K, bool DIAGK>
__global__ __launch_bounds__(WG)
#if SIM3OPT_BATCH_WAVES > 0
__attribute__((amdgpu_waves_per_eu(SIM3OPT_BATCH_WAVES, SIM3OPT_BATCH_WAVES)))
#endif
void k_spmv_span_k(int nb, const int32_t* __restrict__ wrow,
                                                  const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ colidx,
                                                  const VT* __restrict__ vals,
                                                  const double* __restrict__ p,
                                                  double* __restrict__ q, double lambda,
                                                  double* __restrict__ partials,
                                                  const double* __restrict__ rvec,
                                                  double* __restrict__ partials_r,
                                                  DevScalars* __restrict__ sc,
                                                  const double* __restrict__ Minv, int lam_sc,
                                                  const int32_t* __restrict__ agg,
                                                  double xc_scale, BatchStrides bs,
                                                  const float* __restrict__ diagk) {
  __shared__ double sh[K][4];
  __shared__ double sh2[K][4];
  __shared__ int sh_cnt;
  if (MODE == 2 || MODE == 0) {  // (arrival counter of the barrier-free partial sums below)
    if (threadIdx.x == 0) sh_cnt = 0;
    __syncthreads();
  }
  // gridDim.y > 1 (coarse levels: launch-latency-bound, the matrix sits in cache): every slice of the grid
  // takes K of the systems -- more wavefronts instead of longer ones
  if (gridDim.y > 1) {
    const size_t s0 = (size_t)blockIdx.y * K;
    p += s0 * bs.vec;
    q += s0 * bs.vec;
    if (rvec) rvec += s0 * bs.vec;
    if (Minv) Minv += s0 * bs.minv;
    if (MODE == 3) partials_r += s0 * bs.xc;
    if (DIAGK) diagk += s0 * bs.diag;
    if (partials) partials += s0 * bs.part;
    if (sc) sc += s0;
  }
  // per-system damping (level 0: a scalar added at the row end; coarse levels carry it in their per-system
  // diagonal blocks, DIAGK); a finished system's vectors are computed along and ignored by the PCG step
  double lam[K];
#pragma unroll
  for (int s = 0; s < K; ++s) lam[s] = lambda;
  if (sc) {
    bool all_done = true;
#pragma unroll
    for (int s = 0; s < K; ++s) all_done = all_done && sc[s].done;
    if (all_done) return;
    if (lam_sc) {
#pragma unroll
      for (int s = 0; s < K; ++s) lam[s] = sc[s].lambda;
    }
    if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int s = 0; s < K; ++s)
        if (sc[s].stop) sc[s].done = 1;
      sc[0].n_spmv_work += 1;  // (launches are stream-ordered: one writer at a time)
    }
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int r = lane % 7;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int c49 = l49 / 7;
  constexpr int NG = (CH + 7) / 8;  // shared gathers of p per chunk: eight blocks each
  const int gu = lane / 7 < 8 ? lane / 7 : 7, gc = lane % 7;
  const int rA = wrow[w], rB = wrow[w + 1];
  double pq[K], pr[K];
  // per-row operands are requested when the row starts and consumed when it ends
  double pi_n[K], rv_n[K], mv[K], acc[K];
#pragma unroll
  for (int s = 0; s < K; ++s) pq[s] = pr[s] = pi_n[s] = rv_n[s] = mv[s] = acc[s] = 0.0;
  int kfirst = 0;  // index of the current row's first (= diagonal) block
  // (every row starts with its diagonal block, so the row's own entries of p are the gather of that
  // block -- position u of the chunk in flight: a shuffle instead of one more vector-memory
  // instruction per row; the kernel is bound by the number of those, not by their bytes)
  auto row_begin = [&](int row, int u, const double (*xg)[NG]) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      pi_n[s] = __shfl(xg[s][u / 8], 7 * (u % 8) + r);
      if (rvec) rv_n[s] = rvec[(size_t)s * bs.vec + (size_t)7 * row + r];
      if (MODE >= 2) mv[s] = Minv[(size_t)s * bs.minv + (size_t)49 * row + l49];  // symmetric: entry (r, c49)
      // the row's own (per-system, damped) diagonal block times its own entries of the input: what the single
      // system's stream adds first (0 + d x is exact, so the row sum is bit for bit the one-system sum)
      if (DIAGK)
        acc[s] = (double)diagk[(size_t)s * bs.diag + (size_t)49 * row + l49] * __shfl(xg[s][u / 8], 7 * (u % 8) + c49);
      else
        acc[s] = 0.0;
    }
  };
  // a block row is complete: reduce its 7 columns, add the damping, apply the epilogue
  // (row sums are valid in lanes 0..6)
  auto row_end = [&](int row) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      double y = acc[s];
#pragma unroll
      for (int cc = 1; cc < 7; ++cc) y += __shfl(acc[s], r + 7 * cc);
      const double pi = pi_n[s];
      y += lam[s] * pi;
      double* qs = q + (size_t)s * bs.vec;
      if (MODE == 0) {
        if (lane < 7) {
          qs[(size_t)7 * row + lane] = y;
          pq[s] += pi * y;
          if (rvec) pr[s] += rv_n[s] * pi;
        }
      } else {
        const double d = rv_n[s] - y;
        if (MODE == 1) {
          if (lane < 7) qs[(size_t)7 * row + lane] = d;
        } else {
          const double pr_ = mv[s] * __shfl(d, c49);  // Minv(r, c) d_c
          double o = pr_;
#pragma unroll
          for (int cc = 1; cc < 7; ++cc) o += __shfl(pr_, r + 7 * cc);
          if (lane < 7) {
            const double zo = pi + o;
            qs[(size_t)7 * row + lane] = zo;
            if (MODE == 2 && partials) pr[s] += rv_n[s] * zo;
          }
        }
      }
    }
  };
  if (rA < rB) {
    const int kbeg = rowptr[rA], kend = rowptr[rB];
    // row ends of this span, 64 at a time, one per lane
    int rbase = rA;
    int rpv = rbase + 1 + lane <= rB ? rowptr[rbase + 1 + lane] : kend;
    int row = rA;
    int k1 = __builtin_amdgcn_readlane(rpv, 0);
    // FP32 blocks come in interleaved pairs (f32_pair_index): chunks start at an even block index,
    // a leading block of the previous span is loaded and skipped
    constexpr bool PAIR = sizeof(VT) == 4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int k0 = PAIR ? (kbeg & ~1) : kbeg;
    const int pmax = (kend - 1) >> 1;
    auto load_chunk = [&](int ks, VT* dst) {
      if (PAIR) {
#pragma unroll
        for (int u = 0; u < CH; u += 2) {
          const int pp = (ks + u) >> 1;
          const f32x2* vp = reinterpret_cast<const f32x2*>(vals) + (size_t)49 * (pp < pmax ? pp : pmax) + l49;
          const f32x2 t = NT ? __builtin_nontemporal_load(vp) : *vp;
          dst[u] = (VT)t.x;
          dst[u + 1] = (VT)t.y;
        }
      } else {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int kk = ks + u < kend ? ks + u : kend - 1;
          const VT* vp = vals + (size_t)49 * kk + l49;
          dst[u] = NT ? __builtin_nontemporal_load(vp) : *vp;
        }
      }
    };
    // column indices, 64 blocks at a time, one per lane; window w covers [k0 + 64 w, +64)
    int cbase = k0;
    int cv = cbase + lane < kend ? colidx[cbase + lane] : 0;
    int cvn = cbase + 64 + lane < kend ? colidx[cbase + 64 + lane] : 0;
    VT vc[CH], vn[CH];
    double xgc[K][NG], xgn[K][NG];
    auto gather = [&](int ks, double (*xg)[NG]) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int kk = ks + 8 * g + gu < kend ? ks + 8 * g + gu : kend - 1;
        const int colu = __shfl(cv, kk - cbase);
        const int ag = MODE == 3 ? agg[colu] : 0;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          xg[s][g] = p[(size_t)s * bs.vec + (size_t)7 * colu + gc];
          if (MODE == 3) xg[s][g] += xc_scale * partials_r[(size_t)s * bs.xc + (size_t)7 * ag + gc];
        }
      }
    };
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
      for (int g = 0; g < NG; ++g) xgn[s][g] = 0.0;
    // prologue: chunk at k0
    load_chunk(k0, vc);
    gather(k0, xgc);
    kfirst = kbeg;
    row_begin(row, kbeg - k0, xgc);
    for (int k = k0; k < kend; k += CH) {
      const int kn = k + CH;
      if (kn < kend) {  // issue the next chunk before consuming this one
        if (kn - cbase >= 64) {  // next chunk starts a new 64-block window (CH divides 64)
          cbase += 64;
          cv = cvn;
          cvn = cbase + 64 + lane < kend ? colidx[cbase + 64 + lane] : 0;
        }
        load_chunk(kn, vn);
        gather(kn, xgn);
      }
      // a chunk that lies inside the span and inside the current row (two in three on config 3) needs no
      // per-block tests, and its shuffles are in flight together -- the same products in the same order
      // (bit-identical; round 3, A/B on one box: -6.5 % on the FP64 pass, -13...16 % on the coarse levels'
      // passes, the level-0 FP32 passes unchanged).  A third path for interior chunks WITH a row boundary
      // (no validity tests) raised the register count and lost more than it won
      // (profiles/r3_negative_results.log)
      const bool interior = FASTPATH && k >= kbeg && k + CH <= kend;
      auto next_row = [&](int u) {  // row `row` is complete; block u of this chunk starts the next one
        row_end(row);
        ++row;
        kfirst = k + u;
        row_begin(row, u, xgc);
        if (row - rbase >= 64) {
          rbase += 64;
          rpv = rbase + 1 + lane <= rB ? rowptr[rbase + 1 + lane] : kend;
        }
        k1 = __builtin_amdgcn_readlane(rpv, row - rbase);
      };
      // (DIAGK: a chunk may hold the row's diagonal block, which is replaced per system: the tested path)
      if (interior && k1 >= k + CH && !(DIAGK && kfirst >= k)) {
#pragma unroll
        for (int h = 0; h < CH; h += 4) {
#pragma unroll
          for (int s = 0; s < K; ++s) {
            double xs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xs[u] = __shfl(xgc[s][(h + u) / 8], 7 * ((h + u) % 8) + c49);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[s] += (double)vc[h + u] * xs[u];
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int kk = k + u;
          if (kk >= kbeg && kk < kend) {
            if (kk == k1) next_row(u);
            const double vv = DIAGK && kk == kfirst ? 0.0 : (double)vc[u];
#pragma unroll
            for (int s = 0; s < K; ++s) acc[s] += vv * __shfl(xgc[s][u / 8], 7 * (u % 8) + c49);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) vc[u] = vn[u];
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) xgc[s][g] = xgn[s][g];
    }
    row_end(row);  // last row of the span
  }
  // barrier-free partial sums per system (see k_spmv_span): the wavefront that arrives last adds the four in
  // index order -- the same sums in the same order as the one-system kernel
  if (MODE == 2 && partials) {
    double t[K];
#pragma unroll
    for (int s = 0; s < K; ++s) t[s] = wave_sum(pr[s]);
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < K; ++s) sh[s][threadIdx.x >> 6] = t[s];
      __threadfence_block();
      if (atomicAdd(&sh_cnt, 1) == 3) {
        __threadfence_block();
#pragma unroll
        for (int s = 0; s < K; ++s)
          partials[(size_t)s * bs.part + blockIdx.x] = (sh[s][0] + sh[s][1]) + (sh[s][2] + sh[s][3]);
      }
    }
    return;
  }
  if (MODE != 0) return;
  {
    double sa[K], sb[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      sa[s] = wave_sum(pq[s]);
      sb[s] = rvec ? wave_sum(pr[s]) : 0.0;
    }
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < K; ++s) {
        sh[s][threadIdx.x >> 6] = sa[s];
        sh2[s][threadIdx.x >> 6] = sb[s];
      }
      __threadfence_block();
      if (atomicAdd(&sh_cnt, 1) == 3) {
        __threadfence_block();
#pragma unroll
        for (int s = 0; s < K; ++s) {
          if (partials) partials[(size_t)s * bs.part + blockIdx.x] = (sh[s][0] + sh[s][1]) + (sh[s][2] + sh[s][3]);
          if (rvec) partials_r[(size_t)s * bs.part + blockIdx.x] = (sh2[s][0] + sh2[s][1]) + (sh2[s][2] + sh2[s][3]);
        }
      }
    }
  }
}

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
