// spmv_kernel.hpp -- the software-pipelined 7x7 block-CSR SpMV with its multigrid epilogues (a template:
// instantiated by engine_pcg.hip for the PCG's own product, by engine_amg.hip for the cycle's matrix passes and by
// engine_batch.hip for K right-hand sides at once -- ONE kernel source: the one-system kernel is K = 1; measured
// on the driver's command: 47.1-47.2 LM it/s against 46.6-46.9 with a separate one-system kernel).
// LinearSolverEigen's role, kitti_surf.cpp:553-554; SURVEY.md 8(a) row a9.
#pragma once
// q = (H + lambda I) p with the partial dot products p.q and (optionally) rvec.p per workgroup --
// the block-CSR SpMV of the PCG (LinearSolverEigen's role, kitti_surf.cpp:553-554).
// One wavefront owns a CONTIGUOUS span of block rows (host table `wrow`, balanced by block count);
// lane = one of the 49 entries of the current 7x7 block, so its blocks and column indices are one
// contiguous HBM stream, software-pipelined across row boundaries:
//   * the loads of chunk k+1 (CH blocks of 392 B + ONE shared gather of p: lane 7u+c reads
//     p[7 col_u + c]) are in flight while chunk k is consumed; the p entries reach the (r, c) lanes
//     through the LDS crossbar (ds_bpermute), which is otherwise idle -- with one gather per block
//     the address unit, not HBM, was the co-bottleneck (measured);
//   * lanes 49..63 mirror lanes 0..14: every lane issues a valid coalesced load, no exec masking;
//   * column indices / row ends: one coalesced vector load per 64, then v_readlane / ds_bpermute;
//   * NT: the once-read block stream bypasses the cache policy so p stays in L2 / Infinity Cache;
//   * a row ends with a wave-uniform branch (reduce 7 columns, add lambda p, store q, dots).
// MODE 0: q = A p (+ the dot partials; the PCG's SpMV).  The multigrid preconditioner reuses the
// same stream for its two matrix passes per level: MODE 1: q = rvec - A p (residual),
// MODE 2: q = p + Minv (rvec - A p) (one damped block-Jacobi step; Minv = omega D^-1, row-major).
// MODE 3 (coarse multigrid levels): MODE 2 applied to p + xc[agg] -- the piecewise-constant
// prolongation of the coarser level's correction is added while the input vector is gathered
// (xc through `partials_r`, which the non-PCG modes do not use).
// VT = float: the multigrid preconditioner's matrix passes stream an FP32 copy of the blocks (half
// the bytes; vectors, accumulation and the smoother inverses stay FP64) -- the PCG's own SpMV
// (MODE 0) always reads the FP64 blocks.
#ifndef SIM3OPT_SPMV_FASTPATH
#define SIM3OPT_SPMV_FASTPATH 1
#endif
constexpr bool FASTPATH = SIM3OPT_SPMV_FASTPATH != 0;

// K > 1 (engine_batch.hip: the rejected trials of one LM iteration, solved together): the block stream, the
// column indices and the row bookkeeping are shared, everything that depends on the vector is an array over the
// systems; vectors of system s live at base + s * stride (BatchStrides).  Level 0 adds the damping as lambda_s x
// at the row end; on a coarse level (DIAGK) the damping sits in the diagonal block, so system s takes ITS
// diagonal block from diagk and the shared stream's diagonal block is skipped.  Per system the operations and
// their order are those of K = 1 (0 + d x is exact): the K solutions are bit for bit K one-system results.
// gridDim.y > 1: every slice of the grid takes K of the systems (coarse levels: launch-latency-bound).
struct BatchStrides {
  int64_t vec;   // between the systems' vectors of this level (doubles)
  int64_t minv;  // between their smoother inverses (49 doubles per row)
  int64_t xc;    // between their coarse corrections (mode 3; vectors of the next level)
  int64_t diag;  // between their damped diagonal blocks (coarse levels; 49 floats per row)
  int part;      // between their arrays of partial sums
};

#ifndef SIM3OPT_BATCH_WAVES
#define SIM3OPT_BATCH_WAVES 0  // tuning: force this many wavefronts per SIMD (0: the compiler's choice)
#endif
template <int CH, bool NT, int MODE, typename VT = double, int K = 1, bool DIAGK = false>
__global__ __launch_bounds__(WG)
#if SIM3OPT_BATCH_WAVES > 0
__attribute__((amdgpu_waves_per_eu(SIM3OPT_BATCH_WAVES, SIM3OPT_BATCH_WAVES)))
#endif
void k_spmv_span(int nb, const int32_t* __restrict__ wrow,
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
  // The shared gather of the input reaches the (r, c) lanes through a wavefront-private LDS copy: one ds_write_b64 per
  // chunk, one ds_read_b64 per block and system (seven distinct words: broadcasts, no bank conflicts) -- half the LDS
  // instructions of the two ds_bpermute_b32 a 64-bit shuffle costs, the same values in the same order (bit-identical).
  // End of round 4, A/B on one box: 47.2 -> 49.3 LM it/s on the driver's command, the burst iteration 136 -> 128 ms.
  // (0: the shuffles, kept for A/B)
#ifndef SIM3OPT_X_LDS
#define SIM3OPT_X_LDS 1
#endif
  // (not for the FP64 pass over several systems: measured 260 -> 279 us for four; the one-system FP64 pass is neutral)
  constexpr bool XLDS = SIM3OPT_X_LDS != 0 && !(sizeof(VT) == 8 && K > 1);
  constexpr int NGX = (CH + 7) / 8;
  __shared__ double xl[XLDS ? 4 : 1][XLDS ? K : 1][XLDS ? NGX : 1][64];
  double (*xw)[NGX][64] = xl[XLDS ? (threadIdx.x >> 6) : 0];
  // ... and the row epilogue's sums over the seven columns likewise: lane (r, c) writes its term to slot 8 r + c, lane r
  // reads its row back (three ds_read_b128 + one b64) and adds c = 0 ... 6 in that order -- the order of the six
  // 64-bit shuffles it replaces (twelve ds_bpermute_b32), bit-identical
#ifndef SIM3OPT_ROWEND_LDS
#define SIM3OPT_ROWEND_LDS 1  // (0: the shuffles, kept for A/B)
#endif
  constexpr bool RLDS = XLDS && SIM3OPT_ROWEND_LDS != 0;
  __shared__ __attribute__((aligned(16))) double rl[XLDS ? 4 : 1][64];
  double* const rw = rl[XLDS ? (threadIdx.x >> 6) : 0];
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
  // entry `src` (a lane index of the shared gather) of block position u of the current chunk, system s
  auto xget = [&](const double (*xg)[NG], int s, int u, int src) -> double {
    if (XLDS) return xw[s][u / 8][src];
    return __shfl(xg[s][u / 8], src);
  };
  auto xstage = [&](const double (*xg)[NG]) {  // (one wavefront: wave_lds_sync, no workgroup barrier)
    if (XLDS) {
      wave_lds_sync();  // (the reads of the previous chunk are done)
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) xw[s][g][threadIdx.x & 63] = xg[s][g];
      wave_lds_sync();
    }
  };
  auto row_begin = [&](int row, int u, const double (*xg)[NG]) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      pi_n[s] = xget(xg, s, u, 7 * (u % 8) + r);
      if (rvec) rv_n[s] = rvec[(size_t)s * bs.vec + (size_t)7 * row + r];
      if (MODE >= 2) mv[s] = Minv[(size_t)s * bs.minv + (size_t)49 * row + l49];  // symmetric: entry (r, c49)
      // the row's own (per-system, damped) diagonal block times its own entries of the input: what the single
      // system's stream adds first (0 + d x is exact, so the row sum is bit for bit the one-system sum)
      if (DIAGK)
        acc[s] = (double)diagk[(size_t)s * bs.diag + (size_t)49 * row + l49] * xget(xg, s, u, 7 * (u % 8) + c49);
      else
        acc[s] = 0.0;
    }
  };
  // a block row is complete: reduce its 7 columns, add the damping, apply the epilogue
  // (row sums are valid in lanes 0..6)
  typedef double d2_t __attribute__((ext_vector_type(2)));
  auto colsum = [&](double v) -> double {  // sum over c of the terms of lanes (r, c); valid in lanes 0..6
    if (RLDS) {
      wave_lds_sync();
      rw[8 * r + c49] = v;
      wave_lds_sync();
      const d2_t* pp = reinterpret_cast<const d2_t*>(rw + 8 * (lane < 7 ? lane : 0));
      const d2_t a = pp[0], b = pp[1], c = pp[2];
      const double e = rw[8 * (lane < 7 ? lane : 0) + 6];
      return (((((a.x + a.y) + b.x) + b.y) + c.x) + c.y) + e;
    }
    double y = v;
#pragma unroll
    for (int cc = 1; cc < 7; ++cc) y += __shfl(v, r + 7 * cc);
    return y;
  };
  auto row_end = [&](int row) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      double y = colsum(acc[s]);
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
          double dc;  // d_c for lane (r, c)
          if (RLDS) {
            if (lane < 7) rw[56 + lane] = d;
            wave_lds_sync();
            dc = rw[56 + c49];
          } else {
            dc = __shfl(d, c49);
          }
          const double pr_ = mv[s] * dc;  // Minv(r, c) d_c
          const double o = colsum(pr_);
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
    xstage(xgc);
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
            for (int u = 0; u < 4; ++u) xs[u] = xget(xgc, s, h + u, 7 * ((h + u) % 8) + c49);
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
            for (int s = 0; s < K; ++s) acc[s] += vv * xget(xgc, s, u, 7 * (u % 8) + c49);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) vc[u] = vn[u];
#pragma unroll
      for (int s = 0; s < K; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) xgc[s][g] = xgn[s][g];
      xstage(xgc);
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
