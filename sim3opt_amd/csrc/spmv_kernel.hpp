// spmv_kernel.hpp -- the software-pipelined 7x7 block-CSR SpMV with its multigrid epilogues (a template:
// instantiated by engine_pcg.hip for the PCG's own product and by engine_amg.hip for the cycle's matrix
// passes).  LinearSolverEigen's role, kitti_surf.cpp:553-554; SURVEY.md 8(a) row a9.
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
template <int CH, bool NT, int MODE, typename VT = double>
__global__ __launch_bounds__(WG)
// (no occupancy floor: the FP32 smoothing pass at 88 VGPRs / 5 wavefronts per SIMD without spills runs 0.5-1 %
// faster end to end than forced to 80 VGPRs / 6 wavefronts with 3-5 spilled registers; r3_negative_results.log)
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
                                                  double xc_scale) {
  __shared__ double sh[4];
  __shared__ double sh2[4];
  __shared__ int sh_cnt;
  if (MODE == 2 || MODE == 0) {  // (arrival counter of the barrier-free partial sums below)
    if (threadIdx.x == 0) sh_cnt = 0;
    __syncthreads();
  }
  if (sc) {
    if (sc->done) return;
    if (lam_sc) lambda = sc->lambda;  // captured launches cannot carry a per-solve kernel argument
    // the previous update was the last allowed one: later launches become no-ops
    if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
      if (sc->stop) sc->done = 1;
      sc->n_spmv_work += 1;  // (launches are stream-ordered: one writer at a time)
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
  double pq = 0.0, pr = 0.0;
  // per-row operands are requested when the row starts and consumed when it ends
  double pi_n = 0.0, rv_n = 0.0, mv = 0.0;
  // (every row starts with its diagonal block, so the row's own entries of p are the gather of that
  // block -- position u of the chunk in flight: a shuffle instead of one more vector-memory
  // instruction per row; the kernel is bound by the number of those, not by their bytes)
  auto row_begin = [&](int row, int u, const double* xg) {
    pi_n = __shfl(xg[u / 8], 7 * (u % 8) + r);
    if (rvec) rv_n = rvec[(size_t)7 * row + r];
    if (MODE >= 2) mv = Minv[(size_t)49 * row + l49];  // symmetric: entry (r, c49)
  };
  // a block row is complete: reduce its 7 columns, add the damping, apply the epilogue
  // (row sums are valid in lanes 0..6)
  auto row_end = [&](int row, double acc) {
    double y = acc;
#pragma unroll
    for (int cc = 1; cc < 7; ++cc) y += __shfl(acc, r + 7 * cc);
    const double pi = pi_n;
    y += lambda * pi;
    if (MODE == 0) {
      if (lane < 7) {
        q[(size_t)7 * row + lane] = y;
        pq += pi * y;
        if (rvec) pr += rv_n * pi;
      }
    } else {
      const double d = rv_n - y;
      if (MODE == 1) {
        if (lane < 7) q[(size_t)7 * row + lane] = d;
      } else {
        const double pr_ = mv * __shfl(d, c49);  // Minv(r, c) d_c
        double o = pr_;
#pragma unroll
        for (int cc = 1; cc < 7; ++cc) o += __shfl(pr_, r + 7 * cc);
        if (lane < 7) {
          const double zo = pi + o;
          q[(size_t)7 * row + lane] = zo;
          // the PCG's r.z where z is born (level 0's last pass writes z = M^-1 r and holds r): the
          // SpMV that follows then needs no load of r -- 11 us of its 166 (measured)
          if (MODE == 2 && partials) pr += rv_n * zo;
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
    double acc = 0.0;
    VT vc[CH], vn[CH];
    double xgc[NG], xgn[NG];
    auto gather = [&](int ks, double* xg) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int kk = ks + 8 * g + gu < kend ? ks + 8 * g + gu : kend - 1;
        const int colu = __shfl(cv, kk - cbase);
        xg[g] = p[(size_t)7 * colu + gc];
        if (MODE == 3) xg[g] += xc_scale * partials_r[(size_t)7 * agg[colu] + gc];
      }
    };
#pragma unroll
    for (int g = 0; g < NG; ++g) xgn[g] = 0.0;
    // prologue: chunk at k0
    load_chunk(k0, vc);
    gather(k0, xgc);
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
        row_end(row, acc);
        acc = 0.0;
        ++row;
        row_begin(row, u, xgc);
        if (row - rbase >= 64) {
          rbase += 64;
          rpv = rbase + 1 + lane <= rB ? rowptr[rbase + 1 + lane] : kend;
        }
        k1 = __builtin_amdgcn_readlane(rpv, row - rbase);
      };
      if (interior && k1 >= k + CH) {
#pragma unroll
        for (int h = 0; h < CH; h += 4) {  // (four at a time: eight live values cost the FP32 smoothing pass its occupancy)
          double xs[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) xs[u] = __shfl(xgc[(h + u) / 8], 7 * ((h + u) % 8) + c49);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += (double)vc[h + u] * xs[u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int kk = k + u;
          if (kk >= kbeg && kk < kend) {
            if (kk == k1) next_row(u);
            acc += (double)vc[u] * __shfl(xgc[u / 8], 7 * (u % 8) + c49);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) vc[u] = vn[u];
#pragma unroll
      for (int g = 0; g < NG; ++g) xgc[g] = xgn[g];
    }
    row_end(row, acc);  // last row of the span
  }
  if (MODE == 2 && partials) {
    // no barrier at the end of a streaming kernel: every wavefront leaves its sum in LDS and goes;
    // the one that arrives last adds the four in index order (deterministic) and writes the partial
    const double t = wave_sum(pr);
    if (lane == 0) {
      sh[threadIdx.x >> 6] = t;
      __threadfence_block();
      if (atomicAdd(&sh_cnt, 1) == 3) {
        __threadfence_block();
        partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
      }
    }
    return;
  }
  if (MODE != 0) return;
  {  // the same barrier-free partial sums (w.z, and r.z when this pass reads r)
    const double s = wave_sum(pq);
    const double t = rvec ? wave_sum(pr) : 0.0;
    if (lane == 0) {
      sh[threadIdx.x >> 6] = s;
      sh2[threadIdx.x >> 6] = t;
      __threadfence_block();
      if (atomicAdd(&sh_cnt, 1) == 3) {
        __threadfence_block();
        if (partials) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (rvec) partials_r[blockIdx.x] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
      }
    }
  }
}
