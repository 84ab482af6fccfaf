// symm_proto.hpp -- MEASUREMENT PROTOTYPE, not on the product path (sim3opt_bench_spmv_symmetric).
//
// SURVEY.md 8(d) quotes 0.54 GB per PCG iteration for upper-triangle storage against 0.94 GB for the
// full-symmetric block CSR the solver streams.  This is the deterministic two-phase form of that
// SpMV, written to find out what it would buy before the assembly, the Galerkin products and the
// smoothers (all of which read whole rows) are rebuilt around a second storage scheme:
//   phase 1  per stored block (i, j), j >= i:  y_i += A x_j   and, for j > i,  t_k = A^T x_i  (7 doubles
//            per block, written in block order -- one contiguous stream)
//   phase 2  y_j += sum of the t_k of the blocks stored in column j, in a fixed order (host-built list)
// Lane map of phase 1: lane = r + 8 c (56 lanes): the sum over r that t needs stays inside groups
// of 8 lanes (three xor steps), the sum over c that y needs is three more at the end of the row.
#pragma once
// (included inside namespace sim3opt)

__global__ __launch_bounds__(WG) void k_symm_copy(int nU, const int32_t* __restrict__ usrc,
                                                  const double* __restrict__ vals, double* __restrict__ uvals) {
  const int lane = threadIdx.x & 63;
  const int l49 = lane < 49 ? lane : lane - 49;
  for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nU; k += gridDim.x * 4) {
    const double v = vals[(size_t)49 * usrc[k] + l49];
    if (lane < 49) uvals[(size_t)49 * k + lane] = v;
  }
}

__global__ __launch_bounds__(WG) void k_symm_phase1(int nb, const int32_t* __restrict__ urowptr,
                                                    const int32_t* __restrict__ ucol,
                                                    const double* __restrict__ uvals,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    double* __restrict__ tvec) {
  const int lane = threadIdx.x & 63;
  const int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (i >= nb) return;
  const int r = lane & 7, c = lane >> 3;
  const bool act = r < 7 && c < 7;
  const int e = act ? r + 7 * c : 0;
  const int cc = c < 7 ? c : 0, rr = r < 7 ? r : 0;
  const double xi = x[(size_t)7 * i + rr];
  double accy = 0.0;
  const int k0 = urowptr[i], k1 = urowptr[i + 1];
  for (int k = k0; k < k1; k += 4) {
    int j[4];
    double a[4], xj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k + u < k1 ? k + u : k1 - 1;
      j[u] = ucol[kk];
      a[u] = __builtin_nontemporal_load(uvals + (size_t)49 * kk + e);
      xj[u] = x[(size_t)7 * j[u] + cc];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k + u >= k1) break;
      const double av = act ? a[u] : 0.0;
      accy += av * xj[u];
      double tt = av * xi;
      tt += __shfl_xor(tt, 1);
      tt += __shfl_xor(tt, 2);
      tt += __shfl_xor(tt, 4);
      if (r == 0 && c < 7 && j[u] != i) tvec[(size_t)7 * (k + u) + c] = tt;
    }
  }
  accy += __shfl_xor(accy, 8);
  accy += __shfl_xor(accy, 16);
  accy += __shfl_xor(accy, 32);
  if (lane < 7) y[(size_t)7 * i + lane] = accy;
}

__global__ __launch_bounds__(WG) void k_symm_phase2(int n7, const int32_t* __restrict__ lptr,
                                                    const int32_t* __restrict__ lidx,
                                                    const double* __restrict__ tvec, double* __restrict__ y) {
  const int t = blockIdx.x * WG + threadIdx.x;
  if (t >= n7) return;
  const int j = t / 7, c = t % 7;
  double acc = y[t];
  int m = lptr[j];
  const int m1 = lptr[j + 1];
  for (; m + 3 < m1; m += 4) {  // four gathers in flight
    const double v0 = tvec[(size_t)7 * lidx[m] + c], v1 = tvec[(size_t)7 * lidx[m + 1] + c];
    const double v2 = tvec[(size_t)7 * lidx[m + 2] + c], v3 = tvec[(size_t)7 * lidx[m + 3] + c];
    acc += (v0 + v1) + (v2 + v3);
  }
  for (; m < m1; ++m) acc += tvec[(size_t)7 * lidx[m] + c];
  y[t] = acc;
}

// Round 3: phase 1 with the product kernel's machinery -- a contiguous span of rows per wavefront
// (balanced by stored upper blocks), chunks of 8 blocks software-pipelined across row boundaries,
// ONE shared gather of x per chunk (lane 7u + c reads x[7 col_u + c]), column indices 64 at a time --
// and the t vectors of a chunk written by ONE store: after the xor butterfly every lane of the 8-lane
// group c holds t_u(c), so lane (r, c) keeps the one of block u = r and the 56 lanes write the 448
// contiguous bytes of the chunk's eight t vectors (diagonal blocks get a slot nobody reads).
// lane exchange inside rows of 16 lanes on the VALU (DPP) instead of the LDS crossbar: the three xor
// steps per block of the t reduction made the first span version LDS-instruction-bound (it took
// the time of the naive one)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over the 8 lanes r = 0..7 of a group (all 8 get it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ double sum8_dpp(double t) {
  t += dpp_f64<0xB1>(t);
  t += dpp_f64<0x4E>(t);
  t += dpp_f64<0x141>(t);
  return t;
}

__global__ __launch_bounds__(WG) void k_symm_phase1_span(int nb, const int32_t* __restrict__ uwrow,
                                                         const int32_t* __restrict__ urowptr,
                                                         const int32_t* __restrict__ ucol,
                                                         const double* __restrict__ uvals,
                                                         const double* __restrict__ x, double* __restrict__ y,
                                                         double* __restrict__ tvec) {
  constexpr int CH = 8;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int r = lane & 7, c = lane >> 3;
  const bool act = r < 7 && c < 7;
  const int e = act ? r + 7 * c : 0;
  const int cc = c < 7 ? c : 0, rr = r < 7 ? r : 0;
  const int gu = lane / 7 < CH ? lane / 7 : CH - 1, gc = lane % 7;
  const int rA = uwrow[w], rB = uwrow[w + 1];
  if (rA >= rB) return;
  const int kbeg = urowptr[rA], kend = urowptr[rB];
  int rbase = rA;
  int rpv = rbase + 1 + lane <= rB ? urowptr[rbase + 1 + lane] : kend;
  int row = rA;
  int k1 = __builtin_amdgcn_readlane(rpv, 0);
  int cbase = kbeg;
  int cv = cbase + lane < kend ? ucol[cbase + lane] : 0;
  int cvn = cbase + 64 + lane < kend ? ucol[cbase + 64 + lane] : 0;
  double vc[CH], vn[CH];
  double xgc, xgn = 0.0;
  auto load_chunk = [&](int ks, double* dst) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int kk = ks + u < kend ? ks + u : kend - 1;
      dst[u] = __builtin_nontemporal_load(uvals + (size_t)49 * kk + e);
    }
  };
  {
    load_chunk(kbeg, vc);
    const int kk = kbeg + gu < kend ? kbeg + gu : kend - 1;
    xgc = x[(size_t)7 * __shfl(cv, kk - cbase) + gc];
  }
  double xi = __shfl(xgc, rr);  // a row starts with its diagonal block: its own entries of x
  double acc = 0.0;
  auto row_end = [&](int rw, double a) {
    a += __shfl_xor(a, 8);
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 32);
    if (lane < 7) y[(size_t)7 * rw + lane] = a;
  };
  for (int k = kbeg; k < kend; k += CH) {
    const int kn = k + CH;
    if (kn < kend) {
      if (kn - cbase >= 64) {
        cbase += 64;
        cv = cvn;
        cvn = cbase + 64 + lane < kend ? ucol[cbase + 64 + lane] : 0;
      }
      load_chunk(kn, vn);
      const int kk = kn + gu < kend ? kn + gu : kend - 1;
      xgn = x[(size_t)7 * __shfl(cv, kk - cbase) + gc];
    }
    double tmine = 0.0;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int kk = k + u;
      if (kk < kend) {
        if (kk == k1) {  // the previous row is complete; this block is the next row's diagonal
          row_end(row, acc);
          acc = 0.0;
          ++row;
          if (row - rbase >= 64) {
            rbase += 64;
            rpv = rbase + 1 + lane <= rB ? urowptr[rbase + 1 + lane] : kend;
          }
          k1 = __builtin_amdgcn_readlane(rpv, row - rbase);
          xi = __shfl(xgc, 7 * u + rr);
        }
        const double av = act ? vc[u] : 0.0;
        acc += av * __shfl(xgc, 7 * u + cc);
        const double tt = sum8_dpp(av * xi);
        if (r == u) tmine = tt;
      }
    }
    if (c < 7 && k + r < kend) tvec[(size_t)7 * (k + r) + c] = tmine;
#pragma unroll
    for (int u = 0; u < CH; ++u) vc[u] = vn[u];
    xgc = xgn;
  }
  row_end(row, acc);
}
