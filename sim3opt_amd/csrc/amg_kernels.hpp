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
// Every fine block is read exactly once here, so the FP32 copy the cycle's matrix passes stream
// (vals32_f, may be null) is written on the way.
template <bool HASP>
__global__ __launch_bounds__(WG) void k_amg_galerkin(int cb0, int cb1, const int32_t* __restrict__ gptr,
                                                     const int32_t* __restrict__ gblk,
                                                     const int32_t* __restrict__ grow,
                                                     const int32_t* __restrict__ colidx_f,
                                                     const double* __restrict__ vals_f,
                                                     const double* __restrict__ P,
                                                     double* __restrict__ vals_c,
                                                     float* __restrict__ vals32_f) {
  const int lane = threadIdx.x & 63;
  const int cb = cb0 + __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (cb >= cb1) return;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  double acc = 0.0;
  const int e0 = gptr[cb], e1 = gptr[cb + 1];
  // The lists of a coarse block arrive by vector loads, one entry per lane (64 at a time; an aggregate of 8
  // rows has at most 64 fine blocks) -- fine block, its row's and its column's vertex -- so that the operand
  // loads of every fine block are independent of everything but those: the kernel is a chain of memory round
  // trips per wavefront, and with an index chain in front of every block it ran three per fine block
  // (0.98 ms per linearisation on config 3; round 3).
  for (int eb = e0; eb < e1; eb += 64) {
    const int n = e1 - eb < 64 ? e1 - eb : 64;
    const int le = lane < n ? lane : n - 1;
    const int kkv = gblk[eb + le];
    if (HASP) {
      // The operands of the two products that exist in memory -- column c of P_j, column r of P_i --
      // are loaded in the layout the lane needs them in (seven consecutive doubles each, from lines
      // the cache holds); only A and T = A P_j go through the LDS crossbar.  With all four operands
      // shuffled the kernel was bound by that crossbar (56 ds_bpermute per fine block); the texture
      // path was idle.  Two fine blocks' operands in flight.  Same products in the same order as ever:
      // bit-identical results.
      const int giv = grow[eb + le];
      const int cjv = colidx_f[kkv];
      double av[2], pj[2][7], pi[2][7];
      int kk[2];
      auto load = [&](int u, int b) {
        kk[b] = __builtin_amdgcn_readlane(kkv, u);
        av[b] = vals_f[(size_t)49 * kk[b] + l49];
        const double* pjp = P + (size_t)49 * __builtin_amdgcn_readlane(cjv, u) + 7 * c;
        const double* pip = P + (size_t)49 * __builtin_amdgcn_readlane(giv, u) + 7 * r;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          pj[b][q] = pjp[q];
          pi[b][q] = pip[q];
        }
      };
      auto consume = [&](int b) {
        if (vals32_f && lane < 49) vals32_f[f32_pair_index(kk[b], lane)] = (float)av[b];
        double t = 0.0;  // T = A P_j
#pragma unroll
        for (int q = 0; q < 7; ++q) t += __shfl(av[b], r + 7 * q) * pj[b][q];
        double o = 0.0;  // P_i^T T
#pragma unroll
        for (int q = 0; q < 7; ++q) o += pi[b][q] * __shfl(t, q + 7 * c);
        acc += o;
      };
      for (int u = 0; u < n; ++u) {
        load(u, 0);
        consume(0);
      }
    } else {
      for (int u0 = 0; u0 < n; u0 += 4) {
        const int m = n - u0 < 4 ? n - u0 : 4;
        int kk[4];
        double av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < m) {
            kk[u] = __builtin_amdgcn_readlane(kkv, u0 + u);
            av[u] = vals_f[(size_t)49 * kk[u] + l49];
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < m) {
            if (vals32_f && lane < 49) vals32_f[f32_pair_index(kk[u], lane)] = (float)av[u];
            acc += av[u];
          }
      }
    }
  }
  if (lane < 49) vals_c[(size_t)49 * cb + lane] = acc;
}

// keeps the undamped Galerkin diagonal blocks (a trial overwrites the ones inside vals)
__global__ __launch_bounds__(WG) void k_amg_copydiag(int lo, int hi, const int32_t* __restrict__ rowptr,
                                                     const double* __restrict__ vals,
                                                     double* __restrict__ diagH) {  // rows [lo, hi): this rank's
  for (int idx = 49 * lo + blockIdx.x * WG + threadIdx.x; idx < 49 * hi; idx += gridDim.x * WG)
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

// Coarse levels (piecewise-constant prolongation): r_c[a] = sum over members i of t_f[i]; then
// x_c[a] = Minv_c[a] r_c[a] (first smoothing step of the coarser level from a zero guess).
// 63 lanes = 9 aggregates x 7 entries.
__global__ __launch_bounds__(WG) void k_amg_restrict(int a_lo, int nc, const int32_t* __restrict__ mptr,
                                                     const int32_t* __restrict__ mem,
                                                     const double* __restrict__ t_f,
                                                     double* __restrict__ r_c,
                                                     const double* __restrict__ Minv_c,
                                                     double* __restrict__ x_c) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int a0 = a_lo + (blockIdx.x * 4 + wave) * 9; a0 < nc; a0 += gridDim.x * 36) {  // aggregates [a_lo, nc)
    const int a = a0 + sub;
    const bool act = lane < 63 && a < nc;
    const int e0 = act ? mptr[a] : 0, e1 = act ? mptr[a + 1] : 0;
    double acc = 0.0;
    int e = e0;
    for (; e + 3 < e1; e += 4) {  // four members in flight
      const int i0 = mem[e], i1 = mem[e + 1], i2 = mem[e + 2], i3 = mem[e + 3];
      const double v0 = t_f[(size_t)7 * i0 + rr], v1 = t_f[(size_t)7 * i1 + rr];
      const double v2 = t_f[(size_t)7 * i2 + rr], v3 = t_f[(size_t)7 * i3 + rr];
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; e < e1; ++e) acc += t_f[(size_t)7 * mem[e] + rr];
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

// Level-0 restriction, one wavefront per aggregate: lane (m, c) = entry l49 = m + 7c of P_i (one
// coalesced 392-byte read per member), r_c[c] = sum_i sum_m P_i[m][c] t_i[m]; then the coarse
// level's first smoothing step x_c = Minv_c r_c with Minv_c read the same way.
__global__ __launch_bounds__(WG) void k_amg_restrict0(int a_lo, int nc, const int32_t* __restrict__ mptr,
                                                      const int32_t* __restrict__ mem,
                                                      const double* __restrict__ P,
                                                      const double* __restrict__ t_f,
                                                      double* __restrict__ r_c,
                                                      const double* __restrict__ Minv_c,
                                                      double* __restrict__ x_c,
                                                      const DevScalars* __restrict__ sc) {
  if (sc && sc->done) return;
  const int lane = threadIdx.x & 63;
  const int a = a_lo + __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));  // [a_lo, nc)
  if (a >= nc) return;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int m = l49 % 7, c = l49 / 7;
  double acc = 0.0;
  const int e0 = mptr[a], e1 = mptr[a + 1];
  for (int e = e0; e < e1; ++e) {
    const int i = mem[e];  // (multi-GPU: an aggregate's members all belong to the rank that owns it)
    acc += P[(size_t)49 * i + l49] * t_f[(size_t)7 * i + m];
  }
  // sum over m inside each group of 7 lanes (fixed c): lanes 7c .. 7c+6
  double rc = 0.0;
#pragma unroll
  for (int q = 0; q < 7; ++q) rc += __shfl(acc, 7 * c + q);
  if (lane < 49 && m == 0) r_c[(size_t)7 * a + c] = rc;
  if (Minv_c) {  // x_c[m] = sum_c Minv[m][c] r_c[c]; lane (m, c) holds r_c[c]
    const double pr = Minv_c[(size_t)49 * a + l49] * rc;  // symmetric: entry (m, c)
    double xv = pr;
#pragma unroll
    for (int q = 1; q < 7; ++q) xv += __shfl(pr, m + 7 * ((c + q) % 7));
    if (lane < 7) x_c[(size_t)7 * a + lane] = xv;
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
// rows [row_lo, row_hi), or -- with a list -- rows list[row_lo .. row_hi)
template <bool HASP>
__global__ __launch_bounds__(WG) void k_amg_prolong(int row_lo, int row_hi, const int32_t* __restrict__ list,
                                                    const int32_t* __restrict__ agg,
                                                    const double* __restrict__ P,
                                                    const double* __restrict__ x_c,
                                                    const double* x_in, double* x_out,
                                                    const DevScalars* __restrict__ sc, double scale) {
  if (sc && sc->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / 7, rr = lane % 7, base = lane - rr;
  for (int row0 = row_lo + (blockIdx.x * 4 + wave) * 9; row0 < row_hi; row0 += gridDim.x * 36) {
    const bool act = lane < 63 && row0 + sub < row_hi;
    const int row = act ? (list ? list[row0 + sub] : row0 + sub) : 0;
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
    if (act) x_out[(size_t)7 * row + rr] = x_in[(size_t)7 * row + rr] + scale * add;
  }
}

// Coarsest level (<= AMG_MAX_COARSEST block rows): dense copy of its block-CSR matrix (unique
// columns per row, damping already in the diagonal blocks) ...
constexpr int AMG_DENSE_MAX_N = 7 * AMG_MAX_COARSEST;
__global__ __launch_bounds__(WG) void k_amg_dense_fill(int nb, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ colidx,
                                                       const double* __restrict__ vals,
                                                       double* __restrict__ A,
                                                       const double* __restrict__ diag64 = nullptr) {
  // diag64 (one of several systems solved together): the damped diagonal blocks of THIS system, [row][49]
  const int n = 7 * nb;
  const int nnz = 49 * rowptr[nb];
  for (int t = blockIdx.x * WG + threadIdx.x; t < nnz; t += gridDim.x * WG) {
    const int k = t / 49, e = t % 49;
    int lo = 0, hi = nb;  // block row of block k: last i with rowptr[i] <= k
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rowptr[mid] <= k) lo = mid; else hi = mid;
    }
    const double v = diag64 && k == rowptr[lo] ? diag64[(size_t)49 * lo + e] : vals[(size_t)49 * k + e];
    A[(size_t)(7 * lo + e % 7) * n + 7 * colidx[k] + e / 7] = v;
  }
}

// ... inverted by block Gauss-Jordan without pivot search (SPD: positive pivots), one launch per
// pivot block of PB = 14 rows (a last one of 7 when the row count is odd; every launch costs ~13 us
// whatever its work, so half the launches is half the time), out of place (B = step(A), buffers
// ping-pong) so that no workgroup reads what another one overwrites:
//   P = A_kk^-1;  B_kk = P;  B_kj = P A_kj;  B_ik = -A_ik P;  B_ij = A_ij - A_ik (P A_kj)
// A workgroup owns a 64 x 64 tile and keeps its slices of P A_k. and A_.k in LDS.  The inverse of the
// pivot block is not on the step's critical path (round 3; it was 8 of a step's 23 us): P arrives
// from the previous launch, and one extra workgroup (blockIdx.y == 0, dispatched first) looks ahead --
// it updates the NEXT pivot block exactly as its tile would, inverts it in its first wavefront with
// one row per lane (shuffles, no barrier) and leaves it for the next launch.

// in-place Gauss-Jordan inverse of an NP x NP block, row `lane` in lane < NP of one wavefront
template <int NP>
__device__ inline bool gj_invert_rows(double (&prow)[NP], int lane) {
  bool spd = true;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    double pk[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) pk[c] = __shfl(prow[c], k);
    if (!(pk[k] > 0.0)) spd = false;
    const double d = 1.0 / pk[k];
    const double f = prow[k];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const double rk = pk[j] * d;  // scaled pivot row
      if (lane == k) prow[j] = j == k ? d : rk;
      else prow[j] = j == k ? -f * d : prow[j] - f * rk;
    }
  }
  return spd;
}

// the same elimination with a row spread over two lanes: lane (r = lane & 31, h = lane >> 5) holds columns
// [h NP/2, (h+1) NP/2) of row r -- half the multiply-adds, selects and cross-lane reads per lane (a 28 x 28
// block: 23 -> 13 us; the look-ahead workgroup's chain load -> update -> invert is what a step waits for)
template <int NP>
__device__ inline bool gj_invert_halfrows(double (&prow)[NP / 2], int lane) {
  constexpr int H = NP / 2;
  const int r = lane & 31, h = lane >> 5;
  bool spd = true;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int hk = k / H, kk = k % H;
    double pk[H];
#pragma unroll
    for (int c = 0; c < H; ++c) pk[c] = __shfl(prow[c], k + 32 * h);
    const double pkk = __shfl(prow[kk], k + 32 * hk);
    if (!(pkk > 0.0)) spd = false;
    const double d = 1.0 / pkk;
    const double f = __shfl(prow[kk], r + 32 * hk);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const double rk = pk[j] * d;  // scaled pivot row
      const bool pivcol = j == kk && h == hk;
      if (r == k) prow[j] = pivcol ? d : rk;
      else prow[j] = pivcol ? -f * d : prow[j] - f * rk;
    }
  }
  return spd;
}

// the NP x NP block at src (row stride ld) -> its inverse in Pout (NP x NP, dense), by one wavefront
template <int NP>
__device__ inline void gj_invert_block(const double* src, int ld, double* __restrict__ Pout, int lane,
                                       DevScalars* sc) {
  bool spd;
  if constexpr (NP % 2 == 0) {
    constexpr int H = NP / 2;
    double prow[H];
    const int r = lane & 31, h = lane >> 5, rr = r < NP ? r : 0;
#pragma unroll
    for (int c = 0; c < H; ++c) prow[c] = src[(size_t)rr * ld + H * h + c];
    spd = gj_invert_halfrows<NP>(prow, lane);
    if (r < NP) {
#pragma unroll
      for (int c = 0; c < H; ++c) Pout[r * NP + H * h + c] = prow[c];
    }
  } else {
    double prow[NP];
    const int rr = lane < NP ? lane : 0;
#pragma unroll
    for (int c = 0; c < NP; ++c) prow[c] = src[(size_t)rr * ld + c];
    spd = gj_invert_rows<NP>(prow, lane);
    if (lane < NP) {
#pragma unroll
      for (int c = 0; c < NP; ++c) Pout[lane * NP + c] = prow[c];
    }
  }
  if (!spd && lane == 0) sc->fail = 1;
}

// the first pivot block's inverse (one wavefront)
template <int PB>
__global__ __launch_bounds__(64) void k_amg_dense_gj_first(int n, const double* __restrict__ A,
                                                           double* __restrict__ Pout, DevScalars* sc) {
  gj_invert_block<PB>(A, n, Pout, threadIdx.x, sc);
}

// pbn = rows of the next pivot block (<= PB; 0: this is the last step).  Two workgroups per CU (<= 256 VGPRs:
// the tile grid of 1141 unknowns is 324 workgroups, a little more than one per CU)
template <int PB>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(2))) void k_amg_dense_gj_step(int n, int k0, const double* __restrict__ A,
                                                          double* __restrict__ B,
                                                          const double* __restrict__ Pin,
                                                          double* __restrict__ Pout, int pbn,
                                                          DevScalars* sc) {
  __shared__ double P[PB][PB + 1];
  __shared__ double ak[PB][64];       // pivot rows A_k. for the tile's columns (look-ahead: then the updated next pivot block)
  __shared__ double rowk[PB][64];     // (P A_k.) for the tile's columns
  __shared__ double colk[64][PB + 1]; // A_.k for the tile's rows
  const int tid = threadIdx.x;
  const bool ahead = blockIdx.y == 0;
  if (ahead && (blockIdx.x != 0 || pbn == 0)) return;
  // the look-ahead workgroup's "tile" is the next pivot block
  const int ext = ahead ? pbn : 64;
  const int i0 = ahead ? k0 + PB : (int)(blockIdx.y - 1) * 64, j0 = ahead ? k0 + PB : (int)blockIdx.x * 64;
  // every global read is issued before anything waits -- the tile itself included (a thread owns the 4 x 4
  // elements (ty + 16 a, tx + 16 b): 128-byte row segments per 16 lanes): one memory round trip per step
  const int ty = tid >> 4, tx = tid & 15;
  double v[4][4];
  if (ahead) {  // (at most 28 x 28 = 784 elements: 4 per thread, element t = tid + 256 e)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = tid + WG * e;
      v[0][e] = t < ext * ext ? A[(size_t)(i0 + t / ext) * n + j0 + t % ext] : 0.0;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
        v[a][b] = i < n && j < n ? A[(size_t)i * n + j] : 0.0;
      }
  }
  for (int t = tid; t < PB * PB; t += WG) P[t / PB][t % PB] = Pin[t];
  for (int t = tid; t < ext * PB; t += WG) {
    const int i = t / PB, m = t % PB;
    colk[i][m] = i0 + i < n ? A[(size_t)(i0 + i) * n + k0 + m] : 0.0;
  }
  for (int t = tid; t < PB * 64; t += WG) {
    const int q = t / 64, j = t % 64;
    ak[q][j] = j < ext && j0 + j < n ? A[(size_t)(k0 + q) * n + j0 + j] : 0.0;
  }
  __syncthreads();
  // R = P A_k. for the tile's columns; in the pivot columns themselves R = P, which turns the four cases of
  // the update into one:  B_ij = R_(i-k0)j on pivot rows, else (A_ij, or 0 in a pivot column) - sum_m A_im R_mj
  for (int t = tid; t < PB * 64; t += WG) {
    const int m = t / 64, j = t % 64, jp = j0 + j - k0;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < PB; ++q) acc += P[m][q] * ak[q][j];
    rowk[m][j] = jp >= 0 && jp < PB ? P[m][jp] : acc;
  }
  __syncthreads();
  if (ahead) {
    // (no row or column of the next pivot block lies in the current one)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = tid + WG * e;
      if (t < ext * ext) {
        const int il = t / ext, jl = t % ext;
        double w = v[0][e];
#pragma unroll
        for (int m = 0; m < PB; ++m) w -= colk[il][m] * rowk[m][jl];
        ak[il][jl] = w;
      }
    }
    __syncthreads();
    if (tid < 64) {
      if (pbn == PB) gj_invert_block<PB>(&ak[0][0], 64, Pout, tid, sc);
      else if (pbn == 14) gj_invert_block<14>(&ak[0][0], 64, Pout, tid, sc);
      else gj_invert_block<7>(&ak[0][0], 64, Pout, tid, sc);
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int jp = j0 + tx + 16 * b - k0;
    if (jp >= 0 && jp < PB) {
#pragma unroll
      for (int a = 0; a < 4; ++a) v[a][b] = 0.0;
    }
  }
#pragma unroll 7
  for (int m = 0; m < PB; ++m) {
    double cm[4], rm[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) cm[a] = colk[ty + 16 * a][m];
#pragma unroll
    for (int b = 0; b < 4; ++b) rm[b] = rowk[m][tx + 16 * b];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) v[a][b] -= cm[a] * rm[b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ty + 16 * a, ip = i - k0;
    const bool ik = ip >= 0 && ip < PB;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int jl = tx + 16 * b, j = j0 + jl;
      if (i < n && j < n) B[(size_t)i * n + j] = ik ? rowk[ip][jl] : v[a][b];
    }
  }
}

// x = Ainv r on the coarsest level: a wavefront per row, lanes along the row (coalesced; r is a few
// KB and stays in cache)
__global__ __launch_bounds__(WG) void k_amg_dense_apply(int n, const double* __restrict__ Ainv,
                                                        const double* __restrict__ r,
                                                        double* __restrict__ x,
                                                        const DevScalars* __restrict__ sc) {
  if (sc && sc->done) return;
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
    const double* row = Ainv + (size_t)i * n;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four loads in flight per lane
    int j = lane;
    for (; j + 192 < n; j += 256) {
      const double m0 = row[j], m1 = row[j + 64], m2 = row[j + 128], m3 = row[j + 192];
      a0 += m0 * r[j]; a1 += m1 * r[j + 64]; a2 += m2 * r[j + 128]; a3 += m3 * r[j + 192];
    }
    for (; j < n; j += 64) a0 += row[j] * r[j];
    double acc = (a0 + a1) + (a2 + a3);
    acc = wave_sum(acc);
    if (lane == 0) x[i] = acc;
  }
}

// FP32 copy of a block array (the multigrid's matrix passes read it)
__global__ __launch_bounds__(WG) void k_to_f32(size_t n, const double* __restrict__ src,
                                               float* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG)
    dst[f32_pair_index((int64_t)(i / 49), (int)(i % 49))] = (float)src[i];
}


