// direct_kernels.hpp -- device side of the exact sparse block Cholesky (included by engine.hip after
// DevScalars; plan from direct.cpp).  Stands in for LinearSolverEigen::solve (SimplicialLDLT,
// kitti_surf.cpp:553-554) where the step has to be exact: (H + lambda I) = L L^T on the 7x7 block
// pattern, then L y = b and L^T x = y, all in elimination order, result scattered back.
//
// Left-looking by levels of the elimination tree; a level is a contiguous range of columns and of
// stored blocks (direct.hpp).  One workgroup owns one group of the schedule and walks its levels
// with workgroup barriers.  A factorisation is a few MFLOP: what it costs is dependent memory
// round trips, so a level is organised to need few of them --
//   * once per linearisation k_ldl_gather sums every block's source blocks of H into `Aperm` and
//     permutes b: a trial starts from there (+ lambda on the diagonal), without index chains;
//   * the host split every level into rounds of cells (direct.hpp): ONE scalar read gives a
//     wavefront its contiguous blocks and products; the product indices of a cell arrive by one
//     vector load (lane = product), the operands of up to 8 products are requested together;
//   * phase A+B  per block in turn: raw = Aperm (+ lambda) - sum of the listed L(i,k) L(j,k)^T in
//     list order; a diagonal block is factored right away (7x7 Cholesky and inverse through LDS,
//     forward solve y_j riding along), an off-diagonal block waits in LDS;
//   * barrier; phase C: off-diagonal blocks times L(j,j)^-T; barrier.
// The backward solve walks the levels downwards, a wavefront per column.
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
  const int32_t* pcol;
  const int32_t* gptr;
  const int32_t* lcolp;
  const int32_t* bord;
  const int32_t* brow;
  const int32_t* tpre;   // the top group's warm-up lists (blocks of L, columns of y) and their lengths
  const int32_t* tprey;
  int32_t ntpre, ntprey;
  const int32_t* rptr;
  const int32_t* cells;
  const double* vals;  // block-CSR values of H (column-major 7x7)
  const double* b;     // right-hand side, block rows of H
  double* Aperm;       // nL x 49: blocks of H in the layout of L (no damping)
  double* bp;          // 7 nb: b in elimination order
  double* L;           // nL x 49
  double* Dinv;        // nb x 49: L(j,j)^-1 (lower triangular, column-major)
  double* y;           // 7 nb, elimination order
  double* xp;          // 7 nb, elimination order
  double* x;           // 7 nb, block rows of H (the result)
  int32_t nb, nL;
  double lambda;
  DevScalars* sc;
  long long* dbg;  // tuning aid (SIM3OPT_DIRECT_TRACE): wall_clock64 stamps of the top group's levels
};

constexpr int LDL_WG_TOP = 64 * DirectPlan::CELL_WAVES;  // the top of the tree: one workgroup of 8
// wavefronts (512 threads leave each wavefront 256 VGPRs: with 1024 the operand batches spilled)
constexpr int LDL_WG_SUB = 512;                          // bottom subtrees: 8 wavefronts each too (round 3 sweep)
constexpr int LDL_CS = DirectPlan::CELL_SLOTS;
constexpr int LDL_ST = DirectPlan::CELL_STRIDE;
constexpr int LDL_NW = DirectPlan::CELL_WAVES;

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

// once per linearisation: Aperm[s] = sum of the H blocks behind block s of L; bp = permuted b
__global__ __launch_bounds__(WG) void k_ldl_gather(LdlArgs A) {
  const int lane = threadIdx.x & 63;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int nwv = gridDim.x * 4;
  for (int s = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))); s < A.nL; s += nwv) {
    double acc = 0.0;
    for (int k = A.srcptr[s]; k < A.srcptr[s + 1]; ++k) acc += A.vals[(size_t)49 * A.src[k] + l49];
    if (lane < 49) A.Aperm[(size_t)49 * s + lane] = acc;
  }
  for (int j = blockIdx.x * WG + threadIdx.x; j < 7 * A.nb; j += gridDim.x * WG)
    A.bp[j] = A.b[(size_t)7 * A.perm[j / 7] + j % 7];
}

// 7x7 Cholesky of the block in `acc` (lane = entry), its inverse, y_j; stores L(j,j), Dinv[j], y[j]
__device__ __forceinline__ void ldl_factor_diag(const LdlArgs& A, int s, int j, double acc, double tacc,
                                               double bpv, int lane, int l49, int r, int c,
                                               double* w /*98 doubles of LDS, this wavefront's*/) {
  double* wi = w + 49;
  if (lane < 49) w[lane] = acc;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {  // lower triangle, entry (r, c) at r + 7c
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
    A.L[(size_t)49 * s + lane] = lf;
    A.Dinv[(size_t)49 * j + lane] = li;
  }
  // y_j = L(j,j)^-1 (b_j - sum_k L(j,k) y_k): the sum arrives as tacc (entry (r,c) = L(j,k)(r,c) y_k(c))
  const double yraw = bpv - ldl_sum_over_c(tacc, r);    // depends on r only
  const double yc = __shfl(yraw, c);                    // lane c holds row c
  const double yr = ldl_sum_over_c(li * yc, r);
  if (lane < 7) A.y[(size_t)7 * j + lane] = yr;
  __builtin_amdgcn_wave_barrier();
}

// x_j = L(j,j)^-T (y_j - sum_{i > j} L(i,j)^T x_i)
__device__ __forceinline__ void ldl_back(const LdlArgs& A, int j, int lane, int l49, int r, int c) {
  const int s0 = A.colptr[j], s1 = A.colptr[j + 1];
  const double yj = A.y[(size_t)7 * j + c];
  const double li = A.Dinv[(size_t)49 * j + l49];
  double t = 0.0;
  for (int sb = s0 + 1; sb < s1; sb += 64) {
    // up to 64 blocks, one per lane, in the plan's fixed order of summation (bord / brow: the same
    // order under every schedule, direct.cpp)
    const int vs = sb + lane < s1 ? A.bord[sb + lane] : 0;
    const int vr = sb + lane < s1 ? A.brow[sb + lane] : 0;
    const int nn = s1 - sb < 64 ? s1 - sb : 64;
    for (int q = 0; q < nn; q += 8) {  // eight blocks (and their x rows) in flight
      double lv[8], xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (q + i < nn) {
          lv[i] = A.L[(size_t)49 * __builtin_amdgcn_readlane(vs, q + i) + l49];
          xv[i] = A.xp[(size_t)7 * __builtin_amdgcn_readlane(vr, q + i) + r];
        }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (q + i < nn) t += lv[i] * xv[i];
    }
  }
  const double z = yj - ldl_sum_over_r(t, c);  // z(c), the same in lanes (., c)
  const double p = __shfl(li, c + 7 * r) * z;  // Linv(c, r) z(c)
  const double xr = ldl_sum_over_c(p, r);
  if (lane < 7) {
    A.xp[(size_t)7 * j + lane] = xr;
    A.x[(size_t)7 * A.perm[j] + lane] = xr;
  }
}

template <bool UP, bool DOWN>
__global__ __launch_bounds__(LDL_WG_TOP) void k_ldl(LdlArgs A, int g0) {
  __shared__ double lds_raw[LDL_NW][LDL_CS][49];  // a cell's blocks: Aperm rows, then raw blocks
  __shared__ double lds_ch[LDL_NW][98];           // Cholesky scratch
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const int l49 = lane < 49 ? lane : lane - 49;
  const int r = l49 % 7, c = l49 / 7;
  const int g = g0 + blockIdx.x;
  const int lv0 = A.gptr[g], lv1 = A.gptr[g + 1];
  const bool trace = A.dbg != nullptr && UP && DOWN && threadIdx.x == 0;
  int ti = 0;
  if (UP && DOWN && A.ntpre > 0) {
    // The top group's operands from the bottom groups were written by other CUs of all XCDs: every
    // dependent load of one would be a miss of this XCD's L2 (~2 us instead of ~0.7 per round trip,
    // eight round trips per level).  Touch them all once, 16 per wavefront in flight, so that they are
    // L2 hits when the product stream asks for them (round 3; nothing is kept in registers).
    for (int t0 = wave * 16; t0 < A.ntpre; t0 += nw * 16) {
      const int vt = t0 + (lane & 15) < A.ntpre ? A.tpre[t0 + (lane & 15)] : A.tpre[A.ntpre - 1];
      double keep[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) keep[i] = A.L[(size_t)49 * __builtin_amdgcn_readlane(vt, i) + l49];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(keep[i]));
    }
    for (int t = threadIdx.x; t < A.ntprey * 7; t += blockDim.x) {
      const double yk = A.y[(size_t)7 * A.tprey[t / 7] + t % 7];
      asm volatile("" ::"v"(yk));
    }
  }
  if (UP) {
    for (int l = lv0; l < lv1; ++l) {
      const int q0 = A.rptr[l], q1 = A.rptr[l + 1];
      if (trace && ti < 254) A.dbg[ti++] = wall_clock64();
      for (int q = q0; q < q1; ++q) {
        const int32_t* cell = A.cells + (size_t)LDL_ST * q;
        const int sa = cell[wave], sb = cell[wave + 1];
        const int ka = cell[LDL_NW + 1 + wave], kb = cell[LDL_NW + 1 + wave + 1];
        const int n = sb - sa;  // <= LDL_CS
        // ---- index vectors of the cell and its Aperm rows: one round trip ----
        const int li_ = lane < n ? lane : (n > 0 ? n - 1 : 0);
        int vpp = 0, vlc = 0, vlr = 0;
        if (n > 0) {
          vpp = A.pairptr[sa + (lane <= n ? lane : n)];
          vlc = A.lcol[sa + li_];
          vlr = A.lrow[sa + li_];
        }
        int kbase = ka;
        int ia = 0, ib = 0, ic = 0;
        if (kbase + lane < kb) { ia = A.pa[kbase + lane]; ib = A.pb[kbase + lane]; ic = A.pcol[kbase + lane]; }
        {
          double tmp[LDL_CS];
#pragma unroll
          for (int t = 0; t < LDL_CS; ++t)
            if (t < n) tmp[t] = A.Aperm[(size_t)49 * (sa + t) + l49];
#pragma unroll
          for (int t = 0; t < LDL_CS; ++t)
            if (t < n && lane < 49) lds_raw[wave][t][lane] = tmp[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase A + B: the cell's products as one stream, eight operands pairs at a time ----
        int t = 0, k = ka, kend = 0, jt = 0;
        bool diag = false;
        double acc = 0.0, tacc = 0.0, bpv = 0.0;
        auto begin_slot = [&]() {
          jt = __builtin_amdgcn_readlane(vlc, t);
          diag = __builtin_amdgcn_readlane(vlr, t) == jt;
          kend = __builtin_amdgcn_readlane(vpp, t + 1);
          acc = lds_raw[wave][t][l49];
          if (diag && r == c) acc += A.lambda;
          tacc = 0.0;
          bpv = diag ? A.bp[(size_t)7 * jt + r] : 0.0;
        };
        auto end_slot = [&]() {
          if (diag) ldl_factor_diag(A, sa + t, jt, acc, tacc, bpv, lane, l49, r, c, lds_ch[wave]);
          else if (lane < 49) lds_raw[wave][t][lane] = acc;  // waits for L(j,j)^-1 (phase C)
        };
        if (n > 0) begin_slot();
        while (t < n) {
          if (k == kend) {  // this block has all its products
            end_slot();
            ++t;
            if (t < n) begin_slot();
            continue;
          }
          if (k - kbase >= 64) {  // next 64 product indices
            kbase = k;
            ia = ib = ic = 0;
            if (kbase + lane < kb) { ia = A.pa[kbase + lane]; ib = A.pb[kbase + lane]; ic = A.pcol[kbase + lane]; }
          }
          const int off = k - kbase;
          constexpr int PBATCH = 2;  // products in flight (3 and more spill: the kernel sits at 227 VGPRs)
          int m = kb - k < PBATCH ? kb - k : PBATCH;
          if (64 - off < m) m = 64 - off;
          // L(i,k) arrives in the lane's own layout and is shuffled; row c of L(j,k) -- the other
          // operand of entry (r, c) -- is read straight from memory (seven strided loads from lines
          // that one load of the block would fetch anyway): half the LDS-crossbar traffic
          double av[PBATCH], bm[PBATCH][7], yv[PBATCH];
#pragma unroll
          for (int i = 0; i < PBATCH; ++i)
            if (i < m) {
              const int s_a = __builtin_amdgcn_readlane(ia, off + i);
              const int s_b = __builtin_amdgcn_readlane(ib, off + i);
              av[i] = A.L[(size_t)49 * s_a + l49];
#pragma unroll
              for (int mm = 0; mm < 7; ++mm) bm[i][mm] = A.L[(size_t)49 * s_b + c + 7 * mm];
              yv[i] = A.y[(size_t)7 * __builtin_amdgcn_readlane(ic, off + i) + c];
            }
#pragma unroll
          for (int i = 0; i < PBATCH; ++i)
            if (i < m) {
              while (k + i == kend) {  // (a block boundary inside the batch)
                end_slot();
                ++t;
                begin_slot();  // t < n: product k + i belongs to a block of this cell
              }
#pragma unroll
              for (int mm = 0; mm < 7; ++mm) acc -= __shfl(av[i], 7 * mm + r) * bm[i][mm];
              tacc += av[i] * yv[i];  // (used by diagonal blocks only)
            }
          k += m;
        }
        __syncthreads();
        if (trace && ti < 254) A.dbg[ti++] = wall_clock64();
        // ---- phase C: off-diagonal blocks of the cell times L(j,j)^-T ----
        {
          double dv[LDL_CS];
#pragma unroll
          for (int tt = 0; tt < LDL_CS; ++tt)
            if (tt < n) {
              const int jj = __builtin_amdgcn_readlane(vlc, tt);
              dv[tt] = A.Dinv[(size_t)49 * jj + l49];
            }
#pragma unroll
          for (int tt = 0; tt < LDL_CS; ++tt)
            if (tt < n) {
              const int jj = __builtin_amdgcn_readlane(vlc, tt);
              if (__builtin_amdgcn_readlane(vlr, tt) != jj) {
                const double raw = lds_raw[wave][tt][l49];
                double xv = 0.0;
#pragma unroll
                for (int mm = 0; mm < 7; ++mm) xv += __shfl(raw, 7 * mm + r) * __shfl(dv[tt], 7 * mm + c);
                if (lane < 49) A.L[(size_t)49 * (sa + tt) + lane] = xv;
              }
            }
        }
        __syncthreads();
      }
    }
  }
  if (trace && ti < 254) A.dbg[ti++] = wall_clock64();
  if (DOWN) {
    for (int l = lv1 - 1; l >= lv0; --l) {
      const int c0 = A.lcolp[l], c1 = A.lcolp[l + 1];
      for (int j = c0 + wave; j < c1; j += nw) ldl_back(A, j, lane, l49, r, c);
      __syncthreads();
    }
  }
  if (trace) { if (ti < 254) A.dbg[ti++] = wall_clock64(); A.dbg[255] = ti; }
}
