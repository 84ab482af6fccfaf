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
//     wavefront its contiguous blocks and products, and every wavefront the product range of the round;
//   * the operands of ALL the round's products (final blocks of earlier levels, y rows) are fetched by
//     all wavefronts together into LDS -- one round trip for the indices, one for the blocks, four
//     products per wavefront in flight (round 3; a round holds at most STAGE_PRODUCTS, a block with more
//     is staged piece by piece);
//   * phase A+B  per block in turn: raw = Aperm (+ lambda) - sum of the listed L(i,k) L(j,k)^T in
//     list order, entry (r, c) = row r of one staged block times row c of the other, read from LDS
//     as broadcasts (no shuffles); a diagonal block is factored right away (7x7 Cholesky and inverse
//     in registers, forward solve y_j riding along), an off-diagonal block waits in LDS;
//   * barrier; phase C: off-diagonal blocks times L(j,j)^-T; barrier.
// The backward solve walks the levels downwards, a wavefront per column.
// Lane l of a wavefront holds entry l49 = l mod 49 of a column-major 7x7 block (lanes 49..63 mirror
// lanes 0..14: every lane issues a valid load).  No atomics, fixed summation order that does not
// depend on the schedule (direct.cpp): bit-reproducible.
// Where the time of a level went (wall_clock64 stamps, KITTI-00, a level of one column with 16
// products): round 2 -- 8 memory round trips for the products, two at a time, 6 us; 7x7 Cholesky by one
// lane walking its LDS copy, 2.6 us; 10-12 us in all.  Now: staging 1.5, products from LDS 2.9, Cholesky
// + inverse in registers + y 2.0, phase C 0.8 = 7.2 us.
#pragma once
// (included inside namespace sim3opt)

#include "direct_args.hpp"

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
  // The 7x7 Cholesky and the triangular inverse run in REGISTERS (round 3): every lane reads the block from
  // LDS once (28 broadcast reads) and works through the same operations in the same order as before, when
  // lane 0 walked the LDS copy entry by entry -- ~200 dependent LDS round trips, 7 of a level's 10 us.
  double a[7][7];  // lower triangle
#pragma unroll
  for (int cc = 0; cc < 7; ++cc)
#pragma unroll
    for (int rr = cc; rr < 7; ++rr) a[rr][cc] = w[rr + 7 * cc];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double d = a[k][k];
#pragma unroll
    for (int m = 0; m < k; ++m) d -= a[k][m] * a[k][m];
    if (!(d > 0.0) || !(d < DBL_MAX)) { ok = false; d = 1.0; }  // not positive definite (g2o: the solve fails)
    const double lkk = sqrt(d), inv = 1.0 / lkk;
    a[k][k] = lkk;
#pragma unroll
    for (int rr = k + 1; rr < 7; ++rr) {
      double v = a[rr][k];
#pragma unroll
      for (int m = 0; m < k; ++m) v -= a[rr][m] * a[k][m];
      a[rr][k] = v * inv;
    }
  }
  if (!ok && lane == 0) A.sc->fail = A.fail_token;
  // column `lane` of the inverse of the lower-triangular factor (lanes 0..6; the others follow along)
  double wic[7];
  {
    const int cc = lane < 7 ? lane : 0;
#pragma unroll
    for (int rr = 0; rr < 7; ++rr) {
      double v = rr == cc ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < rr; ++m) v -= a[rr][m] * (m >= cc ? wic[m] : 0.0);
      wic[rr] = rr >= cc ? v / a[rr][rr] : 0.0;
    }
  }
  __builtin_amdgcn_wave_barrier();  // (everybody has read w)
  if (lane == 0) {
#pragma unroll
    for (int cc = 0; cc < 7; ++cc)
#pragma unroll
      for (int rr = 0; rr < 7; ++rr) w[rr + 7 * cc] = rr >= cc ? a[rr][cc] : 0.0;
  }
  if (lane < 7) {
#pragma unroll
    for (int rr = 0; rr < 7; ++rr) wi[rr + 7 * lane] = wic[rr];
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
  // the operands of a round's products (L(i,k), L(j,k), y_k), staged by all wavefronts together
  __shared__ double st_a[UP ? LDL_STAGE : 1][49];
  __shared__ double st_b[UP ? LDL_STAGE : 1][49];
  __shared__ double st_y[UP ? LDL_STAGE : 1][7];
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
        const int ka = cell[LDL_NW + 1 + wave];
        const int K0 = cell[LDL_NW + 1], K1 = cell[LDL_NW + 1 + LDL_NW];  // all products of the round
        const int n = sb - sa;  // <= LDL_CS
        // ---- index vectors of the cell and its Aperm rows: one round trip ----
        const int li_ = lane < n ? lane : (n > 0 ? n - 1 : 0);
        int vpp = 0, vlc = 0, vlr = 0;
        if (n > 0) {
          vpp = A.pairptr[sa + (lane <= n ? lane : n)];
          vlc = A.lcol[sa + li_];
          vlr = A.lrow[sa + li_];
        }
        {
          double tmp[LDL_CS];
#pragma unroll
          for (int t = 0; t < LDL_CS; ++t)
            if (t < n) tmp[t] = A.Aperm[(size_t)49 * (sa + t) + l49];
#pragma unroll
          for (int t = 0; t < LDL_CS; ++t)
            if (t < n && lane < 49) lds_raw[wave][t][lane] = tmp[t];
        }
        // ---- phase A + B: the cell's products as one stream out of LDS ----
        // A product's operands are final blocks of earlier levels.  Fetched by the wavefront that owns the
        // target block they cost it a memory round trip per two products -- eight in a row where a level
        // is one column with 16 products and seven wavefronts wait at the barrier (round 2).  Now every
        // wavefront fetches its share of ALL the round's operands (four products in flight) into LDS and
        // the owners run their chains of multiply-adds -- the same ones in the same order -- from there.
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
        bool started = false;
        // A round of at most LDL_STAGE products is staged by everybody at once (a level that is one column
        // with 16 products: eight wavefronts fetch two each).  A wider round has work for every wavefront:
        // each stages its OWN products, LDL_WCH at a time, in its own slice of the buffers -- no workgroup
        // barrier, no waiting for the slowest.
        const bool coop = K1 - K0 <= LDL_STAGE;
        const int kbw = cell[LDL_NW + 1 + wave + 1];
        const int P0 = coop ? K0 : ka, P1 = coop ? K1 : kbw, PS = coop ? LDL_STAGE : LDL_WCH;
        const int sbase = coop ? 0 : wave * LDL_WCH, pstride = coop ? nw : 1, poff = coop ? wave : 0;
        for (int c0 = P0;; c0 += PS) {
          const int c1 = P1 - c0 < PS ? P1 : c0 + PS;
          if (c1 > c0) {  // stage this wavefront's share of products [c0, c1)
            const int pl = c0 + poff + pstride * lane;
            int ja = 0, jb = 0, jc = 0;
            if (pl < c1) { ja = A.pa[pl]; jb = A.pb[pl]; jc = A.pcol[pl]; }
            const int cnt = c1 - c0 > poff ? (c1 - c0 - poff + pstride - 1) / pstride : 0;
            for (int i0 = 0; i0 < cnt; i0 += 4) {
              double va[4], vb[4], vy[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (i0 + i < cnt) {
                  va[i] = A.L[(size_t)49 * __builtin_amdgcn_readlane(ja, i0 + i) + l49];
                  vb[i] = A.L[(size_t)49 * __builtin_amdgcn_readlane(jb, i0 + i) + l49];
                  vy[i] = A.y[(size_t)7 * __builtin_amdgcn_readlane(jc, i0 + i) + r];
                }
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (i0 + i < cnt) {
                  const int slot = sbase + poff + pstride * (i0 + i);
                  if (lane < 49) { st_a[slot][lane] = va[i]; st_b[slot][lane] = vb[i]; }
                  if (lane < 7) st_y[slot][lane] = vy[i];
                }
            }
          }
          if (coop) {
            __syncthreads();  // (also orders this wavefront's lds_raw rows before its own reads)
          } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
          if (!started) {
            started = true;
            if (n > 0) begin_slot();
          }
          while (t < n) {
            if (k == kend) {  // this block has all its products
              end_slot();
              ++t;
              if (t < n) begin_slot();
              continue;
            }
            if (k >= c1) break;  // the rest of this block's products are in the next piece
            // entry (r, c) of L(i,k) L(j,k)^T: row r of the one staged block times row c of the other, read
            // straight from LDS (seven lanes share an address: broadcasts) -- no shuffles; two products of a
            // block at a time where it has two left (a rotating prefetch of product k + 1 cost more in
            // register copies than the LDS round trip it hid: measured)
            const int sl = sbase + k - c0;
            if (k + 1 < kend && k + 1 < c1) {  // two products of this block: their reads in flight together
              double am[7], bm[7], am2[7], bm2[7];
#pragma unroll
              for (int mm = 0; mm < 7; ++mm) {
                am[mm] = st_a[sl][r + 7 * mm];
                bm[mm] = st_b[sl][c + 7 * mm];
              }
              const double av = st_a[sl][l49], yv = st_y[sl][c];
#pragma unroll
              for (int mm = 0; mm < 7; ++mm) {
                am2[mm] = st_a[sl + 1][r + 7 * mm];
                bm2[mm] = st_b[sl + 1][c + 7 * mm];
              }
              const double av2 = st_a[sl + 1][l49], yv2 = st_y[sl + 1][c];
#pragma unroll
              for (int mm = 0; mm < 7; ++mm) acc -= am[mm] * bm[mm];
              tacc += av * yv;
#pragma unroll
              for (int mm = 0; mm < 7; ++mm) acc -= am2[mm] * bm2[mm];
              tacc += av2 * yv2;
              k += 2;
              continue;
            }
            double am[7], bm[7];
#pragma unroll
            for (int mm = 0; mm < 7; ++mm) {
              am[mm] = st_a[sl][r + 7 * mm];
              bm[mm] = st_b[sl][c + 7 * mm];
            }
            const double av = st_a[sl][l49], yv = st_y[sl][c];
#pragma unroll
            for (int mm = 0; mm < 7; ++mm) acc -= am[mm] * bm[mm];
            tacc += av * yv;  // (used by diagonal blocks only)
            ++k;
          }
          if (c1 >= P1) break;
          if (coop) __syncthreads();  // everybody is done with this piece before the next one overwrites it
          else __builtin_amdgcn_wave_barrier();
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
