// rowlane_proto.hpp -- MEASUREMENT PROTOTYPE, not on the product path (sim3opt_bench_spmv_rowlane).
//
// The product SpMV (spmv_kernel.hpp) gives a lane ONE entry of the current 7x7 block; the entries of the input
// vector reach the lanes through the LDS crossbar (two ds_bpermute_b32 per block and system), and that crossbar
// is what the level-0 FP32 passes and, four-fold, the batched passes queue on (counters).
// This is the other mapping, written to find out what it would buy before the cycle's matrix passes are rebuilt
// around a second FP32 layout:
//   * a GROUP of 7 lanes walks its own span of block rows, one block per step; lane r of the group holds ROW r of
//     the block (FP32, row-major copy: 28 contiguous bytes) and ALL seven entries of the input vector at the
//     block's column (56 bytes, the same for the seven lanes): the 7x7 product is seven in-lane multiply-adds --
//     no cross-lane traffic per block, none per row in the residual pass, seven shuffles per row in the
//     smoothing pass (Minv d);
//   * nine groups per wavefront (63 of 64 lanes busy instead of 49), each with its own row boundaries (exec-masked
//     row epilogue); the loads of step k+1 and the column index of step k+2 are in flight while step k is consumed;
//   * K systems share the block stream (K gathers of the input, K accumulators);
//   * RL_XLDS (default): a lane gathers ONE entry of the input (x[7 col + r]: 56 contiguous bytes per group, one
//     vector-memory instruction per step and system) and the group's seven entries reach every lane through LDS
//     (one ds_write_b64, three ds_read_b128 + one ds_read_b64, no bank conflicts: the seven lanes of a group read
//     the same words) -- the texture-address unit, not memory, bounded the first version, which gathered all
//     seven entries per lane (four instructions per step and system; counters in profiles/r4_negative_results.log).
// MODE 1: q = rvec - A p;  MODE 2: q = p + Minv (rvec - A p) (+ per-wavefront partial sums of rvec . q).
#pragma once
// (included inside namespace sim3opt)

// FP32 copy of the column-major FP64 blocks in two row-major planes per block: columns 0..3 of row r at
// 49 k + 4 r (16 bytes per lane, 112 contiguous bytes per lane group), columns 4..6 at 49 k + 28 + 3 r (12 bytes per
// lane, 84 contiguous bytes): two fully coalesced loads per step, 196 bytes per block as before
__global__ __launch_bounds__(WG) void k_rl_copy(size_t n, const double* __restrict__ src, float* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const size_t k = i / 49;
    const int e = (int)(i % 49), r = e % 7, c = e / 7;
    dst[49 * k + (c < 4 ? 4 * r + c : 28 + 3 * r + (c - 4))] = (float)src[i];
  }
}

#ifndef RL_NT
#define RL_NT 1
#endif
#ifndef RL_XLDS
#define RL_XLDS 1
#endif
typedef float rl_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float rl_f3 __attribute__((ext_vector_type(3), aligned(4)));
typedef double rl_d2 __attribute__((ext_vector_type(2), aligned(8)));

template <int MODE, int K>
__global__ __launch_bounds__(WG) void k_spmv_rowlane(int ngroups, const int32_t* __restrict__ grow,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colidx, int nnzb,
                                                     const float* __restrict__ vals, const double* __restrict__ p,
                                                     double* __restrict__ q, const double* __restrict__ rvec,
                                                     const double* __restrict__ Minv, double* __restrict__ partials,
                                                     int64_t vstride, int64_t mstride) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = lane < 63 ? lane / 7 : 8;
  const int r = lane < 63 ? lane % 7 : 6;  // lane 63 shadows lane 62 (valid addresses, no stores)
  const bool live = lane < 63;
  const int gid = wave * 9 + g;
  int rA = 0, rB = 0;
  if (gid < ngroups) {
    rA = grow[gid];
    rB = grow[gid + 1];
  }
  int k = rowptr[rA];
  const int kend = rowptr[rB];
  auto clampk = [&](int kk) { kk = kk < kend ? kk : kend - 1; return kk < 0 ? 0 : (kk < nnzb ? kk : nnzb - 1); };
  auto loadA = [&](int kk, float* a) {
    const float* vp = vals + (size_t)49 * kk;
    const rl_f4* p0 = reinterpret_cast<const rl_f4*>(vp + 4 * r);
    const rl_f3* p1 = reinterpret_cast<const rl_f3*>(vp + 28 + 3 * r);
    const rl_f4 t0 = RL_NT ? __builtin_nontemporal_load(p0) : *p0;
    const rl_f3 t1 = RL_NT ? __builtin_nontemporal_load(p1) : *p1;
    a[0] = t0.x; a[1] = t0.y; a[2] = t0.z; a[3] = t0.w; a[4] = t1.x; a[5] = t1.y; a[6] = t1.z;
  };
  auto loadX = [&](int col, double (*x)[7]) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const double* xp = p + (size_t)s * vstride + (size_t)7 * col;
      const rl_d2 t0 = *reinterpret_cast<const rl_d2*>(xp);
      const rl_d2 t1 = *reinterpret_cast<const rl_d2*>(xp + 2);
      const rl_d2 t2 = *reinterpret_cast<const rl_d2*>(xp + 4);
      x[s][0] = t0.x; x[s][1] = t0.y; x[s][2] = t1.x; x[s][3] = t1.y; x[s][4] = t2.x; x[s][5] = t2.y;
      x[s][6] = xp[6];
    }
  };
  float a[7], an[7];
  double x[K][7], xn[K][7];
  double xe[K], xen[K];  // RL_XLDS: this lane's entry of the input at the current / next block's column
  __shared__ double xs[4][K][72];  // per wavefront and system: nine groups x 8 doubles (seven used)
  double (*xw)[72] = xs[threadIdx.x >> 6];
  auto loadXe = [&](int col, double* e) {
#pragma unroll
    for (int s = 0; s < K; ++s) e[s] = p[(size_t)s * vstride + (size_t)7 * col + r];
  };
  auto spread = [&](const double* e, double (*xo)[7]) {  // (one wavefront: LDS operations complete in order)
#pragma unroll
    for (int s = 0; s < K; ++s) xw[s][8 * g + r] = e[s];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const rl_d2* xp = reinterpret_cast<const rl_d2*>(&xw[s][8 * g]);
      const rl_d2 t0 = xp[0], t1 = xp[1], t2 = xp[2];
      xo[s][0] = t0.x; xo[s][1] = t0.y; xo[s][2] = t1.x; xo[s][3] = t1.y; xo[s][4] = t2.x; xo[s][5] = t2.y;
      xo[s][6] = xw[s][8 * g + 6];
    }
  };
  double acc[K], rv[K], pi[K], pr[K];
#pragma unroll
  for (int s = 0; s < K; ++s) acc[s] = rv[s] = pi[s] = pr[s] = xe[s] = xen[s] = 0.0;
  int row = rA;
  int k1 = rA < rB ? rowptr[rA + 1] : k;
  int k2 = rA + 2 <= rB ? rowptr[rA + 2] : kend;
  auto row_begin = [&](int rw) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      rv[s] = rvec[(size_t)s * vstride + (size_t)7 * rw + r];
      if (MODE == 2) pi[s] = p[(size_t)s * vstride + (size_t)7 * rw + r];
      acc[s] = 0.0;
    }
  };
  auto row_end = [&](int rw) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const double d = rv[s] - acc[s];
      if (MODE == 1) {
        if (live) q[(size_t)s * vstride + (size_t)7 * rw + r] = d;
      } else {
        const double* mp = Minv + (size_t)s * mstride + (size_t)49 * rw + 7 * r;  // symmetric: row r
        double o = 0.0;
#pragma unroll
        for (int c = 0; c < 7; ++c) o += mp[c] * __shfl(d, 7 * g + c);
        const double z = pi[s] + o;
        if (live) {
          q[(size_t)s * vstride + (size_t)7 * rw + r] = z;
          pr[s] += rv[s] * z;
        }
      }
    }
  };
  if (k < kend) {
    loadA(k, a);
    if (RL_XLDS) loadXe(colidx[k], xe); else loadX(colidx[k], x);
    row_begin(row);
  }
  int cn = colidx[clampk(k + 1)];
  while (__any(k < kend)) {
    const bool act = k < kend;
    const int cn2 = colidx[clampk(k + 2)];
    loadA(clampk(k + 1), an);
    if (RL_XLDS) {
      loadXe(cn, xen);
      spread(xe, x);  // (all lanes: the LDS round trip is wave-wide)
    } else {
      loadX(cn, xn);
    }
    if (act) {
#pragma unroll
      for (int s = 0; s < K; ++s) {
        double t = acc[s];
#pragma unroll
        for (int c = 0; c < 7; ++c) t += (double)a[c] * x[s][c];
        acc[s] = t;
      }
      ++k;
      if (k == k1) {
        row_end(row);
        ++row;
        k1 = k2;
        k2 = row + 2 <= rB ? rowptr[row + 2] : kend;
        if (row < rB) row_begin(row);
      }
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) a[c] = an[c];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      xe[s] = xen[s];
      if (!RL_XLDS) {
#pragma unroll
        for (int c = 0; c < 7; ++c) x[s][c] = xn[s][c];
      }
    }
    cn = cn2;
  }
  if (MODE == 2 && partials) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const double t = wave_sum(pr[s]);
      if (lane == 0) partials[(size_t)s * gridDim.x * 4 + wave] = t;
    }
  }
}
