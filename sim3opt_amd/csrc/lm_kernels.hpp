// lm_kernels.hpp -- device side of the LM outer loop (included by engine.hip after DevScalars, inside
// namespace sim3opt): fixed-order reductions, per-edge residuals / chi2 (EdgeSim3::computeError), the
// numeric-Jacobian linearisation with its Gram phase (BaseBinaryEdge::linearizeOplus +
// constructQuadraticForm), the per-row reduction of the diagonal contributions, the 7x7 block-Jacobi
// inverses and the FP32 block copies.  Reference call sites: kitti_surf.cpp:674-675 (everything here is
// reached from optimizer.optimize(100)); SURVEY.md 8(a) rows a5-a8.
#pragma once
__global__ __launch_bounds__(WG) void k_final_sum(const double* __restrict__ partials, int n,
                                                  double* __restrict__ out) {
  __shared__ double sh[4];
  const double s = sum_partials(partials, n, sh);
  if (threadIdx.x == 0) *out = s;
}

// after k_diag_reduce: trace(H) (summed as k_final_sum sums it) and max |H_dd| over the workgroups' maxima
__global__ __launch_bounds__(WG) void k_final_trace_max(const double* __restrict__ trace_partials,
                                                        const double* __restrict__ max_partials, int n,
                                                        double* __restrict__ trace_out,
                                                        unsigned long long* __restrict__ maxdiag_bits) {
  __shared__ double sh[4];
  __shared__ double shm[4];
  const double s = sum_partials(trace_partials, n, sh);
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += WG) m = fmax(m, max_partials[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    *trace_out = s;
    m = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
    *maxdiag_bits = (unsigned long long)__double_as_longlong(m);  // non-negative doubles order like their bits
  }
}

// chi2 and the step's scale delta^T (lambda delta + b) of an LM trial: two partial arrays of different lengths,
// each summed exactly as k_final_sum sums it, in one launch (outa and outb are adjacent in DevScalars)
// With a pinned host mirror the three scalars the host decides a trial on (chi2, scale, the exact solve's
// verdict) are written there as well: the host then only waits for the stream -- no copy command, whose
// launch and marker are 9 us of a 215-us KITTI-00 trial.
__global__ __launch_bounds__(WG) void k_final_sum_two(const double* __restrict__ pa, int na,
                                                      double* __restrict__ outa,
                                                      const double* __restrict__ pb, int nb_,
                                                      double* __restrict__ outb,
                                                      DevScalars* __restrict__ host_mirror,
                                                      const DevScalars* __restrict__ dev_sc) {
  __shared__ double sh[4];
  const double a = sum_partials(pa, na, sh);
  const double b = sum_partials(pb, nb_, sh);
  if (threadIdx.x == 0) {
    *outa = a;
    *outb = b;
    if (host_mirror) {
      host_mirror->chi2 = a;
      host_mirror->scale = b;
      host_mirror->fail = dev_sc->fail;
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-edge residual kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double quad_form(const double e[7], const double* __restrict__ Om) {
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    double col = 0.0;
#pragma unroll
    for (int r = 0; r < 7; ++r) col += e[r] * Om[7 * c + r];
    acc += col * e[c];
  }
  return acc;
}

// g2o RobustKernelHuber: rho(e2) and rho'(e2)
__device__ __forceinline__ void huber(double e2, double delta, double& rho, double& w) {
  const double dsqr = delta * delta;
  if (e2 <= dsqr) {
    rho = e2;
    w = 1.0;
  } else {
    const double sq = sqrt(e2);
    rho = 2 * sq * delta - dsqr;
    w = delta / sq;
  }
}


struct EdgeArgs {
  int32_t e_lo, e_hi;  // edge range evaluated by this launch (rank's share in multi-GPU chi2)
  const int32_t* ev0;
  const int32_t* ev1;
  const Sim3* meas;
  const double* info;    // nullptr: identity
  const double* kdelta;  // nullptr: no robust kernel
  const Sim3* states;
  sim3::Opts opts;
};

// computeActiveErrors + activeRobustChi2: one lane per edge, block partials in fixed order.
__global__ __launch_bounds__(WG) void k_chi2(EdgeArgs A, double* __restrict__ partials) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int k = A.e_lo + blockIdx.x * WG + threadIdx.x; k < A.e_hi; k += gridDim.x * WG) {
    const Sim3 C = load_sim3(A.meas + k);
    const Sim3 S0 = load_sim3(A.states + A.ev0[k]);
    const Sim3 S1 = load_sim3(A.states + A.ev1[k]);
    double e[7];
    sim3::edge_error(C, S0, S1, A.opts, e);
    double chi;
    if (A.info) {
      chi = quad_form(e, A.info + (size_t)49 * k);
    } else {
      chi = 0.0;
#pragma unroll
      for (int r = 0; r < 7; ++r) chi += e[r] * e[r];
    }
    if (A.kdelta && A.kdelta[k] > 0.0) {
      double rho, w;
      huber(chi, A.kdelta[k], rho, w);
      chi = rho;
    }
    acc += chi;
  }
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(WG) void k_edge_errors(EdgeArgs A, double* __restrict__ out) {
  for (int k = A.e_lo + blockIdx.x * WG + threadIdx.x; k < A.e_hi; k += gridDim.x * WG) {
    const Sim3 C = load_sim3(A.meas + k);
    const Sim3 S0 = load_sim3(A.states + A.ev0[k]);
    const Sim3 S1 = load_sim3(A.states + A.ev1[k]);
    double e[7];
    sim3::edge_error(C, S0, S1, A.opts, e);
#pragma unroll
    for (int r = 0; r < 7; ++r) out[(size_t)7 * k + r] = e[r];
  }
}

// ------------------------------------------------------------------------------------------
// linearisation, numeric Jacobians (g2o default for EdgeSim3)
//   half-wavefront (32 lanes) per edge, 8 edges per 256-thread workgroup:
//     lanes 0-13 : e(exp(+-delta e_d) S0, S1), d = lane/2        -> columns of A
//     lanes 14-27: e(S0, exp(+-delta e_d) S1)                     -> columns of B
//     lane 28    : unperturbed e
//   then the 14x14 Gram matrix J^T W J and -J^T W e are formed from LDS and written with plain
//   stores: off-diagonal 7x7 blocks straight into the block-CSR values (this edge owns them),
//   diagonal contributions into the per-incidence scratch reduced by k_diag_reduce.
// ------------------------------------------------------------------------------------------
struct LinArgs {
  int32_t n_active;
  const int32_t* active;
  const int32_t* ev0;
  const int32_t* ev1;
  const Sim3* meas;
  const double* info;
  const double* kdelta;
  const Sim3* states;
  const int32_t* slot01;
  const int32_t* slot10;
  const int32_t* inc0;
  const int32_t* inc1;
  double* vals;
  double* scratch;
  double delta;
  sim3::Opts opts;
  const Sim3* ptab;  // exp(+-delta e_d), entry 2 d + (0: +, 1: -)  (k_perturbation_table)
  int32_t dof_mask;  // cleared bit d: Jacobian column d of both endpoints is zero (frozen DoF)
  DevScalars* sc;    // max |H_dd| starts from zero here (k_diag_reduce, the next launch, raises it)
};

struct GramTables {
  unsigned char ga[119], gb[119];  // Gram tasks: (a, b) with a <= b < 14, or b == 14 for J^T W e
  unsigned char tr[28], tc[28];    // upper triangle (r <= c) in column-major order
};
__constant__ GramTables c_tab;

constexpr int EPB = 8;  // edges per workgroup

// The 14 perturbations exp(+-delta e_d) of the numeric Jacobians are the same for every edge: evaluated
// once (the same sim3::exp on the same arguments: bit-identical to evaluating it per lane) instead of
// 28 times per edge and linearisation -- a quarter of k_linearize_numeric's arithmetic.
__global__ __launch_bounds__(64) void k_perturbation_table(double delta, sim3::Opts opts, Sim3* __restrict__ tab) {
  const int t = threadIdx.x;
  if (t >= 14) return;
  const int d = t >> 1;
  const double step = (t & 1) ? -delta : delta;
  double xi[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) xi[i] = (i == d) ? step : 0.0;
  tab[t] = sim3::exp(xi, opts);
}

template <bool HAS_INFO, bool HAS_KERNEL>
__global__ __launch_bounds__(WG) void k_linearize_numeric(LinArgs A) {
  __shared__ double s_in[EPB][24];
  __shared__ double s_J[EPB][15][7];                    // 14 Jacobian columns, column 14 = e
  __shared__ double s_O[HAS_INFO ? EPB : 1][15][7];     // Omega * (J | e)
  __shared__ double s_G[EPB][14][15];                   // upper Gram + column 14 = -J^T W e
  const int l = threadIdx.x & 31, es = threadIdx.x >> 5;
  const int ai = blockIdx.x * EPB + es;
  const bool valid = ai < A.n_active;
  if (blockIdx.x == 0 && threadIdx.x == 0) A.sc->maxdiag_bits = 0ull;  // (instead of a memset: 18 us of host latency)
  int edge = 0;
  if (valid) {
    edge = A.active[ai];
    // coalesced 64-B reads: measurement and the two vertex states, staged in LDS
    if (l < 8) s_in[es][l] = reinterpret_cast<const double*>(A.meas + edge)[l];
    else if (l < 16) s_in[es][l] = reinterpret_cast<const double*>(A.states + A.ev0[edge])[l - 8];
    else if (l < 24) s_in[es][l] = reinterpret_cast<const double*>(A.states + A.ev1[edge])[l - 16];
  }
  __syncthreads();
  double e[7] = {0, 0, 0, 0, 0, 0, 0};
  if (valid && l < 29) {
    Sim3 C, S0, S1;
    const double* in = s_in[es];
    C.q[0] = in[0]; C.q[1] = in[1]; C.q[2] = in[2]; C.q[3] = in[3];
    C.t[0] = in[4]; C.t[1] = in[5]; C.t[2] = in[6]; C.s = in[7];
    S0.q[0] = in[8]; S0.q[1] = in[9]; S0.q[2] = in[10]; S0.q[3] = in[11];
    S0.t[0] = in[12]; S0.t[1] = in[13]; S0.t[2] = in[14]; S0.s = in[15];
    S1.q[0] = in[16]; S1.q[1] = in[17]; S1.q[2] = in[18]; S1.q[3] = in[19];
    S1.t[0] = in[20]; S1.t[1] = in[21]; S1.t[2] = in[22]; S1.s = in[23];
    if (l < 28) {
      const Sim3 P = A.ptab[l % 14];  // direction (l % 14) / 2, sign by the parity of l
      if (l < 14) S0 = sim3::mul(P, S0);
      else S1 = sim3::mul(P, S1);
    }
    sim3::edge_error(C, S0, S1, A.opts, e);
  }
  const double scalar = 1.0 / (2.0 * A.delta);
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const double other = __shfl_down(e[r], 1);  // lane l+1 holds the -delta evaluation
    if (valid && l < 28 && !(l & 1))
      s_J[es][l >> 1][r] = ((A.dof_mask >> ((l % 14) >> 1)) & 1) ? scalar * (e[r] - other) : 0.0;
    if (valid && l == 28) s_J[es][14][r] = e[r];
  }
  __syncthreads();
  if (HAS_INFO) {
    if (valid) {
      const double* Om = A.info + (size_t)49 * edge;  // column-major
      for (int t = l; t < 105; t += 32) {
        const int a = t / 7, r = t % 7;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += Om[7 * k + r] * s_J[es][a][k];
        s_O[es][a][r] = acc;
      }
    }
    __syncthreads();
  }
  double (*OJ)[7] = HAS_INFO ? s_O[es] : s_J[es];
  double w = 1.0;
  if (HAS_KERNEL) {
    if (valid && A.kdelta[edge] > 0.0) {
      double chi = 0.0, rho;
#pragma unroll
      for (int r = 0; r < 7; ++r) chi += s_J[es][14][r] * OJ[14][r];
      huber(chi, A.kdelta[edge], rho, w);
    }
  }
  if (valid) {
    for (int t = l; t < 119; t += 32) {
      const int a = c_tab.ga[t], b = c_tab.gb[t];
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 7; ++r) acc += s_J[es][a][r] * OJ[b][r];
      s_G[es][a][b] = (b == 14 ? -w : w) * acc;
    }
  }
  __syncthreads();
  if (valid) {
    const int s01 = A.slot01[edge], s10 = A.slot10[edge];
    const int i0 = A.inc0[edge], i1 = A.inc1[edge];
    double (*G)[15] = s_G[es];
    for (int o = l; o < 168; o += 32) {
      if (o < 49) {  // H01 = A^T W B, column-major
        if (s01 >= 0) A.vals[(size_t)49 * s01 + o] = G[o % 7][7 + o / 7];
      } else if (o < 98) {  // H10 = H01^T
        const int p = o - 49;
        if (s10 >= 0) A.vals[(size_t)49 * s10 + p] = G[p / 7][7 + p % 7];
      } else if (o < 133) {  // endpoint 0: upper(A^T W A), -A^T W e
        const int t = o - 98;
        if (i0 >= 0)
          A.scratch[(size_t)35 * i0 + t] = t < 28 ? G[c_tab.tr[t]][c_tab.tc[t]] : G[t - 28][14];
      } else {  // endpoint 1
        const int t = o - 133;
        if (i1 >= 0)
          A.scratch[(size_t)35 * i1 + t] =
              t < 28 ? G[7 + c_tab.tr[t]][7 + c_tab.tc[t]] : G[7 + t - 28][14];
      }
    }
  }
}

// One wavefront per block row: sums the per-incidence contributions in edge order, writes the
// full symmetric diagonal block and b, tracks max |H_dd| (computeLambdaInit).
__global__ __launch_bounds__(WG) void k_diag_reduce(int r0, int r1,
                                                    const int32_t* __restrict__ incptr,
                                                    const int32_t* __restrict__ rowptr,
                                                    const double* __restrict__ scratch,
                                                    double* __restrict__ vals,
                                                    double* __restrict__ b, DevScalars* sc,
                                                    double* __restrict__ trace_partials,
                                                    double* __restrict__ max_partials) {
  __shared__ double sh[4];
  __shared__ double shm[4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane % 7, c = lane / 7;
  const int m = r < c ? r : c, M = r < c ? c : r;
  const int tsrc = lane < 49 ? (M * (M + 1)) / 2 + m : 0;
  const int bsrc = lane < 7 ? 28 + lane : 0;
  double dmax = 0.0, tr = 0.0;
  for (int row = r0 + blockIdx.x * 4 + wave; row < r1; row += gridDim.x * 4) {
    const int k0 = incptr[row], k1 = incptr[row + 1];
    double sum = 0.0;
    if (lane < 35)
      for (int k = k0; k < k1; ++k) sum += scratch[(size_t)35 * k + lane];
    const double v = __shfl(sum, tsrc);
    const double bv = __shfl(sum, bsrc);
    if (lane < 49) {
      vals[(size_t)49 * rowptr[row] + lane] = v;
      if (r == c) {
        dmax = fmax(dmax, fabs(v));
        tr += v;
      }
    }
    if (lane < 7) b[(size_t)7 * row + lane] = bv;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
  // (one atomicMax per wavefront on sc->maxdiag_bits -- 8192 of them on one address -- was 80 of this kernel's
  // 98 us on a 10k-vertex graph and a third of its 273 us on config 3: per-workgroup maxima now, reduced by
  // k_final_trace_max together with the trace)
  if (lane == 0) shm[wave] = dmax;
  const double ts = block_sum(tr, sh);  // fixed order: deterministic (its barriers also publish shm)
  if (threadIdx.x == 0) {
    trace_partials[blockIdx.x] = ts;
    max_partials[blockIdx.x] = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
  }
}

// ------------------------------------------------------------------------------------------
// update and scale (oplusImpl, push / pop, computeScale)
// ------------------------------------------------------------------------------------------
// VertexSim3Expmap::oplusImpl: S <- exp(dx) * S for every free vertex
// (sc != nullptr: the exact factorisation reports a non-positive pivot through sc->fail after the
// fact -- it stores the solve's token there, so that nobody has to reset the flag between solves --;
// the step is then garbage and must not be applied -- the host rejects the trial)
// `backup` (may be null) receives the estimates as they were: g2o's push() without a copy of its own.
__global__ __launch_bounds__(WG) void k_oplus(int nv, const int32_t* __restrict__ hidx,
                                              const double* __restrict__ x, Sim3* states,
                                              sim3::Opts opts, const DevScalars* sc, Sim3* backup,
                                              int fail_token) {
  const int v = blockIdx.x * WG + threadIdx.x;
  if (v >= nv) return;
  if (backup) {
    const double* s8 = reinterpret_cast<const double*>(states + v);
    double* b8 = reinterpret_cast<double*>(backup + v);
#pragma unroll
    for (int i = 0; i < 8; ++i) b8[i] = s8[i];
  }
  if (sc && sc->fail == fail_token) return;
  const int h = hidx[v];
  if (h < 0) return;
  double xi[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) xi[i] = x[(size_t)7 * h + i];
  const Sim3 P = sim3::exp(xi, opts);
  const Sim3 S = sim3::mul(P, load_sim3(states + v));
  double* d = reinterpret_cast<double*>(states + v);
  d[0] = S.q[0]; d[1] = S.q[1]; d[2] = S.q[2]; d[3] = S.q[3];
  d[4] = S.t[0]; d[5] = S.t[1]; d[6] = S.t[2]; d[7] = S.s;
}

// pop(): the estimates of a rejected trial go back (a kernel: hipMemcpyAsync costs the host 6-18 us)
__global__ __launch_bounds__(WG) void k_copy_states(int nv, const Sim3* __restrict__ src, Sim3* __restrict__ dst) {
  const int i = blockIdx.x * WG + threadIdx.x;
  if (i < 8 * nv) reinterpret_cast<double*>(dst)[i] = reinterpret_cast<const double*>(src)[i];
}

// computeScale: sum_j x_j (lambda x_j + b_j)
__global__ __launch_bounds__(WG) void k_scale(int j0, int j1, const double* __restrict__ x,
                                              const double* __restrict__ b, double lambda,
                                              double* __restrict__ partials) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int j = j0 + blockIdx.x * WG + threadIdx.x; j < j1; j += gridDim.x * WG)
    acc += x[j] * (lambda * x[j] + b[j]);
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// debug_full_arrays: 32-bit words of [p, p + nwords) that are no longer the 0xFF poison
__global__ __launch_bounds__(WG) void k_count_unpoisoned(const uint32_t* __restrict__ p, size_t nwords,
                                                         unsigned long long* __restrict__ count) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < nwords; i += (size_t)gridDim.x * WG)
    c += p[i] != 0xFFFFFFFFu;
  if (c) atomicAdd(count, c);
}
