// engine_amg.hip -- aggregation-multigrid preconditioner of the PCG: numbers per linearisation / per LM trial, the cycle
#include "engine_impl.hpp"

namespace sim3opt {

#include "spmv_kernel.hpp"
#include "amg_kernels.hpp"

int Engine::amg_init(const Structure& s, bool automatic, std::string& err) {
  (void)err;
  amg_omega = std::max(0.1, std::min(0.95, opt.amg_omega));
  // cycle (measured): multiplicative on level 0; level 1 twice and deeper levels three times
  // per visit since the cycle's matrix passes stream FP32 copies (round 2).  When level 0 is partitioned
  // over >= 4 ranks what is left of a PCG iteration is the latency-bound coarse cycle and its exchanges:
  // level 1 once, deeper levels twice -- half the coarse launches for about 1.5x the iterations, which
  // loses on one GPU and wins there.  The additive level-0 form is a knob (about as fast on
  // config 3, less robust on ill-conditioned chains).
  amg_additive = opt.amg_additive != 0;
  {
    const bool sharded4 = part_world() >= 4;
    int last = 0;
    for (int l = 1; l <= AMG_MAX_LEVELS; ++l) {
      const int o = l <= 4 ? opt.amg_cycle[l - 1] : 0;
      if (o >= 1 && o <= 3) last = o;
      amg_visits[l] = last > 0 ? last : (sharded4 ? (l <= 1 ? 1 : 2) : (l <= 1 ? 2 : 3));
    }
    amg_visits[0] = 1;
  }
  amg_fp32 = opt.amg_fp32 != 0;
  adaptive_prec = opt.adaptive_prec != 0 && automatic;  // (a caller who names the multigrid gets the multigrid)
  amg_pivot = opt.amg_pivot >= 28 ? 28 : 14;
  // measured on config 3: 1.8 into level 0 and 1.6 below cut the PCG iterations of
  // a solve from 56 to 43 (cycle 1/3) and from 29 to 25 (cycle 2/3); 2.0 (the limit for an exact
  // coarse solve) is no better
  for (int l = 0; l <= AMG_MAX_LEVELS; ++l)
    amg_over_l[l] = std::max(0.5, std::min(3.0, l == 0 ? opt.amg_over[0] : opt.amg_over[1]));
  std::string why;
  if (!build_amg_hierarchy(nb, s.rowptr.data(), s.colidx.data(), amg_host, why)) {
    if (opt.verbose) std::fprintf(stderr, "sim3opt: no multigrid hierarchy (%s)\n", why.c_str());
    amg_host.clear();
    return SIM3OPT_OK;
  }
  if (automatic && (double)amg_host[1].nnzb > 0.3 * (double)amg_host[0].nnzb) {
    if (opt.verbose)
      std::fprintf(stderr, "sim3opt: the graph coarsens like an expander (level-1 blocks %.2f of level 0): block-Jacobi\n",
                   (double)amg_host[1].nnzb / (double)amg_host[0].nnzb);
    amg_host.clear();
    return SIM3OPT_OK;
  }
  use_amg = true;
  return SIM3OPT_OK;
}

int Engine::amg_bind(const Structure& s, std::string& err) {
  std::vector<AmgLevelHost>& H = amg_host;
  const int nl = (int)H.size();
  amg.assign(nl, AmgLevel());
  int rc = SIM3OPT_OK;
#define AMGCHK(call) do { rc = (call); if (rc) return rc; } while (0)
  AmgLevel& L0 = amg[0];
  L0.nb = nb; L0.nnzb = nnzb;
  L0.rowptr = d_rowptr; L0.colidx = d_colidx; L0.wrow = d_wrow; L0.span_grid = span_grid;
  L0.vals = d_vals; L0.Minv = d_Minv; L0.r = d_r; L0.x = d_z;
  // (padded like the PCG vectors: the multi-GPU all-gather runs in place with equal counts)
  int64_t padded = 0;
  (void)allgather_equal_plan(offs.data(), comm.world, nullptr, &padded);
  AMGCHK(amg_alloc(d_az, std::max<size_t>((size_t)n, (size_t)padded), err));
  AMGCHK(amg_alloc(d_P, (size_t)49 * nb, err));
  AMGCHK(amg_up(d_row2v, s.row2vertex, err));
  L0.t = d_az;
  for (int l = 0; l < nl; ++l) {
    AmgLevel& L = amg[l];
    const AmgLevelHost& h = H[l];
    if (l > 0) {
      L.nb = h.nb; L.nnzb = h.nnzb;
      AMGCHK(amg_up(L.rowptr, h.rowptr, err));
      AMGCHK(amg_up(L.colidx, h.colidx, err));
      // coarse levels are latency-bound, not bandwidth-bound: one block row per wavefront
      L.span_grid = std::max(1, (L.nb + 3) / 4);
      std::vector<int32_t> wrow(L.span_grid * 4 + 1);
      partition_rows(L.nb, h.rowptr.data(), L.span_grid * 4, wrow.data());
      AMGCHK(amg_up(L.wrow, wrow, err));
      AMGCHK(amg_alloc(L.vals, (size_t)49 * L.nnzb, err));
      AMGCHK(amg_alloc(L.diagH, (size_t)49 * L.nb, err));
      AMGCHK(amg_alloc(L.W, (size_t)49 * L.nb, err));
      AMGCHK(amg_alloc(L.Minv, (size_t)49 * L.nb, err));
      AMGCHK(amg_alloc(L.r, (size_t)7 * L.nb, err));
      AMGCHK(amg_alloc(L.x, (size_t)7 * L.nb, err));
      AMGCHK(amg_alloc(L.t, (size_t)7 * L.nb, err));
    }
    if (amg_fp32) {
      const size_t n32 = (size_t)98 * (size_t)((std::max<int64_t>(L.nnzb, 1) + 1) / 2);  // whole pairs
      HIPCHK(dev_malloc((void**)&L.vals32, sizeof(float) * n32));
      HIPCHK(hipMemset(L.vals32, 0, sizeof(float) * n32));
      amg_owned.push_back(L.vals32);
    }
    if (l + 1 < nl) {
      AMGCHK(amg_up(L.agg, h.agg, err));
      AMGCHK(amg_up(L.mptr, h.mptr, err));
      AMGCHK(amg_up(L.mem, h.mem, err));
      AMGCHK(amg_up(L.gptr, h.gptr, err));
      AMGCHK(amg_up(L.gblk, h.gblk, err));
      AMGCHK(amg_up(L.grow, h.grow, err));
    }
  }
  const size_t nc = (size_t)7 * amg[nl - 1].nb;
  AMGCHK(amg_alloc(d_Ainv, nc * nc, err));
  AMGCHK(amg_alloc(d_Ainv2, nc * nc, err));
  AMGCHK(amg_alloc(d_piv, 2 * 28 * 28, err));  // pivot-block inverses handed from step to step
#undef AMGCHK
  if (opt.verbose) {
    std::fprintf(stderr, "sim3opt: multigrid levels (rows/blocks):");
    for (const AmgLevel& L : amg) std::fprintf(stderr, " %d/%lld", L.nb, (long long)L.nnzb);
    std::fprintf(stderr, "\n");
  }
  amg_host.clear();
  amg_host.shrink_to_fit();
  amg_stale = true;
  return SIM3OPT_OK;
}

// numbers of the hierarchy: once per linearisation (P = Ad(S_v) at the linearisation point)
int Engine::amg_setup(std::string& err) {
  const int nl = (int)amg.size();
  hipLaunchKernelGGL(k_amg_adjoint, dim3(grid_for(nb, WG)), dim3(WG), 0, stream, nb, d_row2v,
                     d_states, d_P);
  for (int l = 0; l + 1 < nl; ++l) {
    const AmgLevel& F = amg[l];
    AmgLevel& Cc = amg[l + 1];
    const int gg = (int)((Cc.nnzb + 3) / 4), gw = (Cc.nb + 3) / 4;
    if (l == 0) {
      hipLaunchKernelGGL((k_amg_galerkin<true>), dim3(gg), dim3(WG), 0, stream, (int)Cc.nnzb, F.gptr,
                         F.gblk, F.grow, F.colidx, F.vals, d_P, Cc.vals, amg_fp32 ? F.vals32 : (float*)nullptr);
      if (comm.active()) {
        // a rank holds the blocks of its own rows (the others are zero): the products above are
        // partial sums; summed over the ranks, level 1 and everything below is replicated
        int rc = comm.allreduce(Cc.vals, (int)(49 * Cc.nnzb), 0, stream, err);
        if (rc) return rc;
      }
      hipLaunchKernelGGL((k_amg_wsum<true>), dim3(gw), dim3(WG), 0, stream, Cc.nb, F.mptr, F.mem,
                         d_P, Cc.W);
    } else {
      hipLaunchKernelGGL((k_amg_galerkin<false>), dim3(gg), dim3(WG), 0, stream, (int)Cc.nnzb, F.gptr,
                         F.gblk, F.grow, F.colidx, F.vals, (const double*)nullptr, Cc.vals,
                         amg_fp32 ? F.vals32 : (float*)nullptr);
      hipLaunchKernelGGL((k_amg_wsum<false>), dim3(gw), dim3(WG), 0, stream, Cc.nb, F.mptr, F.mem,
                         F.W, Cc.W);
    }
    hipLaunchKernelGGL(k_amg_copydiag, dim3(grid_for(49 * (int64_t)Cc.nb, WG)), dim3(WG), 0, stream,
                       Cc.nb, Cc.rowptr, Cc.vals, Cc.diagH);
  }
  if (amg_fp32)  // (the Galerkin products wrote the FP32 copies of the levels they read)
    for (int l = nl - 1; l < nl; ++l) {
      const size_t cnt = (size_t)49 * (size_t)amg[l].nnzb;
      hipLaunchKernelGGL(k_to_f32, dim3(grid_for((int64_t)(cnt / 4), WG)), dim3(WG), 0, stream, cnt,
                         (const double*)amg[l].vals, amg[l].vals32);
    }
  HIPCHK(hipGetLastError());
  amg_stale = false;
  return SIM3OPT_OK;
}

// per trial: damped diagonal blocks, smoother inverses, dense inverse of the coarsest level
void Engine::amg_prepare(double lambda) {
  const int nl = (int)amg.size();
  for (int l = 0; l < nl; ++l) {
    const AmgLevel& L = amg[l];
    const int lo = l == 0 ? r0 : 0, hi = l == 0 ? r1 : L.nb;  // level 0 is row-partitioned
    jacobi(lo, hi, L.rowptr, L.vals, lambda, L.Minv, l == 0 && amg_additive ? 1.0 : amg_omega, L.diagH, L.W,
           l > 0 ? L.vals32 : (float*)nullptr);
  }
  // dense inverse of the coarsest level: one launch per 14-row pivot block, buffers ping-pong
  const AmgLevel& Lc = amg[nl - 1];
  const int nd = 7 * Lc.nb;
  // pivot blocks of `amg_pivot` rows (14: 82 launches for 1141 unknowns), then 14, then 7 for the tail; the
  // buffers ping-pong and the last step must write d_Ainv, which fixes the buffer the matrix is filled into
  // (round 3: 32-row pivots inverted by the whole workgroup in LDS took 36 x 52 us -- the same 1.9 ms as 82 x
  // 23 us; profiles/r3_negative_results.log.  What did pay is taking the pivot inverse off each step's
  // critical path: k_amg_dense_gj_step's look-ahead workgroup)
  auto pivot_rows = [&](int k0) { return nd - k0 >= amg_pivot ? amg_pivot : (nd - k0 >= 14 ? 14 : 7); };
  int nsteps = 0;
  for (int k0 = 0; k0 < nd; k0 += pivot_rows(k0)) ++nsteps;
  double *src = nsteps % 2 ? d_Ainv2 : d_Ainv, *dst = nsteps % 2 ? d_Ainv : d_Ainv2;
  (void)hipMemsetAsync(src, 0, sizeof(double) * (size_t)nd * nd, stream);
  hipLaunchKernelGGL(k_amg_dense_fill, dim3(grid_for(49 * Lc.nnzb, WG)), dim3(WG), 0, stream, Lc.nb,
                     Lc.rowptr, Lc.colidx, Lc.vals, src);
  double *pin = d_piv, *pout = d_piv + 28 * 28;
  switch (pivot_rows(0)) {
    case 28: hipLaunchKernelGGL((k_amg_dense_gj_first<28>), dim3(1), dim3(64), 0, stream, nd, (const double*)src, pin, d_sc); break;
    case 14: hipLaunchKernelGGL((k_amg_dense_gj_first<14>), dim3(1), dim3(64), 0, stream, nd, (const double*)src, pin, d_sc); break;
    default: hipLaunchKernelGGL((k_amg_dense_gj_first<7>), dim3(1), dim3(64), 0, stream, nd, (const double*)src, pin, d_sc);
  }
  const dim3 gt((nd + 63) / 64, (nd + 63) / 64 + 1);  // row 0 of the grid: the look-ahead workgroup
  for (int k0 = 0; k0 < nd;) {
    const int pb = pivot_rows(k0), pbn = k0 + pb < nd ? pivot_rows(k0 + pb) : 0;
    if (pb == 28)
      hipLaunchKernelGGL((k_amg_dense_gj_step<28>), gt, dim3(WG), 0, stream, nd, k0, (const double*)src,
                         dst, (const double*)pin, pout, pbn, d_sc);
    else if (pb == 14)
      hipLaunchKernelGGL((k_amg_dense_gj_step<14>), gt, dim3(WG), 0, stream, nd, k0, (const double*)src,
                         dst, (const double*)pin, pout, pbn, d_sc);
    else
      hipLaunchKernelGGL((k_amg_dense_gj_step<7>), gt, dim3(WG), 0, stream, nd, k0, (const double*)src,
                         dst, (const double*)pin, pout, pbn, d_sc);
    k0 += pb;
    std::swap(src, dst);
    std::swap(pin, pout);
  }  // the inverse is in d_Ainv
}

// mode 3 (coarse levels): mode 2 on v + xc[agg], the coarser level's correction prolonged on the fly
void Engine::spmv_mode(const AmgLevel& L, int mode, int level, const double* v, double* out,
               const double* rvec, const double* xc) {
  // level 0 carries the damping as a scalar (read from DevScalars: capturable); coarse levels
  // have it inside their diagonal blocks.  Level 0 streams once (non-temporal), the rest is small.
  // Only level-0 launches test the `done` flag: on the latency-bound coarse levels that dependent
  // scalar load in front of the kernel costs more than the few idle launches after convergence.
  // level 0's smoothing pass is the cycle's last kernel: it writes z = M^-1 r and leaves the partials
  // of r.z for the PCG (multiplicative cycle only)
  double* const rz_part = level == 0 && mode == 2 && !amg_additive ? d_part_b : nullptr;
#define AMG_SPMV(NTV, MODEV)                                                                     \
hipLaunchKernelGGL((k_spmv_span<8, NTV, MODEV>), dim3(L.span_grid), dim3(WG), 0, stream, L.nb,  \
                   L.wrow, L.rowptr, L.colidx, L.vals, v, out, 0.0, rz_part, rvec,              \
                   const_cast<double*>(xc), level == 0 ? d_sc : (DevScalars*)nullptr, L.Minv, 1,   \
                   (const int32_t*)L.agg, amg_over)
#define AMG_SPMV32(NTV, MODEV)                                                                    \
hipLaunchKernelGGL((k_spmv_span<(NTV) ? SIM3OPT_F32_CH : 8, NTV, MODEV, float>), dim3(L.span_grid), dim3(WG), 0, stream,  \
                   L.nb, L.wrow, L.rowptr, L.colidx, (const float*)L.vals32, v, out, 0.0,         \
                   rz_part, rvec, const_cast<double*>(xc),                                        \
                   level == 0 ? d_sc : (DevScalars*)nullptr, L.Minv, 1, (const int32_t*)L.agg, amg_over)
  if (amg_fp32) {
    if (level == 0) { if (mode == 1) AMG_SPMV32(true, 1); else AMG_SPMV32(true, 2); }
    else { if (mode == 1) AMG_SPMV32(false, 1); else if (mode == 3) AMG_SPMV32(false, 3); else AMG_SPMV32(false, 2); }
  } else {
    if (level == 0) { if (mode == 1) AMG_SPMV(true, 1); else AMG_SPMV(true, 2); }
    else { if (mode == 1) AMG_SPMV(false, 1); else if (mode == 3) AMG_SPMV(false, 3); else AMG_SPMV(false, 2); }
  }
#undef AMG_SPMV32
#undef AMG_SPMV
}

void Engine::amg_restrict(int l, const double* t) {  // r_{l+1} = P^T t, x_{l+1} = Minv r_{l+1}
  const AmgLevel& F = amg[l];
  const AmgLevel& Cc = amg[l + 1];
  const int gr = grid_for((Cc.nb + 8) / 9, 4);
  const bool split = l == 0 && comm.active();  // level 0 is row-partitioned: partial sums
  const double* Minv_c = l + 2 < (int)amg.size() ? Cc.Minv : nullptr;  // coarsest: solved exactly
  if (l == 0)
    hipLaunchKernelGGL(k_amg_restrict0, dim3((Cc.nb + 3) / 4), dim3(WG), 0, stream, Cc.nb, F.mptr,
                       F.mem, d_P, t, Cc.r, split ? (const double*)nullptr : Minv_c, Cc.x,
                       (const DevScalars*)d_sc, r0, r1);
  else
    hipLaunchKernelGGL(k_amg_restrict, dim3(gr), dim3(WG), 0, stream, Cc.nb, F.mptr, F.mem, t, Cc.r,
                       Minv_c, Cc.x);
  if (split) {
    if (amg_status == SIM3OPT_OK) amg_status = comm.allreduce(Cc.r, 7 * Cc.nb, 0, stream, amg_err);
    if (Minv_c)
      hipLaunchKernelGGL(k_amg_bjapply, dim3(gr), dim3(WG), 0, stream, Cc.nb, Minv_c,
                         (const double*)Cc.r, Cc.x);
  }
}

void Engine::amg_prolong(int l, const double* xc, const double* xin, double* xout) {
  const AmgLevel& F = amg[l];
  const int gp = grid_for((F.nb + 8) / 9, 4);
  if (l == 0)
    hipLaunchKernelGGL((k_amg_prolong<true>), dim3(gp), dim3(WG), 0, stream, F.nb, F.agg, d_P, xc,
                       xin, xout, (const DevScalars*)d_sc, amg_over);
  else
    hipLaunchKernelGGL((k_amg_prolong<false>), dim3(gp), dim3(WG), 0, stream, F.nb, F.agg,
                       (const double*)nullptr, xc, xin, xout, (const DevScalars*)nullptr, amg_over);
}

// Solves the level-(l+1) problem approximately (right-hand side amg[l+1].r, first iterate
// amg[l+1].x = Minv r already there) by amg_visits[l+1] cycles; returns the buffer with the result.
const double* Engine::amg_coarse(int l) {
  const int nl = (int)amg.size();
  const AmgLevel& Cc = amg[l + 1];
  if (l + 2 == nl) {
    hipLaunchKernelGGL(k_amg_dense_apply, dim3(std::max(1, std::min(256, (7 * Cc.nb + 3) / 4))), dim3(WG),
                       0, stream, 7 * Cc.nb, d_Ainv, Cc.r, Cc.x, (const DevScalars*)nullptr);
    return Cc.x;
  }
  double* res = amg_cycle(l + 1, Cc.x, Cc.t);
  for (int g = 1; g < amg_visits[l + 1]; ++g) {  // W-cycle: again, from the current iterate
    double* oth = res == Cc.x ? Cc.t : Cc.x;
    spmv_mode(Cc, 2, l + 1, res, oth, Cc.r);  // pre-smoothing step
    res = amg_cycle(l + 1, oth, res);
  }
  return res;
}

// One multigrid cycle on level l from the iterate `cur`; `other` is scratch; returns the buffer
// that holds the new iterate (always `other`):
//   t = r - A cur;  coarse correction;  cur += P x_c;  other = cur + Minv (r - A cur)
double* Engine::amg_cycle(int l, double* cur, double* other) {
  const AmgLevel& F = amg[l];
  spmv_mode(F, 1, l, cur, other, F.r);
  amg_restrict(l, other);
  const double* xc = amg_coarse(l);
  amg_over = amg_over_on ? amg_over_l[l] : 1.0;
  if (l == 0) {
    amg_prolong(l, xc, cur, cur);
    spmv_mode(F, 2, l, cur, other, F.r);
  } else {  // piecewise-constant prolongation: added while the smoothing pass gathers its input
    spmv_mode(F, 3, l, cur, other, F.r, xc);
  }
  return other;
}

// d_az = M^-1 d_r; on entry d_z = Minv_0 d_r (written by the PCG step).  Multiplicative: one
// V(1,1) (or W) cycle from that iterate.  Additive on level 0 (no fine-level matrix pass in the
// preconditioner): M^-1 = D^-1 + P (coarse cycle) P^T.
// Multi-GPU: level 0 is row-partitioned like the PCG (its matrix passes need the whole iterate:
// one all-gather of d_z before, one of d_az after; the restricted residual is all-reduced), the
// coarse levels are replicated and every rank runs the same coarse cycle.
int Engine::amg_apply(std::string& err) {
  amg_status = SIM3OPT_OK;
  if (comm.active()) {
    int rc = exchange_rows(d_z, err);
    if (rc) return rc;
  }
  if (amg_additive) {
    amg_restrict(0, d_r);
    const double* xc0 = amg_coarse(0);
    amg_over = amg_over_on ? amg_over_l[0] : 1.0;
    amg_prolong(0, xc0, d_z, d_az);
  } else {
    amg_cycle(0, d_z, d_az);
  }
  if (amg_status != SIM3OPT_OK) {
    err = amg_err;
    return amg_status;
  }
  if (comm.active()) return exchange_rows(d_az, err);
  return SIM3OPT_OK;
}

}  // namespace sim3opt
