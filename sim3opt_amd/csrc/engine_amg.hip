// engine_amg.hip -- aggregation-multigrid preconditioner of the PCG: numbers per linearisation / per LM trial, the cycle
#include "engine_impl.hpp"

namespace sim3opt {

#include "spmv_kernel.hpp"
#include "amg_kernels.hpp"

int Engine::amg_init(const Structure& s, bool automatic, std::string& err) {
  (void)err;
  amg_omega = std::max(0.1, std::min(0.95, opt.amg_omega));
  // cycle (measured): multiplicative on level 0; level 1 twice and deeper levels three times
  // per visit since the cycle's matrix passes stream FP32 copies (round 2).  When level 0 is partitioned over
  // the ranks what is left of a PCG iteration is the latency-bound coarse cycle and its exchanges: level 1
  // once, deeper levels twice -- half the coarse launches and collectives for about 1.4x the iterations, which
  // loses on one GPU and wins on every partition (at two ranks too, once the Galerkin all-gather
  // of a replicated level 1 is counted).  The additive level-0 form is a knob (about as fast on config 3, less
  // robust on ill-conditioned chains).
  amg_additive = opt.amg_additive != 0;
  {
    const bool partitioned = part_world() >= 2;
    int last = 0;
    for (int l = 1; l <= AMG_MAX_LEVELS; ++l) {
      const int o = l <= 4 ? opt.amg_cycle[l - 1] : 0;
      if (o >= 1 && o <= 3) last = o;
      amg_visits[l] = last > 0 ? last : (partitioned ? (l <= 1 ? 1 : 2) : (l <= 1 ? 2 : 3));
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
  AmgBuildOptions bo;
  bo.max_coarsest = opt.amg_coarsest;
  for (int k = 0; k < 3; ++k) bo.passes[k] = opt.amg_passes[k];
  // the aggregation respects the rank partition (or, on one rank, the partition options.amg_virtual_ranks
  // names): aggregates never straddle two ranks
  std::vector<int32_t> vbegin;
  bo.world = part_world();
  bo.shard_rows = std::max(1, opt.amg_shard_rows);
  if (comm.world > 1) {
    bo.row_begin = row_begin.data();
  } else if (bo.world > 1) {
    vbegin.resize(bo.world + 1);
    partition_rows_equal(nb, bo.world, vbegin.data());
    bo.row_begin = vbegin.data();
  }
  if (!build_amg_hierarchy(nb, s.rowptr.data(), s.colidx.data(), amg_host, why, bo)) {
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
  L0.lo = r0; L0.hi = r1;
  // Which coarse levels are partitioned like level 0 (multi-rank runs): those with more rows than
  // options.amg_shard_rows, except the dense one -- a replicated level costs every rank its whole cycle, a
  // partitioned one costs an exchange per matrix pass; below a few thousand rows both are latency and the
  // replicated form needs no collective.  (Round 4 first kept level 1 of config 3 replicated below
  // four ranks -- threshold x 8 -- on a model that left out what a replicated level 1 costs per linearisation: the
  // all-gather of its 161 MB of Galerkin blocks; with it the partitioned form wins at two ranks as well.)
  const int64_t shard_rows = std::max(1, opt.amg_shard_rows);
  rep_level = 1;
  while (rep_level < nl - 1 && H[rep_level].nb > shard_rows && !H[rep_level].row_begin.empty() &&
         H[rep_level].respects_owner)
    ++rep_level;
  if (comm.active()) {
    parts.resize(rep_level);
    n_sharded = rep_level;
  }
  for (int l = 0; l < nl; ++l) {
    AmgLevel& L = amg[l];
    const AmgLevelHost& h = H[l];
    if (l > 0) {
      L.nb = h.nb; L.nnzb = h.nnzb;
      L.lo = 0; L.hi = L.nb;
      AMGCHK(amg_up(L.rowptr, h.rowptr, err));
      AMGCHK(amg_up(L.colidx, h.colidx, err));
      if (sharded(l)) {
        AMGCHK(level_part_init(l, L.nb, h.rowptr.data(), h.colidx.data(), h.row_begin, err));
        L.lo = parts[l].lo; L.hi = parts[l].hi;
      }
      // coarse levels are latency-bound, not bandwidth-bound: one block row per wavefront (of this rank's rows)
      const int nloc = L.hi - L.lo;
      L.span_grid = std::max(1, (nloc + 3) / 4);
      std::vector<int32_t> wrow(L.span_grid * 4 + 1);
      partition_rows(nloc, h.rowptr.data() + L.lo, L.span_grid * 4, wrow.data());
      for (int32_t& w : wrow) w += L.lo;
      AMGCHK(amg_up(L.wrow, wrow, err));
      // (a partitioned level's blocks: this rank's rows only -- alloc_ranged; replicated levels: all)
      AMGCHK(alloc_ranged(L.vals, 49 * (int64_t)h.rowptr[L.lo], 49 * (int64_t)h.rowptr[L.hi], 49 * (int64_t)L.nnzb, err));
      AMGCHK(amg_alloc(L.diagH, (size_t)49 * L.nb, err));
      AMGCHK(amg_alloc(L.W, (size_t)49 * L.nb, err));
      AMGCHK(amg_alloc(L.Minv, (size_t)49 * L.nb, err));
      AMGCHK(amg_alloc(L.r, (size_t)7 * L.nb, err));
      AMGCHK(amg_alloc(L.x, (size_t)7 * L.nb, err));
      AMGCHK(amg_alloc(L.t, (size_t)7 * L.nb, err));
    }
    if (amg_fp32) {
      // whole pairs of blocks (f32_pair_index); of a partitioned level the pairs that hold this rank's rows' blocks
      // (a span's first / last pair may reach one block into a neighbour's rows: loaded, skipped by the kernel)
      const int64_t b0 = l == 0 ? s.rowptr[L.lo] : h.rowptr[L.lo], b1 = l == 0 ? s.rowptr[L.hi] : h.rowptr[L.hi];
      const int64_t n32 = 98 * ((std::max<int64_t>(L.nnzb, 1) + 1) / 2);
      AMGCHK(alloc_ranged(L.vals32, 98 * (b0 / 2), 98 * ((b1 + 1) / 2), n32, err));
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
  // rows of level l + 1 whose Galerkin blocks / restricted residuals this rank forms from level l: the
  // aggregates of its own rows when level l is partitioned, all of them otherwise -- and where a partitioned
  // level meets a replicated one the owners' pieces are all-gathered (spans kept here)
  lvl_offs.assign(nl, std::vector<int64_t>());
  lvl_blk_offs.assign(nl, std::vector<int64_t>());
  for (int l = 1; l < nl; ++l) {
    amg[l].own_lo = 0; amg[l].own_hi = amg[l].nb;
    if (sharded(l - 1)) {
      // (one rank with forced collectives -- the transport's self-test -- owns everything)
      const std::vector<int32_t> rb = H[l].row_begin.empty() ? std::vector<int32_t>{0, H[l].nb} : H[l].row_begin;
      amg[l].own_lo = rb[comm.rank];
      amg[l].own_hi = rb[comm.rank + 1];
      lvl_offs[l].resize(comm.world + 1);
      lvl_blk_offs[l].resize(comm.world + 1);
      for (int r = 0; r <= comm.world; ++r) {
        lvl_offs[l][r] = 7 * (int64_t)rb[r];
        lvl_blk_offs[l][r] = 49 * (int64_t)H[l].rowptr[rb[r]];
      }
    }
    amg[l].own_b0 = H[l].rowptr[amg[l].own_lo];
    amg[l].own_b1 = H[l].rowptr[amg[l].own_hi];
  }
  if (opt.verbose) {
    std::fprintf(stderr, "sim3opt: multigrid levels (rows/blocks):");
    for (const AmgLevel& L : amg) std::fprintf(stderr, " %d/%lld", L.nb, (long long)L.nnzb);
    if (comm.active()) std::fprintf(stderr, "; levels 0..%d partitioned over the ranks, the rest replicated", n_sharded - 1);
    std::fprintf(stderr, "; visits of levels 1, 2, 3: %d %d %d\n", amg_visits[1], amg_visits[2], amg_visits[3]);
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
    // this rank's coarse blocks: all of them, or -- level l partitioned -- the rows that are aggregates of its
    // own fine rows (complete sums: no aggregate straddles two ranks, so nothing is reduced across ranks)
    const int cb0 = (int)Cc.own_b0, cb1 = (int)Cc.own_b1;
    const int gg = std::max(1, (cb1 - cb0 + 3) / 4), gw = (Cc.nb + 3) / 4;
    if (l == 0)
      hipLaunchKernelGGL((k_amg_galerkin<true>), dim3(gg), dim3(WG), 0, stream, cb0, cb1, F.gptr,
                         F.gblk, F.grow, F.colidx, F.vals, d_P, Cc.vals, amg_fp32 ? F.vals32 : (float*)nullptr);
    else
      hipLaunchKernelGGL((k_amg_galerkin<false>), dim3(gg), dim3(WG), 0, stream, cb0, cb1, F.gptr,
                         F.gblk, F.grow, F.colidx, F.vals, (const double*)nullptr, Cc.vals,
                         amg_fp32 ? F.vals32 : (float*)nullptr);
    if (comm.active() && sharded(l) && !sharded(l + 1)) {
      // the first replicated level: every rank needs all of its blocks (for its own passes, the products of
      // the levels below, the dense inverse) -- the owners' rows are all-gathered, once per linearisation
      int rc = comm.allgatherv(Cc.vals, lvl_blk_offs[l + 1], stream, err);
      if (rc) return rc;
    }
    // W = P^T P for ALL rows on every rank (block diagonal, from replicated data: cheaper than an exchange)
    if (l == 0)
      hipLaunchKernelGGL((k_amg_wsum<true>), dim3(gw), dim3(WG), 0, stream, Cc.nb, F.mptr, F.mem, d_P, Cc.W);
    else
      hipLaunchKernelGGL((k_amg_wsum<false>), dim3(gw), dim3(WG), 0, stream, Cc.nb, F.mptr, F.mem, F.W, Cc.W);
    hipLaunchKernelGGL(k_amg_copydiag, dim3(grid_for(49 * (int64_t)std::max(1, Cc.hi - Cc.lo), WG)), dim3(WG), 0, stream,
                       Cc.lo, Cc.hi, Cc.rowptr, Cc.vals, Cc.diagH);
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
    const int lo = L.lo, hi = L.hi;  // (this rank's rows on a partitioned level)
    jacobi(lo, hi, L.rowptr, L.vals, lambda, L.Minv, l == 0 && amg_additive ? 1.0 : amg_omega, L.diagH, L.W,
           l > 0 ? L.vals32 : (float*)nullptr);
  }
  dense_inverse(nullptr, d_Ainv, d_sc);
}

// dense inverse of the coarsest level into Aout: one launch per 14-row pivot block, buffers ping-pong between
// Aout and d_Ainv2.  diag64: the damped diagonal blocks of one of several systems solved together (else the
// level's own, damped by amg_prepare).
void Engine::dense_inverse(const double* diag64, double* Aout, DevScalars* sc) {
  const int nl = (int)amg.size();
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
  double *src = nsteps % 2 ? d_Ainv2 : Aout, *dst = nsteps % 2 ? Aout : d_Ainv2;
  (void)hipMemsetAsync(src, 0, sizeof(double) * (size_t)nd * nd, stream);
  hipLaunchKernelGGL(k_amg_dense_fill, dim3(grid_for(49 * Lc.nnzb, WG)), dim3(WG), 0, stream, Lc.nb,
                     Lc.rowptr, Lc.colidx, Lc.vals, src, diag64);
  double *pin = d_piv, *pout = d_piv + 28 * 28;
  switch (pivot_rows(0)) {
    case 28: hipLaunchKernelGGL((k_amg_dense_gj_first<28>), dim3(1), dim3(64), 0, stream, nd, (const double*)src, pin, sc); break;
    case 14: hipLaunchKernelGGL((k_amg_dense_gj_first<14>), dim3(1), dim3(64), 0, stream, nd, (const double*)src, pin, sc); break;
    default: hipLaunchKernelGGL((k_amg_dense_gj_first<7>), dim3(1), dim3(64), 0, stream, nd, (const double*)src, pin, sc);
  }
  const dim3 gt((nd + 63) / 64, (nd + 63) / 64 + 1);  // row 0 of the grid: the look-ahead workgroup
  for (int k0 = 0; k0 < nd;) {
    const int pb = pivot_rows(k0), pbn = k0 + pb < nd ? pivot_rows(k0 + pb) : 0;
    if (pb == 28)
      hipLaunchKernelGGL((k_amg_dense_gj_step<28>), gt, dim3(WG), 0, stream, nd, k0, (const double*)src,
                         dst, (const double*)pin, pout, pbn, sc);
    else if (pb == 14)
      hipLaunchKernelGGL((k_amg_dense_gj_step<14>), gt, dim3(WG), 0, stream, nd, k0, (const double*)src,
                         dst, (const double*)pin, pout, pbn, sc);
    else
      hipLaunchKernelGGL((k_amg_dense_gj_step<7>), gt, dim3(WG), 0, stream, nd, k0, (const double*)src,
                         dst, (const double*)pin, pout, pbn, sc);
    k0 += pb;
    std::swap(src, dst);
    std::swap(pin, pout);
  }  // the inverse is in Aout
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
                   (const int32_t*)L.agg, amg_over, BatchStrides{0, 0, 0, 0, 0}, (const float*)nullptr)
#define AMG_SPMV32(NTV, MODEV)                                                                    \
hipLaunchKernelGGL((k_spmv_span<(NTV) ? SIM3OPT_F32_CH : SIM3OPT_COARSE_CH, NTV, MODEV, float>), dim3(L.span_grid), dim3(WG), 0, stream,  \
                   L.nb, L.wrow, L.rowptr, L.colidx, (const float*)L.vals32, v, out, 0.0,         \
                   rz_part, rvec, const_cast<double*>(xc),                                        \
                   level == 0 ? d_sc : (DevScalars*)nullptr, L.Minv, 1, (const int32_t*)L.agg, amg_over, BatchStrides{0, 0, 0, 0, 0}, (const float*)nullptr)
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
  // the aggregates this rank restricts into: those of its own rows (level l partitioned) or all
  const int a0 = Cc.own_lo, a1 = Cc.own_hi;
  const int gr = grid_for((a1 - a0 + 8) / 9, 4);
  // a partitioned level above a replicated one: the owners' pieces of the restricted residual are
  // all-gathered, then every rank applies the first smoothing step to all rows
  const bool gather = comm.active() && sharded(l) && !sharded(l + 1);
  const double* Minv_c = l + 2 < (int)amg.size() ? Cc.Minv : nullptr;  // coarsest: solved exactly
  if (l == 0)
    hipLaunchKernelGGL(k_amg_restrict0, dim3(std::max(1, (a1 - a0 + 3) / 4)), dim3(WG), 0, stream, a0, a1, F.mptr,
                       F.mem, d_P, t, Cc.r, gather ? (const double*)nullptr : Minv_c, Cc.x,
                       (const DevScalars*)d_sc);
  else
    hipLaunchKernelGGL(k_amg_restrict, dim3(gr), dim3(WG), 0, stream, a0, a1, F.mptr, F.mem, t, Cc.r,
                       gather ? (const double*)nullptr : Minv_c, Cc.x);
  if (gather) {
    if (amg_status == SIM3OPT_OK) amg_status = comm.allgatherv(Cc.r, lvl_offs[l + 1], stream, amg_err);
    if (Minv_c)
      hipLaunchKernelGGL(k_amg_bjapply, dim3(grid_for((Cc.nb + 8) / 9, 4)), dim3(WG), 0, stream, Cc.nb, Minv_c,
                         (const double*)Cc.r, Cc.x);
  }
}

// x_out = x_in + scale P x_c on this rank's rows of level l -- and, when level l is partitioned, on the
// foreign rows its rows read (their aggregates' corrections came with the exchange of x_c), so that the
// smoothing pass that follows needs no exchange of its own
void Engine::amg_prolong(int l, const double* xc, const double* xin, double* xout) {
  const AmgLevel& F = amg[l];
  const int gp = grid_for((F.hi - F.lo + 8) / 9, 4);
  const DevScalars* scp = l == 0 ? (const DevScalars*)d_sc : (const DevScalars*)nullptr;
  if (l == 0)
    hipLaunchKernelGGL((k_amg_prolong<true>), dim3(gp), dim3(WG), 0, stream, F.lo, F.hi, (const int32_t*)nullptr,
                       F.agg, d_P, xc, xin, xout, scp, amg_over);
  else
    hipLaunchKernelGGL((k_amg_prolong<false>), dim3(gp), dim3(WG), 0, stream, F.lo, F.hi, (const int32_t*)nullptr,
                       F.agg, (const double*)nullptr, xc, xin, xout, scp, amg_over);
  if (comm.active() && sharded(l) && !parts[l].neighbour) {  // (no neighbour plan: the whole vector travels)
    amg_exchange(l, xout);
  } else if (comm.active() && sharded(l) && parts[l].n_recv > 0 && !parts[l].self_test) {  // (self-test: own rows)
    const int n = parts[l].n_recv, gl = grid_for((n + 8) / 9, 4);
    if (l == 0)
      hipLaunchKernelGGL((k_amg_prolong<true>), dim3(gl), dim3(WG), 0, stream, 0, n, (const int32_t*)parts[l].d_recv,
                         F.agg, d_P, xc, xin, xout, scp, amg_over);
    else
      hipLaunchKernelGGL((k_amg_prolong<false>), dim3(gl), dim3(WG), 0, stream, 0, n, (const int32_t*)parts[l].d_recv,
                         F.agg, (const double*)nullptr, xc, xin, xout, scp, amg_over);
  }
}

// exchange on a partitioned level inside the cycle (errors are collected in amg_status)
void Engine::amg_exchange(int l, double* vec) {
  if (!comm.active() || !sharded(l)) return;
  if (amg_status == SIM3OPT_OK) amg_status = exchange_level(l, vec, amg_err);
}

// Solves the level-(l+1) problem approximately (right-hand side amg[l+1].r, first iterate
// amg[l+1].x = Minv r already there) by amg_visits[l+1] cycles; returns the buffer with the result
// (on a partitioned level: this rank's rows of it).
double* Engine::amg_coarse(int l) {
  // (time_kernels: the visits of the first level a rank partition replicates are bracketed by events --
  // what they add up to is the part of the cycle that does not shrink with the number of ranks)
  if (opt.time_kernels && l + 1 == rep_level && part_world() > 1) {
    if (rep_used + 2 > rep_pool.size()) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (event_acquire(&e0) == hipSuccess && event_acquire(&e1) == hipSuccess) {
        rep_pool.push_back(e0);
        rep_pool.push_back(e1);
      }
    }
    if (rep_used + 2 <= rep_pool.size()) {
      (void)hipEventRecord(rep_pool[rep_used], stream);
      double* res = amg_coarse_body(l);
      (void)hipEventRecord(rep_pool[rep_used + 1], stream);
      rep_used += 2;
      return res;
    }
  }
  return amg_coarse_body(l);
}

double* Engine::amg_coarse_body(int l) {
  const int nl = (int)amg.size();
  const AmgLevel& Cc = amg[l + 1];
  if (l + 2 == nl) {
    hipLaunchKernelGGL(k_amg_dense_apply, dim3(std::max(1, std::min(256, (7 * Cc.nb + 3) / 4))), dim3(WG),
                       0, stream, 7 * Cc.nb, d_Ainv, Cc.r, Cc.x, (const DevScalars*)nullptr);
    return Cc.x;
  }
  amg_exchange(l + 1, Cc.x);  // (a cycle reads its iterate on this rank's rows and on the rows they refer to)
  double* res = amg_cycle(l + 1, Cc.x, Cc.t);
  for (int g = 1; g < amg_visits[l + 1]; ++g) {  // W-cycle: again, from the current iterate
    double* oth = res == Cc.x ? Cc.t : Cc.x;
    amg_exchange(l + 1, res);
    spmv_mode(Cc, 2, l + 1, res, oth, Cc.r);  // pre-smoothing step
    amg_exchange(l + 1, oth);
    res = amg_cycle(l + 1, oth, res);
  }
  return res;
}

// One multigrid cycle on level l from the iterate `cur` (valid on this rank's rows and, on a partitioned
// level, on the foreign rows they read); `other` is scratch; returns the buffer that holds the new iterate
// (always `other`; on a partitioned level: this rank's rows of it):
//   t = r - A cur;  coarse correction;  cur += P x_c;  other = cur + Minv (r - A cur)
double* Engine::amg_cycle(int l, double* cur, double* other) {
  const AmgLevel& F = amg[l];
  spmv_mode(F, 1, l, cur, other, F.r);
  amg_restrict(l, other);
  double* xc = amg_coarse(l);
  // the correction of a partitioned coarser level is needed for the aggregates of the foreign rows too
  amg_exchange(l + 1, xc);
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
// Multi-GPU: level 0 and the large coarse levels are row-partitioned (LevelPart): every matrix pass is
// preceded by the exchange of the rows its input is read on; the first replicated level receives the
// owners' pieces of the restricted residual by an all-gather, and every rank runs the same cycle below it.
int Engine::amg_apply(std::string& err) {
  amg_status = SIM3OPT_OK;
  if (amg_additive) {
    amg_restrict(0, d_r);
    double* xc0 = amg_coarse(0);
    amg_exchange(1, xc0);
    amg_over = amg_over_on ? amg_over_l[0] : 1.0;
    amg_prolong(0, xc0, d_z, d_az);
  } else {
    amg_exchange(0, d_z);
    amg_cycle(0, d_z, d_az);
  }
  if (amg_status == SIM3OPT_OK) amg_exchange(0, d_az);  // the PCG's SpMV reads z on the neighbours' rows
  if (amg_status != SIM3OPT_OK) {
    err = amg_err;
    return amg_status;
  }
  return SIM3OPT_OK;
}

}  // namespace sim3opt
