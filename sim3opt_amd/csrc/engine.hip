// engine.hip -- initialisation, linearisation, chi2 and the LM trial loop (g2o: SparseOptimizer::optimize ->
// OptimizationAlgorithmLevenberg::solve, kitti_surf.cpp:674-675); the C++ interface capi.cpp calls
#include "engine_impl.hpp"

namespace sim3opt {

#include "lm_kernels.hpp"

EdgeArgs Engine::edge_args() const {
  return EdgeArgs{e_lo, e_hi, d_ev0, d_ev1, d_meas, has_info ? d_info : nullptr,
                  has_kernel ? d_kdelta : nullptr, d_states, mopts()};
}

void Engine::release() {
  // (the calling thread's current device may be another one by now: a second graph on another device,
  // a rank thread's parent -- the pools of devmem.cpp key on the creating device, the runtime calls
  // below on the current one)
  int dev_prev = -1;
  if (device_used >= 0 && hipGetDevice(&dev_prev) == hipSuccess && dev_prev != device_used)
    (void)hipSetDevice(device_used);
  else
    dev_prev = -1;
  release_under_device();
  if (dev_prev >= 0) (void)hipSetDevice(dev_prev);
}

void Engine::release_under_device() {
  // cached blocks are handed out again without the device-wide wait a hipFree implies
  if (stream) (void)hipStreamSynchronize(stream);
  void* ptrs[] = {d_states, d_backup, d_meas, d_ev0, d_ev1, d_hidx, d_active, d_info, d_kdelta,
                  d_rowptr, d_colidx, d_incptr, d_wrow, d_slot01, d_slot10, d_inc0, d_inc1,
                  d_b, d_Minv, d_x, d_r, d_z, d_p, d_q, d_s, d_part_a, d_part_b, d_sc,
                  d_sub_first, d_sub_cnt, d_Gm, d_ptab};
  for (void* p : ptrs)
    if (p) dev_free(p);
  for (const RangedArray& a : ranged)
    if (a.alloc) dev_free(a.alloc);
  ranged.clear();
  d_vals = d_scratch = nullptr;
  for (void* p : amg_owned)
    if (p) dev_free(p);
  amg_owned.clear();
  for (LevelPart& lp : parts) {
    void* q[] = {lp.d_send, lp.d_recv, lp.d_sbuf, lp.d_rbuf};
    for (void* p : q)
      if (p) dev_free(p);
  }
  parts.clear();
  n_sharded = 0;
  batch_release();
  for (void* p : direct_owned)
    if (p) dev_free(p);
  direct_owned.clear();
  staged.release();
  if (h_sc) host_free(h_sc);
  for (hipEvent_t e : pool) event_release(e);
  for (hipEvent_t e : rep_pool) event_release(e);
  rep_pool.clear();
  if (ev_a) event_release(ev_a);
  if (ev_b) event_release(ev_b);
  for (hipEvent_t& e : ev_ph) if (e) { event_release(e); e = nullptr; }
  if (pcg_graph) (void)hipGraphExecDestroy(pcg_graph);
  if (stream) stream_release(stream);  // (synchronised above; kept for the next engine on this device)
  comm.release();
}

int Engine::init(const HostGraph& g, const Structure& s, std::string& err) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    err = "no usable HIP device (libsim3opt has no CPU fallback)";
    return SIM3OPT_ERR_NO_DEVICE;
  }
  if (opt.device >= 0) {
    if (opt.device >= ndev) {
      err = "device ordinal out of range";
      return SIM3OPT_ERR_ARG;
    }
    HIPCHK(hipSetDevice(opt.device));
  }
  HIPCHK(hipGetDevice(&device_used));
  if (const char* ev = std::getenv("SIM3OPT_SPMV")) {
    int a = 0, b = 0;
    if (std::sscanf(ev, "%d,%d", &a, &b) == 2) { spmv_chunk = a; spmv_nt = b; }
  }
  HIPCHK(stream_acquire(&stream));
  phase_timing = opt.time_kernels != 0 || opt.verbose != 0 || s.nb > 4096;
  st = s;
  nv = g.nv(); ne = g.ne(); nb = s.nb; n = 7 * nb; nnzb = s.nnzb;
  // row partition (world == 1: everything is local)
  row_begin.assign(comm.world + 1, 0);
  partition_rows_equal(nb, comm.world, row_begin.data());
  r0 = row_begin[comm.rank];
  r1 = row_begin[comm.rank + 1];
  offs.resize(comm.world + 1);
  for (int r = 0; r <= comm.world; ++r) offs[r] = 7 * (int64_t)row_begin[r];
  if (comm.active()) {
    parts.assign(1, LevelPart());
    n_sharded = 1;
    int rc = level_part_init(0, nb, s.rowptr.data(), s.colidx.data(), row_begin, err);
    if (rc) return rc;
  }
  e_lo = (int32_t)((int64_t)ne * comm.rank / comm.world);
  e_hi = (int32_t)((int64_t)ne * (comm.rank + 1) / comm.world);
  // this rank linearises the edges incident to its rows and writes only its rows' blocks
  std::vector<int32_t> l_active, l_s01 = s.slot01, l_s10 = s.slot10, l_i0 = s.inc0, l_i1 = s.inc1;
  if (comm.world > 1) {
    for (int32_t k : s.active) {
      const int32_t a = s.hidx[g.ev0[k]], b = s.hidx[g.ev1[k]];
      const bool la = a >= r0 && a < r1, lb = b >= r0 && b < r1;
      if (!la) { l_s01[k] = -1; l_i0[k] = -1; }
      if (!lb) { l_s10[k] = -1; l_i1[k] = -1; }
      if (la || lb) l_active.push_back(k);
    }
  } else {
    l_active = s.active;
  }
  n_active = (int32_t)l_active.size();
  has_info = g.has_info;
  has_kernel = g.has_kernel;
  const bool itrace = std::getenv("SIM3OPT_INIT_TRACE") != nullptr;
  auto inow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double it0 = inow();
  HIPCHK(event_acquire(&ev_a));
  HIPCHK(event_acquire(&ev_b));
  for (hipEvent_t& e : ev_ph) HIPCHK(event_acquire(&e));
  HIPCHK(upload(staged, stream, d_states, g.states));
  HIPCHK(dev_malloc((void**)&d_backup, sizeof(Sim3) * (size_t)nv));
  HIPCHK(upload(staged, stream, d_meas, g.meas));
  HIPCHK(upload(staged, stream, d_ev0, g.ev0));
  HIPCHK(upload(staged, stream, d_ev1, g.ev1));
  HIPCHK(upload(staged, stream, d_hidx, s.hidx));
  HIPCHK(upload(staged, stream, d_active, l_active));
  if (has_info) HIPCHK(upload(staged, stream, d_info, g.info));
  if (has_kernel) HIPCHK(upload(staged, stream, d_kdelta, g.kdelta));
  HIPCHK(upload(staged, stream, d_rowptr, s.rowptr));
  HIPCHK(upload(staged, stream, d_colidx, s.colidx));
  HIPCHK(upload(staged, stream, d_incptr, s.incptr));
  {  // span SpMV: contiguous row span per wavefront, balanced by stored blocks
    const int nloc = r1 - r0;
    // 3x the resident set (256 CUs x 8 workgroups of 4 wavefronts): shorter spans make the
    // addresses in flight a window that moves through the matrix instead of 8192 streams spread
    // over all of it (measured: 2048 -> 0.172 ms, 4096 -> 0.164, 6144 -> 0.1626, 8192 -> 0.1627,
    // 16384 -> 0.179 on config 3); small systems get one block row per wavefront
    // rule: ~4 block rows per wavefront (16 per workgroup), but never fewer workgroups than the
    // resident set as long as every wavefront still gets a row
    span_grid = std::max(std::min(2048, (nloc + 3) / 4), (nloc + 15) / 16);
    if (const char* ev = std::getenv("SIM3OPT_SPAN_GRID")) span_grid = std::min(std::atoi(ev), (nloc + 3) / 4);  // tuning knob
    span_grid = std::max(8, std::min(SPAN_GRID_MAX, span_grid));
    const int nw = span_grid * 4;
    std::vector<int32_t> wrow(nw + 1);
    partition_rows(nloc, s.rowptr.data() + r0, nw, wrow.data());
    for (int32_t& w : wrow) w += r0;
    HIPCHK(upload(staged, stream, d_wrow, wrow));
  }
  HIPCHK(upload(staged, stream, d_slot01, l_s01));
  HIPCHK(upload(staged, stream, d_slot10, l_s10));
  HIPCHK(upload(staged, stream, d_inc0, l_i0));
  HIPCHK(upload(staged, stream, d_inc1, l_i1));
  // H and the assembly scratch: this rank's rows only (alloc_ranged; one rank: everything)
  {
    int rc = alloc_ranged(d_vals, 49 * (int64_t)s.rowptr[r0], 49 * (int64_t)s.rowptr[r1], 49 * (int64_t)nnzb, err);
    if (rc) return rc;
    rc = alloc_ranged(d_scratch, 35 * (int64_t)s.incptr[r0], 35 * (int64_t)s.incptr[r1], 35 * (int64_t)s.incptr[nb], err);
    if (rc) return rc;
  }
  HIPCHK(dev_malloc((void**)&d_Minv, sizeof(double) * 49 * (size_t)nb));
  // preconditioner choice: chain segments for chain-like graphs (few blocks per row)
  // automatic: chain segments only when almost every edge is a chain link (KITTI with one loop:
  // 3963 PCG iterations per 30 LM iterations instead of 621642); with many loops the low-rank
  // argument is gone and the sequential apply costs more than it saves (measured)
  int64_t chain_links = 0;
  for (int32_t i = 1; i < nb; ++i)
    for (int32_t k = s.rowptr[i] + 1; k < s.rowptr[i + 1]; ++k)
      if (s.colidx[k] == i - 1) { ++chain_links; break; }
  const int64_t off_chain_edges = (nnzb - nb) / 2 - chain_links;
  // Automatic choice: the exact factorisation where it is cheap (KITTI-00, chain-like graphs); else
  // the multigrid hierarchy whenever the graph coarsens like a low-dimensional one (level-1 blocks
  // <= 0.3 x level-0 blocks: chains, Manhattan worlds -- not expanders such as config 2, where
  // block-Jacobi converges in tens of iterations) -- in either arithmetic (round 3: with the
  // coefficient as written the hierarchy sets up without a failing pivot on config 3 and every solve
  // converges, 11 ... 690 iterations, where block-Jacobi stops at its 1000-iteration cap from the
  // sixth LM iteration on; scripts/gpu_refarith_amg.py); graphs too small for a hierarchy
  // (<= 256 rows) get chain segments if they are nearly pure chains -- in the well-posed arithmetic
  // only: as written cond(H + lambda I) reaches 1e12 on a chain and the recursive residual of so
  // strongly preconditioned a CG drifts from the true one --; block-Jacobi otherwise.
  // (naming a preconditioner asks for the PCG)
  const double it1 = inow();
  if (opt.linear_solver == 1 || (opt.linear_solver < 0 && opt.preconditioner < 0)) {
    int rc = direct_init(s, err);
    if (rc) return rc;
  }
  const double it2 = inow();
  if (!use_direct &&
      (opt.preconditioner == 2 || opt.preconditioner < 0)) {
    int rc = amg_init(s, opt.preconditioner < 0, err);
    if (rc) return rc;
  }
  use_chain = !use_amg && !use_direct &&
              (opt.preconditioner == 1 ||
               (opt.preconditioner < 0 && comm.world == 1 && opt.fix_small_angle_b != 0 &&
                off_chain_edges <= std::max<int64_t>(2, nb / 64)));
  chain_seg = std::max(2, std::min(opt.chain_segment > 0 ? opt.chain_segment : 256, CHAIN_SEG_MAX));
  if (use_chain) {
    std::vector<int32_t> sf(nb, -1), scnt(nb, 0);
    for (int32_t i = 1; i < nb; ++i)
      for (int32_t k = s.rowptr[i] + 1; k < s.rowptr[i + 1]; ++k)  // sorted by column after the diagonal
        if (s.colidx[k] == i - 1) {
          if (sf[i] < 0) sf[i] = k;
          ++scnt[i];
        }
    HIPCHK(upload(staged, stream, d_sub_first, sf));
    HIPCHK(upload(staged, stream, d_sub_cnt, scnt));
    HIPCHK(dev_malloc((void**)&d_Gm, sizeof(double) * 49 * (size_t)nb));
  }
  double** vecs[] = {&d_b, &d_x, &d_r, &d_z, &d_p, &d_q, &d_s};
  for (double** v : vecs) {
    // padded to world x (7 x rows per rank) so the all-gather can run in place with equal counts
    int64_t padded = 0;
    (void)allgather_equal_plan(offs.data(), comm.world, nullptr, &padded);
    const size_t n_alloc = std::max<size_t>((size_t)n, (size_t)padded);
    HIPCHK(dev_malloc((void**)v, sizeof(double) * n_alloc));
    HIPCHK(hipMemset(*v, 0, sizeof(double) * n_alloc));
  }
  if (use_amg) {  // level 0 aliases the system's own arrays and vectors
    int rc = amg_bind(s, err);
    if (rc) return rc;
  }
  HIPCHK(dev_malloc((void**)&d_part_a, sizeof(double) * SPAN_GRID_MAX));
  HIPCHK(dev_malloc((void**)&d_part_b, sizeof(double) * SPAN_GRID_MAX));
  HIPCHK(dev_malloc((void**)&d_sc, sizeof(DevScalars)));
  HIPCHK(hipMemset(d_sc, 0, sizeof(DevScalars)));
  HIPCHK(host_malloc((void**)&h_sc, sizeof(DevScalars)));
  // Gram task tables
  GramTables tab;
  int t = 0;
  for (int a = 0; a < 14; ++a)
    for (int b = a; b < 15; ++b) { tab.ga[t] = (unsigned char)a; tab.gb[t] = (unsigned char)b; ++t; }
  t = 0;
  for (int c = 0; c < 7; ++c)
    for (int r = 0; r <= c; ++r) { tab.tr[t] = (unsigned char)r; tab.tc[t] = (unsigned char)c; ++t; }
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &tab, sizeof(tab)));
  HIPCHK(hipDeviceSynchronize());
  staged.release();
  if (itrace)
    std::fprintf(stderr, "sim3opt engine init: uploads %.2f ms, factorisation plan + its uploads %.2f ms, rest %.2f ms\n",
                 it1 - it0, it2 - it1, inow() - it2);
  return SIM3OPT_OK;
}

int Engine::fetch_scalars(std::string& err) {
  HIPCHK(hipMemcpyAsync(h_sc, d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (comm.timing && comm.ev_used) return comm.drain(err);
  return SIM3OPT_OK;
}

// ---- timing helpers ----
int Engine::timed_begin(std::string& err) {
  HIPCHK(hipEventRecord(ev_a, stream));
  return SIM3OPT_OK;
}

int Engine::timed_end(double& ms_acc, std::string& err) {
  HIPCHK(hipEventRecord(ev_b, stream));
  HIPCHK(hipEventSynchronize(ev_b));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ev_a, ev_b));
  ms_acc += ms;
  return SIM3OPT_OK;
}

int Engine::pool_get(hipEvent_t& a, hipEvent_t& b, std::string& err) {
  if (pool_used + 2 > pool.size()) {
    hipEvent_t e0, e1;
    HIPCHK(event_acquire(&e0));
    HIPCHK(event_acquire(&e1));
    pool.push_back(e0);
    pool.push_back(e1);
  }
  a = pool[pool_used];
  b = pool[pool_used + 1];
  pool_used += 2;
  return SIM3OPT_OK;
}

int Engine::pool_drain(std::string& err) {
  if (pool_used > 0) {
    kt.n_spmv += (int64_t)std::max<long long>(0, h_sc->n_spmv_work - spmv_work_seen);
    spmv_work_seen = h_sc->n_spmv_work;
  }
  for (size_t i = 0; i + 1 < pool_used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pool[i], pool[i + 1]));
    kt.ms_spmv += ms;
  }
  pool_used = 0;
  for (size_t i = 0; i + 1 < rep_used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, rep_pool[i], rep_pool[i + 1]));
    kt.ms_replicated_levels += ms;
    kt.n_replicated_visits += 1;
  }
  rep_used = 0;
  return SIM3OPT_OK;
}

// ---- building blocks ----
// scale_parts > 0: d_part_b holds that many partial sums of the trial's scale (k_scale): summed in the
// same launch as chi2's
int Engine::chi2(double* out, std::string& err, hipEvent_t before_fetch, int scale_parts) {
  const int g = grid_for(e_hi - e_lo, WG);
  hipLaunchKernelGGL(k_chi2, dim3(g), dim3(WG), 0, stream, edge_args(), d_part_a);
  // (exact solver on one GPU: small systems, where the copy of the scalar block is a visible share of a trial)
  const bool mirror = scale_parts > 0 && use_direct && !comm.active() && !opt.time_kernels;
  if (scale_parts > 0)
    hipLaunchKernelGGL(k_final_sum_two, dim3(1), dim3(WG), 0, stream, (const double*)d_part_a, g, &d_sc->chi2,
                       (const double*)d_part_b, scale_parts, &d_sc->scale, mirror ? h_sc : (DevScalars*)nullptr,
                       (const DevScalars*)d_sc);
  else
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(WG), 0, stream, d_part_a, g, &d_sc->chi2);
  HIPCHK(hipGetLastError());
  int rc = SIM3OPT_OK;
  if (comm.active()) {  // chi2 and scale are adjacent: one 2-double all-reduce per LM trial
    rc = comm.allreduce(&d_sc->chi2, 2, 0, stream, err);
    if (rc) return rc;
  }
  if (before_fetch) HIPCHK(hipEventRecord(before_fetch, stream));
  if (mirror) HIPCHK(hipStreamSynchronize(stream));  // the kernel wrote h_sc's chi2 / scale / fail itself
  else rc = fetch_scalars(err);
  if (rc) return rc;
  *out = h_sc->chi2;
  kt.n_chi2 += 1;
  return SIM3OPT_OK;
}

// debug_full_arrays: nothing outside this rank's ranges may have been written (reads show as NaN in the results)
int Engine::check_foreign_ranges(std::string& err) {
  if (!opt.debug_full_arrays || ranged.empty()) return SIM3OPT_OK;
  unsigned long long* d_cnt = nullptr;
  HIPCHK(dev_malloc((void**)&d_cnt, sizeof(unsigned long long) * ranged.size()));
  HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * ranged.size(), stream));
  for (size_t i = 0; i < ranged.size(); ++i) {
    const RangedArray& a = ranged[i];
    const char* base = static_cast<const char*>(a.alloc);
    const size_t head = (size_t)a.lo * a.elem / 4, tail0 = (size_t)a.hi * a.elem, tail = ((size_t)a.total * a.elem - tail0) / 4;
    if (head) hipLaunchKernelGGL(k_count_unpoisoned, dim3(1024), dim3(WG), 0, stream, reinterpret_cast<const uint32_t*>(base), head, d_cnt + i);
    if (tail) hipLaunchKernelGGL(k_count_unpoisoned, dim3(1024), dim3(WG), 0, stream, reinterpret_cast<const uint32_t*>(base + tail0), tail, d_cnt + i);
  }
  std::vector<unsigned long long> h(ranged.size(), 0);
  hipError_t e1 = hipMemcpyAsync(h.data(), d_cnt, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  dev_free(d_cnt);
  HIPCHK(e1);
  HIPCHK(e2);
  for (size_t i = 0; i < h.size(); ++i)
    if (h[i]) {
      err = "debug_full_arrays: " + std::to_string(h[i]) + " words outside this rank's range of ranged array " +
            std::to_string(i) + " were written (rank " + std::to_string(comm.rank) + ")";
      return SIM3OPT_ERR_STATE;
    }
  return SIM3OPT_OK;
}

int Engine::linearize(std::string& err) {
  const sim3::Opts mo = mopts();
  if (!d_ptab) HIPCHK(dev_malloc((void**)&d_ptab, 14 * sizeof(Sim3)));
  if (ptab_delta != opt.fd_delta || ptab_opts.eps != mo.eps || ptab_opts.small_rot_half != mo.small_rot_half ||
      ptab_opts.fix_small_b != mo.fix_small_b) {
    hipLaunchKernelGGL(k_perturbation_table, dim3(1), dim3(64), 0, stream, opt.fd_delta, mo, d_ptab);
    ptab_delta = opt.fd_delta;
    ptab_opts = mo;
  }
  LinArgs A{n_active, d_active, d_ev0, d_ev1, d_meas, d_info, d_kdelta, d_states,
            d_slot01, d_slot10, d_inc0, d_inc1, d_vals, d_scratch, opt.fd_delta, mo,
            (const Sim3*)d_ptab, opt.dof_mask, d_sc};
  const int g = (n_active + EPB - 1) / EPB;
  if (g == 0) HIPCHK(hipMemsetAsync(&d_sc->maxdiag_bits, 0, sizeof(unsigned long long), stream));
  if (g > 0) {
    if (has_info && has_kernel)
      hipLaunchKernelGGL((k_linearize_numeric<true, true>), dim3(g), dim3(WG), 0, stream, A);
    else if (has_info)
      hipLaunchKernelGGL((k_linearize_numeric<true, false>), dim3(g), dim3(WG), 0, stream, A);
    else if (has_kernel)
      hipLaunchKernelGGL((k_linearize_numeric<false, true>), dim3(g), dim3(WG), 0, stream, A);
    else
      hipLaunchKernelGGL((k_linearize_numeric<false, false>), dim3(g), dim3(WG), 0, stream, A);
  }
  const int gdr = grid_for(r1 - r0, 4);
  hipLaunchKernelGGL(k_diag_reduce, dim3(gdr), dim3(WG), 0, stream, r0, r1,
                     d_incptr, d_rowptr, d_scratch, d_vals, d_b, d_sc, d_part_a, d_part_b);
  hipLaunchKernelGGL(k_final_trace_max, dim3(1), dim3(WG), 0, stream, (const double*)d_part_a,
                     (const double*)d_part_b, gdr, &d_sc->trace, &d_sc->maxdiag_bits);
  HIPCHK(hipGetLastError());
  if (comm.active()) {  // non-negative doubles order like their bit patterns
    int rc = comm.allreduce(reinterpret_cast<double*>(&d_sc->maxdiag_bits), 1, 1, stream, err);
    if (rc) return rc;
    rc = comm.allreduce(&d_sc->trace, 1, 0, stream, err);  // (every rank must take the same decisions)
    if (rc) return rc;
  }
  if (use_direct) direct_gather();  // the factorisation's starting blocks: H in the layout of L, b permuted
  linearized = true;
  amg_stale = true;
  trace_stale = true;
  kt.n_linearize += 1;
  return SIM3OPT_OK;
}

int Engine::optimize(int32_t max_iters, std::vector<sim3opt_iter_stats>& stats, std::string& err) {
  stats.clear();
  double lambda = 0.0, ni = 2.0;
  bool ok = true;
  int iters = 0;
  chi_known = false;  // (options or estimates may have changed since the last call)
  for (int it = 0; it < max_iters && ok; ++it) {
    sim3opt_iter_stats T{};
    double currentChi = 0.0;
    int rc = SIM3OPT_OK;
    // phase times: event stamps on the stream, read after the trial's chi2 fetch -- the loop has
    // ONE host round trip per trial (plus lambda_0's at the first iteration); waiting on every
    // phase's end event left the GPU idle a quarter of the time on the small graphs
    if (phase_timing) HIPCHK(hipEventRecord(ev_ph[0], stream));
    // computeActiveErrors at the start of an iteration: the estimates are those the last trial
    // evaluated (accepted) or restored (rejected), and the evaluation is deterministic, so the
    // value is already here -- one host round trip less per iteration
    if (chi_known) currentChi = chi_cache;
    else {
      rc = chi2(&currentChi, err);
      if (rc) return rc;
    }
    double tempChi = currentChi;
    T.chi2_before = currentChi;
    rc = linearize(err);
    if (rc) return rc;
    bool lin_pending = true;  // ev_ph[0] -> the first trial's ev_ph[1]
    if (it == 0) {
      rc = fetch_scalars(err);
      if (rc) return rc;
      double maxdiag;
      std::memcpy(&maxdiag, &h_sc->maxdiag_bits, sizeof(double));
      lambda = opt.user_lambda_init > 0 ? opt.user_lambda_init : opt.tau * maxdiag;
      ni = 2.0;
    }
    double rho = 0.0;
    int qmax = 0;
    // solutions of the next trials, solved together after a rejection (engine_batch.hip)
    struct { int n = 0, next = 0; double lam[KB]; int32_t iters[KB]; double rel[KB]; bool capped[KB]; } batch;
    bool prev_ok = false, prev_capped = false;  // the previous trial's solve (of this LM iteration)
    int32_t prev_pit = 0;
    auto elapsed = [&](int a, int b, double& acc) -> int {
      if (!phase_timing) return SIM3OPT_OK;
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ev_ph[a], ev_ph[b]));
      acc += ms;
      return SIM3OPT_OK;
    };
    do {
      if (phase_timing) HIPCHK(hipEventRecord(ev_ph[1], stream));  // (push(): k_oplus keeps the old estimates itself)
      int32_t pit = 0;
      double rres = 0.0;
      bool ok2 = true;
      const double* xsol = d_x;  // the step of this trial
      bool from_batch = false;
      if (batch.next < batch.n && batch.lam[batch.next] == lambda) {
        from_batch = true;
      } else {
        batch.n = batch.next = 0;
        // A rejection has just happened: g2o's rule fixes the dampings of the next trials (lambda *= nu, nu *= 2
        // per rejection), so the systems of the trials that may follow are solved TOGETHER -- one pass over the
        // blocks for all of them -- and evaluated one after the other exactly as before; a trial that is
        // accepted leaves the rest unused.  Only systems the hierarchy would solve anyway: a damping-dominated
        // one (lambda >= the block-Jacobi gate, adaptive_prec) is cheaper on its own.
        // ... and only while this iteration's solves behave: a batch runs until its LAST system is done, every
        // iteration at the price of all of them, and one failing system sends the whole batch to the sequential
        // path's fall-backs -- in the as-written arithmetic (solves of hundreds of iterations, break-downs, a
        // capped one) that made the reference_arithmetic leg 1.7x SLOWER; there the trials stay sequential.
        const bool calm = prev_ok && !prev_capped && prev_pit > 0 && prev_pit <= 100;
        const int cap = qmax >= 1 && calm ? std::min(batch_capacity(), opt.max_trials - qmax) : 0;
        if (cap >= 2) {
          double gate = DBL_MAX;
          if (adaptive_prec && !trace_stale && mean_diag > 0.0) gate = bj_gate >= 0.0 ? bj_gate : 0.05 * mean_diag;
          int nsys = 0;
          double l = lambda, nu = ni;
          while (nsys < cap && l < gate && std::isfinite(l)) {
            batch.lam[nsys++] = l;
            l *= nu;
            nu *= 2.0;
          }
          if (nsys >= 2) {
            bool usable = false;
            rc = pcg_batch(batch.lam, nsys, batch.iters, batch.rel, batch.capped, &usable, err);
            if (rc) return rc;
            if (usable) {
              batch.n = nsys;
              from_batch = true;
            }
          }
        }
      }
      if (from_batch) {
        const int s = batch.next++;
        xsol = b_x + (size_t)s * b_vs;
        pit = batch.iters[s];
        rres = batch.rel[s];
        last_capped = batch.capped[s];
      } else {
        rc = pcg(lambda, &pit, &rres, &ok2, err);
        if (rc) return rc;
      }
      if (phase_timing) HIPCHK(hipEventRecord(ev_ph[2], stream));
      T.pcg_iters += pit;
      T.pcg_rel_res = rres;
      if (last_capped) T.pcg_capped += 1;
      if (opt.verbose >= 2)
        std::fprintf(stderr, "  trial %d: lambda %.6g, %d PCG iterations (rel %.2e)\n", qmax, lambda, pit, rres);
      double scale = 0.0;
      if (ok2) {
        hipLaunchKernelGGL(k_oplus, dim3((nv + WG - 1) / WG), dim3(WG), 0, stream, nv, d_hidx,
                           xsol, d_states, mopts(), use_direct ? (const DevScalars*)d_sc : nullptr, d_backup,
                           fail_token);
        const int ge = grid_for(7 * (int64_t)(r1 - r0), WG);
        hipLaunchKernelGGL(k_scale, dim3(ge), dim3(WG), 0, stream, 7 * r0, 7 * r1, xsol, d_b,
                           lambda, d_part_b);
        HIPCHK(hipGetLastError());
        rc = chi2(&tempChi, err, phase_timing ? ev_ph[3] : nullptr, ge);  // also sums and brings back scale (and the factorisation's verdict)
        if (rc) return rc;
        rc = elapsed(2, 3, T.ms_update);
        if (rc) return rc;
        scale = h_sc->scale;
        kt.n_update += 1;
        if (use_direct && h_sc->fail == fail_token) {  // not positive definite: g2o's solver returns false
          tempChi = DBL_MAX;
          scale = 0.0;
        }
      } else {
        tempChi = DBL_MAX;  // solver failed: g2o forces rejection
        if (phase_timing) HIPCHK(hipEventSynchronize(ev_ph[2]));
        else HIPCHK(hipStreamSynchronize(stream));
      }
      rc = elapsed(1, 2, T.ms_solve);
      if (rc) return rc;
      if (lin_pending) {
        rc = elapsed(0, 1, T.ms_linearize);
        if (rc) return rc;
        kt.ms_linearize += T.ms_linearize;
        lin_pending = false;
      }
      rho = currentChi - tempChi;
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1.0 - std::pow(2 * rho - 1, 3);
        alpha = std::min(alpha, opt.good_step_upper);
        lambda *= std::max(opt.good_step_lower, alpha);
        ni = 2.0;
        currentChi = tempChi;  // discardTop
      } else {
        lambda *= ni;
        ni *= 2.0;
        if (ok2)  // pop (a failed solve never touched the estimates -- nor the backup)
          hipLaunchKernelGGL(k_copy_states, dim3((8 * nv + WG - 1) / WG), dim3(WG), 0, stream, nv,
                             (const Sim3*)d_backup, d_states);
      }
      prev_ok = ok2;
      prev_capped = last_capped;
      prev_pit = pit;
      ++qmax;
    } while (rho < 0 && qmax < opt.max_trials);
    kt.ms_update += T.ms_update;
    chi_known = true;
    chi_cache = currentChi;
    T.chi2_after = currentChi;
    T.lambda = lambda;
    T.rho = rho;
    T.trials = qmax;
    stats.push_back(T);
    ++iters;
    if (opt.verbose)
      std::fprintf(stderr,
                   "iteration= %d\t chi2= %.9g\t lambda= %.6g\t levenbergIter= %d\t pcg= %d "
                   "(rel %.2e)\t ms lin/solve/upd= %.3f/%.3f/%.3f\n",
                   it, currentChi, lambda, qmax, T.pcg_iters, T.pcg_rel_res, T.ms_linearize,
                   T.ms_solve, T.ms_update);
    if (qmax == opt.max_trials || rho == 0 || !std::isfinite(lambda)) ok = false;  // Terminate
  }
  HIPCHK(hipStreamSynchronize(stream));
  return iters;
}

// ------------------------------------------------------------------------------------------
// C++ interface used by capi.cpp
// ------------------------------------------------------------------------------------------
Engine* engine_create(const HostGraph& g, const Structure& s, const sim3opt_options& opt,
                      Comm* comm, std::string& err, int& status) {
  Engine* e = new Engine();
  e->opt = opt;
  if (comm) {  // the engine takes the communicator over
    e->comm = *comm;
    *comm = Comm();
  }
  e->comm.timing = opt.time_kernels != 0;
  status = e->init(g, s, err);
  if (status != SIM3OPT_OK) {
    delete e;
    return nullptr;
  }
  return e;
}

void engine_destroy(Engine* e) { delete e; }

void engine_take_comm(Engine* e, Comm* out) {
  *out = e->comm;   // the caller owns the communicator again (re-initialisation keeps the ranks)
  e->comm = Comm();
}

int engine_set_options(Engine* e, const sim3opt_options& opt) {
  const int dev = e->opt.device, full = e->opt.debug_full_arrays;  // (fixed at initialisation)
  e->opt = opt;
  e->opt.device = dev;
  e->opt.debug_full_arrays = full;
  e->comm.timing = opt.time_kernels != 0;
  // (the events exist in any case: the flags may be set after sim3opt_initialize)
  e->phase_timing = opt.time_kernels != 0 || opt.verbose != 0 || e->nb > 4096;
  return SIM3OPT_OK;
}

int engine_optimize(Engine* e, int32_t max_iters, std::vector<sim3opt_iter_stats>& stats,
                    std::string& err) {
  int rc = e->optimize(max_iters, stats, err);
  if (rc) return rc;
  return e->check_foreign_ranges(err);
}

int engine_chi2(Engine* e, double* chi2, std::string& err) { return e->chi2(chi2, err); }

int engine_get_states(Engine* e, Sim3* out, std::string& err) {
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out, e->d_states, sizeof(Sim3) * (size_t)e->nv, hipMemcpyDeviceToHost));
  return SIM3OPT_OK;
}

int engine_set_states(Engine* e, const Sim3* in, std::string& err) {
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(e->d_states, in, sizeof(Sim3) * (size_t)e->nv, hipMemcpyHostToDevice));
  e->linearized = false;
  e->chi_known = false;
  return SIM3OPT_OK;
}

int engine_edge_errors(Engine* e, double* out, std::string& err) {
  double* d_out = nullptr;
  HIPCHK(dev_malloc((void**)&d_out, sizeof(double) * 7 * std::max<size_t>((size_t)e->ne, 1)));
  EdgeArgs ea = e->edge_args();
  ea.e_lo = 0;
  ea.e_hi = e->ne;
  hipLaunchKernelGGL(k_edge_errors, dim3(grid_for(e->ne, WG)), dim3(WG), 0, e->stream, ea, d_out);
  hipError_t le = hipGetLastError();
  if (le == hipSuccess) le = hipStreamSynchronize(e->stream);
  if (le == hipSuccess)
    le = hipMemcpy(out, d_out, sizeof(double) * 7 * (size_t)e->ne, hipMemcpyDeviceToHost);
  dev_free(d_out);
  if (le != hipSuccess) {
    err = std::string("edge_errors: ") + hipGetErrorString(le);
    return SIM3OPT_ERR_HIP;
  }
  return SIM3OPT_OK;
}

int engine_linearize(Engine* e, std::string& err) {
  int rc = e->linearize(err);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  return e->check_foreign_ranges(err);
}

int engine_get_system(Engine* e, int32_t* rowptr, int32_t* colidx, double* values, double* b,
                      std::string& err) {
  if (!e->linearized) {
    err = "get_system: call sim3opt_linearize (or optimize) first";
    return SIM3OPT_ERR_STATE;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  if (rowptr) std::memcpy(rowptr, e->st.rowptr.data(), sizeof(int32_t) * (size_t)(e->nb + 1));
  if (colidx) std::memcpy(colidx, e->st.colidx.data(), sizeof(int32_t) * (size_t)e->nnzb);
  if (values) {  // (a partitioned run holds this rank's rows only: the other rows' blocks read zero)
    const size_t k0 = (size_t)49 * e->st.rowptr[e->r0], k1 = (size_t)49 * e->st.rowptr[e->r1];
    std::memset(values, 0, sizeof(double) * 49 * (size_t)e->nnzb);
    if (k1 > k0) HIPCHK(hipMemcpy(values + k0, e->d_vals + k0, sizeof(double) * (k1 - k0), hipMemcpyDeviceToHost));
  }
  if (b) HIPCHK(hipMemcpy(b, e->d_b, sizeof(double) * (size_t)e->n, hipMemcpyDeviceToHost));
  return SIM3OPT_OK;
}

int engine_solve(Engine* e, double lambda, double* x, int32_t* iters, double* rel_res,
                 std::string& err) {
  if (!e->linearized) {
    err = "solve: call sim3opt_linearize (or optimize) first";
    return SIM3OPT_ERR_STATE;
  }
  int32_t it = 0;
  double rr = 0.0;
  bool ok = true;
  int rc = e->pcg(lambda, &it, &rr, &ok, err);
  if (rc) return rc;
  if (e->use_direct) {  // the factorisation reports a non-positive pivot through the scalars
    rc = e->fetch_scalars(err);
    if (rc) return rc;
    ok = e->h_sc->fail != e->fail_token;
  }
  if (iters) *iters = it;
  if (rel_res) *rel_res = rr;
  if (x) HIPCHK(hipMemcpy(x, e->d_x, sizeof(double) * (size_t)e->n, hipMemcpyDeviceToHost));
  if (!ok) {
    err = "solve: PCG breakdown (system not positive definite)";
    return SIM3OPT_ERR_STATE;
  }
  return SIM3OPT_OK;
}

void engine_local_rows(const Engine* e, int32_t* begin, int32_t* end) {
  if (begin) *begin = e->r0;
  if (end) *end = e->r1;
}

int engine_preconditioner(const Engine* e) { return e->use_amg ? 2 : (e->use_chain ? 1 : 0); }

int engine_linear_solver(const Engine* e) { return e->use_direct ? 1 : 0; }

void engine_device_bytes(const Engine* e, int64_t bytes[2]) {
  bytes[0] = e->ranged_bytes();
  bytes[1] = 0;
  for (const Engine::RangedArray& a : e->ranged) bytes[1] += (int64_t)a.elem * a.total;
}

void engine_amg_in_use(const Engine* e, int32_t* n_levels, int32_t* n_partitioned, int32_t visits[4]) {
  if (n_levels) *n_levels = e->use_amg ? (int32_t)e->amg.size() : 0;
  if (n_partitioned) *n_partitioned = e->use_amg ? e->n_sharded : 0;
  if (visits)
    for (int l = 0; l < 4; ++l) visits[l] = e->use_amg ? e->amg_visits[l + 1] : 0;
}

int engine_kernel_times(Engine* e, sim3opt_kernel_times* out, bool reset) {
  if (out) *out = e->kt;
  if (reset) {
    e->kt = sim3opt_kernel_times{};
    e->comm.times = sim3opt_comm_times{};
  }
  return SIM3OPT_OK;
}

int engine_comm_times(Engine* e, sim3opt_comm_times* out) {
  std::string err;
  if (e->comm.ev_used) {  // pairs recorded since the last synchronisation
    if (hipStreamSynchronize(e->stream) != hipSuccess) return SIM3OPT_ERR_HIP;
    int rc = e->comm.drain(err);
    if (rc) return rc;
  }
  *out = e->comm.times;
  return SIM3OPT_OK;
}


}  // namespace sim3opt
