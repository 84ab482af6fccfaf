// engine_batch.hip -- the multigrid-preconditioned CG for several right-hand sides at once: the rejected
// trials of one LM iteration (OptimizationAlgorithmLevenberg::solve, reached from kitti_surf.cpp:675) solve
// (H + lambda_k I) x_k = b for a known sequence lambda_k; after the first rejection the next ones are solved
// together -- one pass over the blocks for K vectors, K vectors per coarse launch -- and evaluated in g2o's
// order (Engine::optimize).  Per system the arithmetic is that of Engine::pcg_attempt with the multigrid
// preconditioner, operation by operation: the K solutions are bit for bit those of K sequential solves.
#include "engine_impl.hpp"

namespace sim3opt {

#include "spmv_kernel.hpp"
#include "batch_kernels.hpp"

static inline int64_t pad64(int64_t n) { return (n + 63) / 64 * 64; }

// every batched kernel is instantiated for 2, 3 and 4 systems: a batch of three must not pay for four
#define BATCH_DISPATCH(NS, ...) \
  do {                          \
    if ((NS) == 1) { constexpr int KS = 1; __VA_ARGS__; } \
    else if ((NS) == 2) { constexpr int KS = 2; __VA_ARGS__; } \
    else if ((NS) == 3) { constexpr int KS = 3; __VA_ARGS__; } \
    else { constexpr int KS = 4; __VA_ARGS__; } \
  } while (0)

// buffers of the batched solve, allocated at its first use (single GPU, multigrid path)
int Engine::batch_alloc(std::string& err) {
  if (batch_ready) return SIM3OPT_OK;
  const int nl = (int)amg.size();
  auto alloc = [&](double*& p, size_t count) -> int {
    HIPCHK(dev_malloc((void**)&p, sizeof(double) * std::max<size_t>(count, 1)));
    batch_owned.push_back(p);
    HIPCHK(hipMemsetAsync(p, 0, sizeof(double) * std::max<size_t>(count, 1), stream));
    return SIM3OPT_OK;
  };
  // (schedule only -- the results do not depend on it: which levels run one system per grid slice)
  if (const char* ev = std::getenv("SIM3OPT_BATCH_SLICE_BLOCKS")) b_slice_blocks = std::atoll(ev);
  b_vs = pad64(n);
  double** v0[] = {&b_x, &b_r, &b_z, &b_p, &b_q, &b_s, &b_az};
  for (double** v : v0) {
    int rc = alloc(*v, (size_t)KB * b_vs);
    if (rc) return rc;
  }
  blv.assign(nl, BatchLevel());
  for (int l = 0; l < nl; ++l) {
    BatchLevel& B = blv[l];
    const AmgLevel& L = amg[l];
    B.vs = l == 0 ? b_vs : pad64(7 * (int64_t)L.nb);
    B.ms = (int64_t)49 * L.nb;
    int rc = alloc(B.Minv, (size_t)KB * B.ms);
    if (rc) return rc;
    if (l == 0) {
      B.r = b_r; B.x = b_z; B.t = b_az;
    } else {
      if ((rc = alloc(B.r, (size_t)KB * B.vs))) return rc;
      if ((rc = alloc(B.x, (size_t)KB * B.vs))) return rc;
      if ((rc = alloc(B.t, (size_t)KB * B.vs))) return rc;
      HIPCHK(dev_malloc((void**)&B.diag32, sizeof(float) * (size_t)KB * B.ms));
      batch_owned.push_back(B.diag32);
      HIPCHK(hipMemsetAsync(B.diag32, 0, sizeof(float) * (size_t)KB * B.ms, stream));
    }
  }
  const size_t nc = (size_t)7 * amg[nl - 1].nb;
  b_as = (int64_t)(nc * nc);
  int rc = alloc(b_Ainv, (size_t)KB * b_as);
  if (rc) return rc;
  if ((rc = alloc(b_diag64, (size_t)KB * 49 * amg[nl - 1].nb))) return rc;
  if ((rc = alloc(b_part_a, (size_t)KB * SPAN_GRID_MAX))) return rc;
  if ((rc = alloc(b_part_b, (size_t)KB * SPAN_GRID_MAX))) return rc;
  HIPCHK(dev_malloc((void**)&d_bsc, sizeof(DevScalars) * KB));
  batch_owned.push_back(d_bsc);
  HIPCHK(hipMemsetAsync(d_bsc, 0, sizeof(DevScalars) * KB, stream));
  HIPCHK(host_malloc((void**)&h_bsc, sizeof(DevScalars) * KB));
  batch_ready = true;
  return SIM3OPT_OK;
}

void Engine::batch_release() {
  for (void* p : batch_owned)
    if (p) dev_free(p);
  batch_owned.clear();
  if (h_bsc) host_free(h_bsc);
  h_bsc = nullptr;
  d_bsc = nullptr;
  blv.clear();
  batch_ready = false;
}

// ---- the cycle for KB systems (mirrors spmv_mode / amg_restrict / amg_prolong / amg_coarse / amg_cycle) ----
void Engine::b_spmv_mode(int level, int mode, const double* v, double* out, const double* rvec, const double* xc) {
  const AmgLevel& L = amg[level];
  const BatchLevel& B = blv[level];
  double* const rz_part = level == 0 && mode == 2 ? b_part_b : nullptr;
  BatchStrides bs{B.vs, B.ms, level + 1 < (int)amg.size() ? blv[level + 1].vs : 0, B.ms, SPAN_GRID_MAX};
  const double over = amg_over;
#define BSPMV(CHV, NTV, MODEV, DIAGV)                                                                              \
  BATCH_DISPATCH(b_nsys, hipLaunchKernelGGL((k_spmv_span<CHV, NTV, MODEV, float, KS, DIAGV>), dim3(L.span_grid), \
                     dim3(WG), 0, stream,   \
                     L.nb, L.wrow, L.rowptr, L.colidx, (const float*)L.vals32, v, out, 0.0, rz_part, rvec,          \
                     const_cast<double*>(xc), level == 0 ? d_bsc : (DevScalars*)nullptr, (const double*)B.Minv, 1,  \
                     (const int32_t*)L.agg, over, bs, (const float*)B.diag32))
  // a level whose blocks stay in cache (<= b_slice_blocks: levels >= 2 of config 3) runs one system per grid
  // slice: its passes are launch-latency-bound, four times the wavefronts cost what one set costs, while one
  // wavefront carrying four systems takes 2.5x as long (measured)
#define BSPMV_SLICED(MODEV)                                                                                        \
  hipLaunchKernelGGL((k_spmv_span<8, false, MODEV, float, 1, true>), dim3(L.span_grid, b_nsys), dim3(WG), 0,     \
                     stream, L.nb, L.wrow, L.rowptr, L.colidx, (const float*)L.vals32, v, out, 0.0, rz_part, rvec,  \
                     const_cast<double*>(xc), (DevScalars*)nullptr, (const double*)B.Minv, 1,                      \
                     (const int32_t*)L.agg, over, bs, (const float*)B.diag32)
  if (level > 0 && L.nnzb <= b_slice_blocks) {
    if (mode == 1) BSPMV_SLICED(1); else if (mode == 3) BSPMV_SLICED(3); else BSPMV_SLICED(2);
    return;
  }
  if (level == 0) { if (mode == 1) BSPMV(SIM3OPT_F32_CH, true, 1, false); else BSPMV(SIM3OPT_F32_CH, true, 2, false); }
  else { if (mode == 1) BSPMV(8, false, 1, true); else if (mode == 3) BSPMV(8, false, 3, true); else BSPMV(8, false, 2, true); }
#undef BSPMV_SLICED
#undef BSPMV
}

void Engine::b_restrict(int l, const double* t) {
  const AmgLevel& F = amg[l];
  const AmgLevel& Cc = amg[l + 1];
  const BatchLevel &BF = blv[l], &BC = blv[l + 1];
  const double* Minv_c = l + 2 < (int)amg.size() ? BC.Minv : nullptr;  // coarsest: solved exactly
  if (l == 0)
    BATCH_DISPATCH(b_nsys, hipLaunchKernelGGL((k_amg_restrict0_k<KS>), dim3((Cc.nb + 3) / 4), dim3(WG), 0, stream, Cc.nb,
                       F.mptr, F.mem, d_P, t, BC.r, Minv_c, BC.x, (const DevScalars*)d_bsc, BF.vs, BC.vs, BC.ms));
  else  // (coarse vectors are a few MB at most: one system per grid slice, see b_spmv_mode)
    hipLaunchKernelGGL((k_amg_restrict_k<1>), dim3(grid_for((Cc.nb + 8) / 9, 4), b_nsys), dim3(WG), 0, stream, Cc.nb,
                       F.mptr, F.mem, t, BC.r, Minv_c, BC.x, BF.vs, BC.vs, BC.ms);
}

double* Engine::b_coarse(int l) {
  const int nl = (int)amg.size();
  const AmgLevel& Cc = amg[l + 1];
  const BatchLevel& BC = blv[l + 1];
  if (l + 2 == nl) {
    hipLaunchKernelGGL((k_amg_dense_apply_k<1>), dim3(std::max(1, std::min(256, (7 * Cc.nb + 3) / 4)), b_nsys), dim3(WG), 0,
                       stream, 7 * Cc.nb, (const double*)b_Ainv, (const double*)BC.r, BC.x, b_as, BC.vs);
    return BC.x;
  }
  double* res = b_cycle(l + 1, BC.x, BC.t);
  for (int g = 1; g < amg_visits[l + 1]; ++g) {
    double* oth = res == BC.x ? BC.t : BC.x;
    b_spmv_mode(l + 1, 2, res, oth, BC.r, nullptr);
    res = b_cycle(l + 1, oth, res);
  }
  return res;
}

double* Engine::b_cycle(int l, double* cur, double* other) {
  const AmgLevel& F = amg[l];
  const BatchLevel& BF = blv[l];
  b_spmv_mode(l, 1, cur, other, BF.r, nullptr);
  b_restrict(l, other);
  const double* xc = b_coarse(l);
  amg_over = amg_over_on ? amg_over_l[l] : 1.0;
  if (l == 0) {
    BATCH_DISPATCH(b_nsys, hipLaunchKernelGGL((k_amg_prolong0_k<KS>), dim3(grid_for((F.nb + 8) / 9, 4)), dim3(WG), 0, stream,
                       F.nb, F.agg, d_P, xc, (const double*)cur, cur, (const DevScalars*)d_bsc, amg_over, BF.vs,
                       blv[1].vs));
    b_spmv_mode(l, 2, cur, other, BF.r, nullptr);
  } else {
    b_spmv_mode(l, 3, cur, other, BF.r, xc);
  }
  return other;
}

// Solves (H + lams[s] I) x_s = b, s < nsys <= KB, together; x_s is left in b_x + s * b_vs.  *usable = false:
// some system broke down or failed the true-residual check -- the caller then solves the trials one by one
// (the sequential path has the fall-backs: plain cycle instead of the over-corrected one, block-Jacobi).
int Engine::pcg_batch(const double* lams, int nsys, int32_t* iters, double* rel_res, bool* capped, bool* usable,
                      std::string& err) {
  *usable = false;
  int rc = batch_alloc(err);
  if (rc) return rc;
  if (amg_stale) {
    rc = amg_setup(err);
    if (rc) return rc;
  }
  const int nl = (int)amg.size();
  const int max_it = opt.pcg_max_iters > 0 ? opt.pcg_max_iters : (n <= 50000 ? std::max(100, 2 * n) : 1000);
  for (int s = 0; s < KB; ++s) {
    DevScalars& h = h_bsc[s];
    std::memset(&h, 0, sizeof(DevScalars));
    h.max_iter = max_it;
    h.tol2 = opt.pcg_rel_tol * opt.pcg_rel_tol;
    h.lambda = s < nsys ? lams[s] : lams[nsys - 1];
    h.done = s < nsys ? 0 : 1;  // (an unused slot: finished from the start, its vectors stay zero)
  }
  HIPCHK(hipMemcpyAsync(d_bsc, h_bsc, sizeof(DevScalars) * KB, hipMemcpyHostToDevice, stream));
  // per system: damped diagonal blocks, smoother inverses, dense inverse of the coarsest level (amg_prepare)
  for (int s = 0; s < nsys; ++s) {
    for (int l = 0; l < nl; ++l) {
      const AmgLevel& L = amg[l];
      const BatchLevel& B = blv[l];
      jacobi(0, L.nb, L.rowptr, L.vals, lams[s], B.Minv + (size_t)s * B.ms, amg_omega, L.diagH, L.W, nullptr,
             d_bsc + s, l == nl - 1 ? b_diag64 + (size_t)s * 49 * L.nb : nullptr,
             l > 0 ? B.diag32 + (size_t)s * B.ms : nullptr);
    }
    dense_inverse(b_diag64 + (size_t)s * 49 * amg[nl - 1].nb, b_Ainv + (size_t)s * b_as, d_bsc + s);
  }
  const int gv = grid_for((nb + 8) / 9, 4);
  const int gs = span_grid;
  const BatchLevel& B0 = blv[0];
  BatchStrides bs0{b_vs, B0.ms, 0, 0, SPAN_GRID_MAX};
  b_nsys = nsys;
  BATCH_DISPATCH(nsys, hipLaunchKernelGGL((k_pcg_init_k<KS>), dim3(gv), dim3(WG), 0, stream, 0, nb, (const double*)d_b,
                     (const double*)B0.Minv, b_x, b_r, b_z, b_p, b_s, bs0));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h_bsc, d_bsc, sizeof(DevScalars) * KB, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  for (int s = 0; s < nsys; ++s)
    if (h_bsc[s].fail) return SIM3OPT_OK;  // a non-positive pivot of some set-up: not usable
  b_cycle(0, b_z, b_az);
  const int chunk = std::min(4, std::max(1, opt.pcg_check_every));
  int it = 0, par = 0;
  for (;;) {
    HIPCHK(hipMemcpyAsync(h_bsc, d_bsc, sizeof(DevScalars) * KB, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    // systems still iterating: [0, live).  The dampings ascend with the trial, so the systems finish from the
    // tail as a rule; the launches that follow carry the first `live` systems only (per system the same
    // operations whatever K is: the results do not depend on when the others finished)
    int live = 0;
    for (int s = 0; s < nsys; ++s)
      if (!(h_bsc[s].done || h_bsc[s].stop || h_bsc[s].fail)) live = s + 1;
    if (live == 0 || it >= max_it) break;
    b_nsys = live;
    const int todo = std::min(chunk, max_it - it);
    for (int c = 0; c < todo; ++c) {
      BATCH_DISPATCH(live, hipLaunchKernelGGL((k_spmv_span<8, true, 0, double, KS, false>), dim3(gs), dim3(WG), 0, stream,
                         nb, d_wrow, d_rowptr, d_colidx, (const double*)d_vals, (const double*)b_az, b_q, 0.0, b_part_a,
                         (const double*)nullptr, b_part_b, d_bsc, (const double*)nullptr, 1, (const int32_t*)nullptr, 1.0,
                         bs0, (const float*)nullptr));
      BATCH_DISPATCH(live, hipLaunchKernelGGL((k_final_sum2_k<KS>), dim3(1), dim3(WG), 0, stream, (const double*)b_part_a,
                         (const double*)b_part_b, gs, SPAN_GRID_MAX, d_bsc));
      BATCH_DISPATCH(live, hipLaunchKernelGGL((k_pcg_step_k<KS>), dim3(gv), dim3(WG), 0, stream, 0, nb, par, it,
                         (const double*)B0.Minv, (const double*)b_az, b_z, (const double*)b_q, b_p, b_s, b_x, b_r, d_bsc,
                         bs0));
      b_cycle(0, b_z, b_az);
      par ^= 1;
      ++it;
    }
    HIPCHK(hipGetLastError());
  }
  // what the stopping test claims, checked in the 2-norm per system (see pcg_attempt)
  bool good = true;
  for (int s = 0; s < nsys; ++s) {
    if (h_bsc[s].fail) { good = false; continue; }
    norms2(b_r + (size_t)s * b_vs, d_b, b_part_a, b_part_b, &d_bsc[s].tmp_pq);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h_bsc, d_bsc, sizeof(DevScalars) * KB, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  int it_max = 0;
  for (int s = 0; s < nsys; ++s) {
    const DevScalars& h = h_bsc[s];
    const double true_rel = h.tmp_rz > 0 ? std::sqrt(h.tmp_pq / h.tmp_rz) : 0.0;
    if (h.fail || true_rel > 1e-3) good = false;
    iters[s] = h.iter;
    rel_res[s] = h.rz0 > 0 ? std::sqrt(std::fabs(h.gam_last) / h.rz0) : 0.0;
    capped[s] = !h.fail && h.iter >= max_it && rel_res[s] > opt.pcg_rel_tol;
    it_max = std::max(it_max, (int)h.iter);
    if (opt.verbose)
      std::fprintf(stderr, "sim3opt: batched multigrid PCG, system %d of %d: lambda %.6g, %d iterations, ||r||_Minv ratio %.2e, "
                   "||r||_2 / ||b||_2 %.2e\n", s, nsys, lams[s], h.iter, rel_res[s], true_rel);
  }
  kt.n_pcg_vec += it_max;
  kt.n_batched_solves += nsys;
  kt.n_batches += 1;
  *usable = good;
  return SIM3OPT_OK;
}

}  // namespace sim3opt
