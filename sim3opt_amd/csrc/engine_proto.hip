// engine_proto.hip -- measurement prototype, NOT part of libsim3opt.so: the two-phase SpMV over upper-triangle
// storage (symm_proto.hpp; evaluated in rounds 2 and 3 and rejected for the product path).
// Compiled only into a SIM3OPT_BENCH_HOOKS build (`SIM3OPT_BENCH_HOOKS=1 python -m sim3opt_amd.build --force`),
// which scripts/gpu_spmv_symm.py needs.
#include "engine_impl.hpp"

namespace sim3opt {

#include "symm_proto.hpp"
#include "spmv_kernel.hpp"
#include "rowlane_proto.hpp"

// Measurement prototype (symm_proto.hpp): out[0] = ms of phase 1, out[1] = ms of phase 2, out[2] = max
// |difference| to the product SpMV relative to max |q|, out[3] = bytes of the upper-triangle stream
// (blocks + column indices + the t vectors written and read back).  Single GPU only.
int engine_bench_spmv_symmetric(Engine* e, int32_t reps, double out[4], std::string& err) {
  if (!e->linearized) {
    err = "bench_spmv_symmetric: call sim3opt_linearize (or optimize) first";
    return SIM3OPT_ERR_STATE;
  }
  if (e->comm.active()) {
    err = "bench_spmv_symmetric: single GPU only";
    return SIM3OPT_ERR_STATE;
  }
  const int nb = e->nb;
  std::vector<int32_t> rowptr(nb + 1), colidx((size_t)e->nnzb);
  HIPCHK(hipMemcpy(rowptr.data(), e->d_rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(colidx.data(), e->d_colidx, sizeof(int32_t) * colidx.size(), hipMemcpyDeviceToHost));
  std::vector<int32_t> urowptr(nb + 1, 0), ucol, usrc, lptr(nb + 1, 0), lidx;
  for (int i = 0; i < nb; ++i) {
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (colidx[k] >= i) {
        if (colidx[k] > i) ++lptr[colidx[k] + 1];
        ucol.push_back(colidx[k]);
        usrc.push_back(k);
      }
    urowptr[i + 1] = (int32_t)ucol.size();
  }
  for (int j = 0; j < nb; ++j) lptr[j + 1] += lptr[j];
  lidx.resize(lptr[nb]);
  {
    std::vector<int32_t> fill(lptr.begin(), lptr.end() - 1);
    for (int i = 0; i < nb; ++i)
      for (int k = urowptr[i]; k < urowptr[i + 1]; ++k)
        if (ucol[k] > i) lidx[fill[ucol[k]]++] = k;
  }
  const int nU = (int)ucol.size();
  int32_t *d_ur = nullptr, *d_uc = nullptr, *d_us = nullptr, *d_lp = nullptr, *d_li = nullptr;
  double *d_uv = nullptr, *d_t = nullptr, *d_y = nullptr;
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* p : tmp) dev_free(p); };
#define SYM_UP(D, H)                                                                              \
  do {                                                                                            \
    if (dev_malloc((void**)&D, sizeof(int32_t) * std::max<size_t>(H.size(), 1)) != hipSuccess) {    \
      cleanup(); err = "bench_spmv_symmetric: hipMalloc"; return SIM3OPT_ERR_HIP; }               \
    tmp.push_back(D);                                                                             \
    (void)hipMemcpy(D, H.data(), sizeof(int32_t) * H.size(), hipMemcpyHostToDevice);               \
  } while (0)
  SYM_UP(d_ur, urowptr); SYM_UP(d_uc, ucol); SYM_UP(d_us, usrc); SYM_UP(d_lp, lptr); SYM_UP(d_li, lidx);
#undef SYM_UP
  if (dev_malloc((void**)&d_uv, sizeof(double) * 49 * (size_t)nU) != hipSuccess ||
      dev_malloc((void**)&d_t, sizeof(double) * 7 * (size_t)nU) != hipSuccess ||
      dev_malloc((void**)&d_y, sizeof(double) * 7 * (size_t)nb) != hipSuccess) {
    if (d_uv) tmp.push_back(d_uv);
    if (d_t) tmp.push_back(d_t);
    cleanup();
    err = "bench_spmv_symmetric: hipMalloc";
    return SIM3OPT_ERR_HIP;
  }
  tmp.push_back(d_uv); tmp.push_back(d_t); tmp.push_back(d_y);
  hipStream_t st = e->stream;
  hipLaunchKernelGGL(k_symm_copy, dim3(2048), dim3(WG), 0, st, nU, d_us, e->d_vals, d_uv);
  (void)hipMemcpyAsync(e->d_p, e->d_b, sizeof(double) * (size_t)e->n, hipMemcpyDeviceToDevice, st);
  // reference: the product SpMV, q = H p
  e->spmv_raw(0.0, e->d_p, e->d_q, e->d_b, nullptr);
  const int g1 = (nb + 3) / 4, g2 = (7 * nb + WG - 1) / WG;
  // variant 1 (SIM3OPT_SYMM_VARIANT=1, round 3): phase 1 with row spans, pipelining, shared gather and
  // batched t stores (k_symm_phase1_span); its spans are balanced by the stored upper blocks
  const bool span = std::getenv("SIM3OPT_SYMM_VARIANT") && std::atoi(std::getenv("SIM3OPT_SYMM_VARIANT")) == 1;
  int gs = std::max(std::min(2048, (nb + 3) / 4), (nb + 15) / 16);
  if (const char* ev = std::getenv("SIM3OPT_SPAN_GRID")) gs = std::max(8, std::atoi(ev));
  int32_t* d_uw = nullptr;
  {
    std::vector<int32_t> uw(gs * 4 + 1);
    partition_rows(nb, urowptr.data(), gs * 4, uw.data());
    if (dev_malloc((void**)&d_uw, sizeof(int32_t) * uw.size()) != hipSuccess) { cleanup(); err = "bench_spmv_symmetric: hipMalloc"; return SIM3OPT_ERR_HIP; }
    tmp.push_back(d_uw);
    (void)hipMemcpy(d_uw, uw.data(), sizeof(int32_t) * uw.size(), hipMemcpyHostToDevice);
  }
  auto phase1 = [&]() {
    if (span) hipLaunchKernelGGL(k_symm_phase1_span, dim3(gs), dim3(WG), 0, st, nb, d_uw, d_ur, d_uc, d_uv, e->d_p, d_y, d_t);
    else hipLaunchKernelGGL(k_symm_phase1, dim3(g1), dim3(WG), 0, st, nb, d_ur, d_uc, d_uv, e->d_p, d_y, d_t);
  };
  phase1();
  hipLaunchKernelGGL(k_symm_phase2, dim3(g2), dim3(WG), 0, st, 7 * nb, d_lp, d_li, d_t, d_y);
  std::vector<double> q((size_t)7 * nb), ys((size_t)7 * nb);
  (void)hipMemcpyAsync(q.data(), e->d_q, sizeof(double) * q.size(), hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(ys.data(), d_y, sizeof(double) * ys.size(), hipMemcpyDeviceToHost, st);
  if (hipStreamSynchronize(st) != hipSuccess) { cleanup(); err = "bench_spmv_symmetric: sync"; return SIM3OPT_ERR_HIP; }
  double qmax = 0.0, dmax = 0.0;
  for (size_t k = 0; k < q.size(); ++k) {
    qmax = std::max(qmax, std::fabs(q[k]));
    dmax = std::max(dmax, std::fabs(q[k] - ys[k]));
  }
  out[2] = qmax > 0 ? dmax / qmax : dmax;
  float ms = 0.f;
  for (int w = 0; w < 3; ++w) phase1();
  (void)hipEventRecord(e->ev_a, st);
  for (int w = 0; w < reps; ++w) phase1();
  (void)hipEventRecord(e->ev_b, st);
  (void)hipEventSynchronize(e->ev_b);
  (void)hipEventElapsedTime(&ms, e->ev_a, e->ev_b);
  out[0] = reps > 0 ? ms / reps : 0.0;
  (void)hipEventRecord(e->ev_a, st);
  for (int w = 0; w < reps; ++w)
    hipLaunchKernelGGL(k_symm_phase2, dim3(g2), dim3(WG), 0, st, 7 * nb, d_lp, d_li, d_t, d_y);
  (void)hipEventRecord(e->ev_b, st);
  (void)hipEventSynchronize(e->ev_b);
  (void)hipEventElapsedTime(&ms, e->ev_a, e->ev_b);
  out[1] = reps > 0 ? ms / reps : 0.0;
  out[3] = (double)nU * (392.0 + 4.0) + 2.0 * 56.0 * (double)(nU - nb) + 4.0 * (double)(nU - nb) +
           (double)(nb + 1) * 8.0 + 3.0 * 56.0 * (double)nb;
  cleanup();
  return hipGetLastError() == hipSuccess ? SIM3OPT_OK : SIM3OPT_ERR_HIP;
}

// Measurement prototype (rowlane_proto.hpp): the level-0 FP32 passes of the multigrid cycle with a group of 7 lanes
// per block row (a block row per lane, in-lane products) against the product kernel, on the same matrix, vectors
// and smoother inverses.  out[0..1] = ms of the product's residual / smoothing pass, out[2..3] = the prototype's,
// out[4..5] = the prototype's with four systems sharing the block stream, out[6..7] = max |difference| of the
// one-system results to the product's relative to max |q|.  `rows_per_group` block rows per lane group
// (spans balanced by block count).  Needs the hierarchy (config 3); single GPU only.
int engine_bench_spmv_rowlane(Engine* e, int32_t reps, int32_t rows_per_group, double out[8], std::string& err) {
  if (!e->linearized || e->amg.empty() || !e->amg[0].vals32 || !e->amg[0].Minv) {
    err = "bench_spmv_rowlane: needs a linearised system with the multigrid hierarchy set up (run optimize first)";
    return SIM3OPT_ERR_STATE;
  }
  if (e->comm.active()) {
    err = "bench_spmv_rowlane: single GPU only";
    return SIM3OPT_ERR_STATE;
  }
  const Engine::AmgLevel& L = e->amg[0];
  const int nb = L.nb;
  const size_t n = (size_t)7 * nb;
  constexpr int KS = 4;
  std::vector<int32_t> rowptr(nb + 1);
  HIPCHK(hipMemcpy(rowptr.data(), L.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  const int nnzb = rowptr[nb];
  // (rows_per_group > 64: the number of lane groups itself)
  const int ngroups = rows_per_group > 64 ? std::min(nb, (int)rows_per_group)
                                          : std::max(9, (nb + std::max(1, rows_per_group) - 1) / std::max(1, rows_per_group));
  std::vector<int32_t> grow(ngroups + 1);
  partition_rows(nb, rowptr.data(), ngroups, grow.data());
  const int nwaves = (ngroups + 8) / 9, grid = (nwaves + 3) / 4;
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* p : tmp) dev_free(p); };
  auto alloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (dev_malloc(&p, bytes) != hipSuccess) return nullptr;
    tmp.push_back(p);
    return p;
  };
  int32_t* d_grow = (int32_t*)alloc(sizeof(int32_t) * grow.size());
  float* d_v32 = (float*)alloc(sizeof(float) * 49 * (size_t)nnzb);
  double* d_pv = (double*)alloc(sizeof(double) * n * KS);
  double* d_rv = (double*)alloc(sizeof(double) * n * KS);
  double* d_q = (double*)alloc(sizeof(double) * n * KS);
  double* d_qref = (double*)alloc(sizeof(double) * n);
  double* d_mi = (double*)alloc(sizeof(double) * 49 * (size_t)nb * KS);
  double* d_part = (double*)alloc(sizeof(double) * (size_t)grid * 4 * KS);
  if (!d_grow || !d_v32 || !d_pv || !d_rv || !d_q || !d_qref || !d_mi || !d_part) {
    cleanup();
    err = "bench_spmv_rowlane: hipMalloc";
    return SIM3OPT_ERR_HIP;
  }
  hipStream_t st = e->stream;
  (void)hipMemcpyAsync(d_grow, grow.data(), sizeof(int32_t) * grow.size(), hipMemcpyHostToDevice, st);
  hipLaunchKernelGGL(k_rl_copy, dim3(4096), dim3(WG), 0, st, (size_t)49 * nnzb, (const double*)L.vals, d_v32);
  for (int s = 0; s < KS; ++s) {  // input: the PCG's right-hand side and direction of the last solve (anything non-trivial)
    (void)hipMemcpyAsync(d_pv + s * n, e->d_z, sizeof(double) * n, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(d_rv + s * n, e->d_b, sizeof(double) * n, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(d_mi + (size_t)s * 49 * nb, L.Minv, sizeof(double) * 49 * (size_t)nb, hipMemcpyDeviceToDevice, st);
  }
  auto product = [&](int mode) {
    if (mode == 1)
      hipLaunchKernelGGL((k_spmv_span<SIM3OPT_F32_CH, true, 1, float>), dim3(L.span_grid), dim3(WG), 0, st, L.nb, L.wrow,
                         L.rowptr, L.colidx, (const float*)L.vals32, (const double*)d_pv, d_qref, 0.0, (double*)nullptr,
                         (const double*)d_rv, (double*)nullptr, (DevScalars*)nullptr, (const double*)L.Minv, 0,
                         (const int32_t*)L.agg, 1.0, BatchStrides{0, 0, 0, 0, 0}, (const float*)nullptr);
    else
      hipLaunchKernelGGL((k_spmv_span<SIM3OPT_F32_CH, true, 2, float>), dim3(L.span_grid), dim3(WG), 0, st, L.nb, L.wrow,
                         L.rowptr, L.colidx, (const float*)L.vals32, (const double*)d_pv, d_qref, 0.0, e->d_part_b,
                         (const double*)d_rv, (double*)nullptr, (DevScalars*)nullptr, (const double*)L.Minv, 0,
                         (const int32_t*)L.agg, 1.0, BatchStrides{0, 0, 0, 0, 0}, (const float*)nullptr);
  };
  auto proto = [&](int mode, int ks) {
#define RL(MODEV, KV)                                                                                              \
  hipLaunchKernelGGL((k_spmv_rowlane<MODEV, KV>), dim3(grid), dim3(WG), 0, st, ngroups, (const int32_t*)d_grow,       \
                     (const int32_t*)L.rowptr, (const int32_t*)L.colidx, nnzb, (const float*)d_v32, (const double*)d_pv, \
                     d_q, (const double*)d_rv, (const double*)d_mi, d_part, (int64_t)n, (int64_t)49 * nb)
    if (mode == 1) { if (ks == 1) RL(1, 1); else RL(1, KS); }
    else { if (ks == 1) RL(2, 1); else RL(2, KS); }
#undef RL
  };
  std::vector<double> qa(n), qb(n);
  for (int mode = 1; mode <= 2; ++mode) {
    product(mode);
    proto(mode, 1);
    (void)hipMemcpyAsync(qa.data(), d_qref, sizeof(double) * n, hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(qb.data(), d_q, sizeof(double) * n, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) { cleanup(); err = "bench_spmv_rowlane: sync"; return SIM3OPT_ERR_HIP; }
    double qmax = 0.0, dmax = 0.0;
    for (size_t k = 0; k < n; ++k) {
      qmax = std::max(qmax, std::fabs(qa[k]));
      dmax = std::max(dmax, std::fabs(qa[k] - qb[k]));
    }
    out[5 + mode] = qmax > 0 ? dmax / qmax : dmax;
  }
  auto timed = [&](auto&& launch) {
    float ms = 0.f;
    for (int w = 0; w < 3; ++w) launch();
    (void)hipEventRecord(e->ev_a, st);
    for (int w = 0; w < reps; ++w) launch();
    (void)hipEventRecord(e->ev_b, st);
    (void)hipEventSynchronize(e->ev_b);
    (void)hipEventElapsedTime(&ms, e->ev_a, e->ev_b);
    return reps > 0 ? (double)ms / reps : 0.0;
  };
  out[0] = timed([&]() { product(1); });
  out[1] = timed([&]() { product(2); });
  out[2] = timed([&]() { proto(1, 1); });
  out[3] = timed([&]() { proto(2, 1); });
  out[4] = timed([&]() { proto(1, KS); });
  out[5] = timed([&]() { proto(2, KS); });
  cleanup();
  return hipGetLastError() == hipSuccess ? SIM3OPT_OK : SIM3OPT_ERR_HIP;
}

}  // namespace sim3opt
