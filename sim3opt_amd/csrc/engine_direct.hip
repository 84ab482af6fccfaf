// engine_direct.hip -- exact sparse block Cholesky (LinearSolverEigen = SimplicialLDLT, kitti_surf.cpp:553-554)
#include "engine_impl.hpp"

namespace sim3opt {

#include "direct_kernels.hpp"

void Engine::direct_gather() {
  ldl.vals = d_vals;
  ldl.b = d_b;
  hipLaunchKernelGGL(k_ldl_gather, dim3(std::max(1, std::min(1024, (ldl.nL + 3) / 4))), dim3(WG), 0, stream, ldl);
}

// plan (host, once per initialize) + buffers; leaves use_direct false when the factorisation
// would be too expensive (the PCG takes over) unless the caller insists
int Engine::direct_init(const Structure& s, std::string& err) {
  const bool forced = opt.linear_solver == 1;
  if (comm.world > 1) {
    if (forced) {
      err = "linear_solver = 1: the exact factorisation runs on one GPU (small graphs are not sharded)";
      return SIM3OPT_ERR_ARG;
    }
    return SIM3OPT_OK;
  }
  // automatic: only where a factorisation costs less than a few PCG iterations would
  int64_t max_pairs = forced ? 30000000 : 300000;
  int32_t subtree = 0;
  if (opt.direct_max_pairs > 0) max_pairs = opt.direct_max_pairs;
  if (const char* ev = std::getenv("SIM3OPT_DIRECT_SUBTREE")) subtree = std::atoi(ev);
  if (const char* ev = std::getenv("SIM3OPT_DIRECT_WG_SUB")) ldl_wg_sub = std::max(64, std::min(LDL_WG_TOP, std::atoi(ev) / 64 * 64));
  if (!forced && nb > 60000) return SIM3OPT_OK;
  std::string why;
  if (!build_direct_plan(nb, s.rowptr.data(), s.colidx.data(), max_pairs, subtree, dplan, why,
                         ldl_wg_sub / 64)) {
    dplan = DirectPlan();
    if (forced) {
      err = "linear_solver = 1: " + why;
      return SIM3OPT_ERR_ARG;
    }
    if (opt.verbose) std::fprintf(stderr, "sim3opt: no exact factorisation (%s): PCG\n", why.c_str());
    return SIM3OPT_OK;
  }
  int rc = SIM3OPT_OK;
#define DCHK(call) do { rc = (call); if (rc) return rc; } while (0)
  DCHK(direct_up(ldl.perm, dplan.perm, err));
  DCHK(direct_up(ldl.colptr, dplan.colptr, err));
  DCHK(direct_up(ldl.lrow, dplan.lrow, err));
  DCHK(direct_up(ldl.lcol, dplan.lcol, err));
  DCHK(direct_up(ldl.srcptr, dplan.srcptr, err));
  DCHK(direct_up(ldl.src, dplan.src, err));
  DCHK(direct_up(ldl.pairptr, dplan.pairptr, err));
  DCHK(direct_up(ldl.pa, dplan.pa, err));
  DCHK(direct_up(ldl.pb, dplan.pb, err));
  DCHK(direct_up(ldl.pcol, dplan.pcol, err));
  DCHK(direct_up(ldl.gptr, dplan.gptr, err));
  DCHK(direct_up(ldl.lcolp, dplan.lcolp, err));
  DCHK(direct_up(ldl.tpre, dplan.tpre, err));
  DCHK(direct_up(ldl.tprey, dplan.tprey, err));
  ldl.ntpre = (int32_t)dplan.tpre.size();
  ldl.ntprey = (int32_t)dplan.tprey.size();
  DCHK(direct_up(ldl.bord, dplan.bord, err));
  DCHK(direct_up(ldl.brow, dplan.brow, err));
  DCHK(direct_up(ldl.rptr, dplan.rptr, err));
  DCHK(direct_up(ldl.cells, dplan.cells, err));
  ldl.nb = nb;
  ldl.nL = (int32_t)dplan.nL;
  DCHK(direct_alloc(ldl.Aperm, (size_t)49 * dplan.nL, err));
  DCHK(direct_alloc(ldl.bp, (size_t)7 * nb, err));
  DCHK(direct_alloc(ldl.L, (size_t)49 * dplan.nL, err));
  DCHK(direct_alloc(ldl.Dinv, (size_t)49 * nb, err));
  DCHK(direct_alloc(ldl.y, (size_t)7 * nb, err));
  DCHK(direct_alloc(ldl.xp, (size_t)7 * nb, err));
#undef DCHK
  ldl.dbg = nullptr;
  if (std::getenv("SIM3OPT_DIRECT_TRACE")) {  // tuning aid: per-level time stamps of the top group
    double* p = nullptr;
    int rc2 = direct_alloc(p, 256, err);
    if (rc2) return rc2;
    ldl.dbg = reinterpret_cast<long long*>(p);
  }
  if (opt.verbose)
    std::fprintf(stderr,
                 "sim3opt: exact block Cholesky: %d columns, %lld blocks in L, %lld block products, "
                 "tree height %d, %d groups\n",
                 nb, (long long)dplan.nL, (long long)dplan.npairs, dplan.height, dplan.ngroups());
  use_direct = true;
  return SIM3OPT_OK;
}

// (H + lambda I) x = b, exactly; x in d_x.  A non-positive pivot raises d_sc->fail (read by the
// caller together with the trial's chi2: no extra round trip).
int Engine::direct_solve(double lambda, std::string& err) {
  // (no reset of d_sc->fail: a failing factorisation stores this solve's token there, older values differ)
  fail_token = fail_token >= (1 << 30) ? 2 : fail_token + 1;
  ldl.fail_token = fail_token;
  ldl.vals = d_vals;
  ldl.b = d_b;
  ldl.x = d_x;
  ldl.sc = d_sc;
  ldl.lambda = lambda;
  const int ng = dplan.ngroups();
  if (ng > 1)
    hipLaunchKernelGGL((k_ldl<true, false>), dim3(ng - 1), dim3(ldl_wg_sub), 0, stream, ldl, 0);
  hipLaunchKernelGGL((k_ldl<true, true>), dim3(1), dim3(LDL_WG_TOP), 0, stream, ldl, ng - 1);
  if (ng > 1)
    hipLaunchKernelGGL((k_ldl<false, true>), dim3(ng - 1), dim3(ldl_wg_sub), 0, stream, ldl, 0);
  HIPCHK(hipGetLastError());
  if (ldl.dbg) {
    long long h[256];
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(h, ldl.dbg, sizeof(h), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "sim3opt: direct solve, top group stamps [us from start] (level start / after A+B per round / ... / down start / end):");
    for (long long i = 0; i < h[255] && i < 255; ++i) std::fprintf(stderr, " %.1f", (h[i] - h[0]) * 0.01);
    std::fprintf(stderr, "\n");
  }
  return SIM3OPT_OK;
}

}  // namespace sim3opt
