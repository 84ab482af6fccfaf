// engine.hip -- device-resident Levenberg-Marquardt on a Sim(3) pose graph, gfx950 (MI355X).
//
// What it replaces in the reference (all third-party g2o code reached from
// optimizer.optimize(100), kitti_surf.cpp:675; restated per SURVEY.md 3.3 / App. C):
//   EdgeSim3::computeError                     -> k_chi2, k_edge_errors, k_linearize_numeric
//   BaseBinaryEdge::linearizeOplus (numeric)   -> k_linearize_numeric (lane = one +-delta evaluation)
//   BaseBinaryEdge::constructQuadraticForm     -> k_linearize_numeric (Gram phase) + k_diag_reduce
//   BlockSolverX::buildSystem / setLambda      -> block-CSR values in HBM; lambda folded into SpMV
//   LinearSolverEigen::solve (SimplicialLDLT)  -> preconditioned CG: k_spmv_span, k_pcg_*; block-Jacobi
//                                                 (k_jacobi), chain segments (k_chain_*) or aggregation
//                                                 multigrid (amg.cpp, amg_kernels.hpp, Engine::amg_*)
//   VertexSim3Expmap::oplusImpl, push/pop      -> k_oplus + device-to-device backup copies
//   OptimizationAlgorithmLevenberg::solve      -> Engine::optimize (host control, 3 scalars per trial)
//
// HBM layout (all FP64, indices int32):
//   states   V x 8   AoS, 64 B per vertex (one gather = one half-line)
//   meas     E x 8   AoS, 64 B per edge; ev0/ev1 SoA int32; info E x 49 only if some edge is not I7
//   vals     nnzb x 49, column-major 7x7 blocks, block row = free vertex, diagonal block first
//   scratch  (#incidences) x 35: per (edge, endpoint) upper triangle of J^T W J (28) and -J^T W e (7)
//   PCG vectors x r z p q b: 7*nb each; Minv nb x 49 row-major
//   multigrid   P nb x 49 (Ad(S_v)), per coarse level its own block-CSR + diagH/W/Minv + 3 vectors,
//               dense inverse of the coarsest level (two n x n buffers, n <= 1792)
// Assembly is atomic-free and reduction orders are fixed, so results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "amg.hpp"
#include "comm.hpp"
#include "devmem.hpp"
#include "direct.hpp"
#include "engine.hpp"

namespace sim3opt {

using sim3::Sim3;

#define HIPCHK(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      err = std::string(#call) + ": " + hipGetErrorString(e_);              \
      return SIM3OPT_ERR_HIP;                                               \
    }                                                                       \
  } while (0)

#ifndef SIM3OPT_F32_CH
#define SIM3OPT_F32_CH 8        // blocks per pipeline step of the level-0 FP32 passes (tuning: 16)
#endif
constexpr int WG = 256;         // 4 wavefronts of 64
constexpr int PCG_GRAPH_ITERS = 16;  // PCG iterations per captured hipGraph (even: parity returns)
constexpr int MAX_GRID = 2048;  // grid cap of the streaming kernels = number of reduction partials
constexpr int SPAN_GRID_MAX = 65536;  // workgroups of the span SpMV (its partials: one pair each)
                                // (256 CUs x 8 workgroups of 4 waves = full occupancy)

// Scalars that live in HBM so the PCG loop needs no host round trip per iteration.
struct DevScalars {
  double rz[2];    // gamma = r.z of the previous PCG iteration (ping-pong by parity)
  double alpha[2]; // step length of the previous PCG iteration (ping-pong by parity)
  double rz0;      // r.z at PCG start
  double chi2;     // sum of (robustified) edge chi2
  double scale;    // x.(lambda x + b)
  unsigned long long maxdiag_bits;  // max |H_dd| as raw bits (non-negative doubles order as integers)
  int32_t iter;      // PCG iterations executed
  int32_t max_iter;  // PCG iteration cap
  int32_t done;      // PCG finished (converged, cap reached or breakdown)
  int32_t stop;      // set by the last allowed update; turned into `done` by the next launch
  int32_t fail;      // PCG breakdown (p.q <= 0 or non-finite) or non-SPD diagonal block
  double tol2;       // squared relative tolerance on ||r||_Minv
  double tmp_pq;     // multi-GPU: w.z summed over ranks  } adjacent: ONE 2-double all-reduce
  double tmp_rz;     // multi-GPU: r.z summed over ranks  } per PCG iteration
  double gam_last;   // r.z seen by the last executed step (reported relative residual)
  double lambda;     // damping of the current solve (read by the captured PCG launches)
  long long n_spmv_work;  // PCG SpMV launches that did their work (launches after `done` return at once)
  double trace;           // sum of the scalar diagonal of H (mean |H_dd|: when is a system damping-dominated?)
};

#include "lm_kernels.hpp"
#include "amg_kernels.hpp"
#include "direct_kernels.hpp"
#include "symm_proto.hpp"

#include "pcg_kernels.hpp"

// ------------------------------------------------------------------------------------------
// Engine
// ------------------------------------------------------------------------------------------
template <typename T>
static hipError_t upload(StagedUploads& staged, hipStream_t stream, T*& dptr, const std::vector<T>& h) {
  const size_t bytes = sizeof(T) * std::max<size_t>(h.size(), 1);
  hipError_t e = dev_malloc((void**)&dptr, bytes);
  if (e != hipSuccess) return e;
  // (small arrays: staged in pinned memory and enqueued; init synchronises once at its end)
  if (!h.empty()) e = staged.put(dptr, h.data(), sizeof(T) * h.size(), stream);
  return e;
}

static inline int grid_for(int64_t items, int per_block) {
  const int64_t g = (items + per_block - 1) / per_block;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, MAX_GRID));
}

class Engine {
 public:
  sim3opt_options opt;
  Structure st;  // host copy of the pattern
  int32_t nv = 0, ne = 0, nb = 0, n = 0, n_active = 0;
  int64_t nnzb = 0;
  bool has_info = false, has_kernel = false;
  hipStream_t stream = nullptr;
  // graph
  Sim3 *d_states = nullptr, *d_backup = nullptr, *d_meas = nullptr;
  int32_t *d_ev0 = nullptr, *d_ev1 = nullptr, *d_hidx = nullptr, *d_active = nullptr;
  double *d_info = nullptr, *d_kdelta = nullptr;
  // system
  int32_t *d_rowptr = nullptr, *d_colidx = nullptr, *d_incptr = nullptr, *d_wrow = nullptr;
  int span_grid = 0;  // workgroups of the span SpMV
  int32_t *d_slot01 = nullptr, *d_slot10 = nullptr, *d_inc0 = nullptr, *d_inc1 = nullptr;
  double *d_vals = nullptr, *d_scratch = nullptr, *d_b = nullptr, *d_Minv = nullptr;
  double *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_p = nullptr, *d_q = nullptr, *d_s = nullptr;
  double *d_part_a = nullptr, *d_part_b = nullptr;
  // chain-segment preconditioner (Sinv lives in d_Minv)
  int32_t *d_sub_first = nullptr, *d_sub_cnt = nullptr;
  double* d_Gm = nullptr;
  bool use_chain = false;
  int chain_seg = 256;
  // aggregation multigrid preconditioner (amg.hpp, amg_kernels.hpp); level 0 aliases the system
  struct AmgLevel {
    int32_t nb = 0;
    int64_t nnzb = 0;
    int32_t *rowptr = nullptr, *colidx = nullptr, *wrow = nullptr;
    int span_grid = 0;
    double *vals = nullptr, *diagH = nullptr, *W = nullptr, *Minv = nullptr;
    float* vals32 = nullptr;  // FP32 copy of vals for the cycle's matrix passes (amg_fp32)
    int32_t *agg = nullptr, *mptr = nullptr, *mem = nullptr, *gptr = nullptr, *gblk = nullptr, *grow = nullptr;
    double *r = nullptr, *x = nullptr, *t = nullptr;  // level right-hand side, iterate, residual / result
  };
  std::vector<AmgLevel> amg;
  std::vector<void*> amg_owned;
  double *d_P = nullptr, *d_Ainv = nullptr, *d_Ainv2 = nullptr, *d_piv = nullptr, *d_az = nullptr;
  int32_t* d_row2v = nullptr;
  bool use_amg = false, amg_stale = true;
  double amg_omega = 0.9;  // damping of the block-Jacobi smoother: eig(D^-1 A) <= 2 on every level
  int amg_visits[AMG_MAX_LEVELS + 1];  // cycles spent on level l per visit of level l-1 (1 = V, 2 = W)
  bool amg_additive = false;           // level 0 additive: no fine-level matrix pass in the cycle
  bool amg_fp32 = true;                // the cycle's matrix passes stream FP32 copies of the blocks
  // over-correction: the coarse correction prolonged INTO level l is scaled by amg_over_l[l]
  // (piecewise-constant prolongation under-estimates the correction; Stueben / Blaheta)
  double amg_over_l[AMG_MAX_LEVELS + 1];
  double amg_over = 1.0;               // (the factor of the launch being issued)
  bool amg_over_on = true;             // cleared when an over-corrected cycle made the PCG break down
  int amg_pivot = 14;                  // pivot block of the dense coarsest inverse (14 or 28 rows: the same
                                       // total time -- the in-wavefront pivot inverse is what costs)
  // Damping-dominated systems (round 3): when lambda is of the order of the diagonal of H -- the LM
  // trials at the noise floor of the delta = 1e-9 Jacobians, lambda 2e2 ... 8e3 on config 3 -- plain
  // block-Jacobi PCG converges in 3-11 iterations of 0.2 ms, while a multigrid solve pays 2 ms for the
  // dense coarsest inverse plus 0.75 ms per iteration (measured from the same states and lambdas: 1.0-8.6
  // ms against 7.5-35 ms per LM iteration, chi2 equal to the last digit; scripts/gpu_easy_solves.py).  A
  // solve with lambda >= bj_gate therefore starts with block-Jacobi; after 8 iterations the observed
  // reduction says how many it would need, and beyond `bj_budget` the solve starts again with the
  // hierarchy.  The gate follows the outcomes (deterministic: same decisions in every run).
  bool adaptive_prec = true;
  double bj_gate = -1.0;   // lambda from which block-Jacobi is tried first (< 0: 0.05 x mean |H_dd|)
  int bj_budget = 48;      // predicted iterations above which the probe is abandoned
  int n_bj_solves = 0, n_bj_abandoned = 0;
  bool trace_stale = true;
  double mean_diag = 0.0;
  int amg_status = 0;                  // first collective error inside a cycle
  std::string amg_err;
  // exact sparse block Cholesky (direct.hpp, direct_kernels.hpp): LinearSolverEigen's role on
  // graphs whose factorisation is cheap (KITTI-00 and other chain-like graphs)
  DirectPlan dplan;
  bool use_direct = false;
  LdlArgs ldl{};
  int ldl_wg_sub = LDL_WG_SUB;
  int fail_token = 1;  // number of the current exact solve (>= 2): see direct_solve
  std::vector<void*> direct_owned;
  // chi2 of the current estimates when it is already known (the last accepted trial computed it)
  bool chi_known = false;
  double chi_cache = 0.0;
  double last_true_rel = 0.0;  // ||r||_2 / ||b||_2 at the end of the last multigrid-preconditioned solve
  // hipGraph of `graph_iters` PCG iterations (single GPU, untimed runs): replayed per chunk
  hipGraphExec_t pcg_graph = nullptr;
  int pcg_graph_kind = -1;
  int graph_iters = PCG_GRAPH_ITERS;
  DevScalars* d_sc = nullptr;
  DevScalars* h_sc = nullptr;  // pinned
  StagedUploads staged;        // small uploads of init() go through one pinned block, on `stream`
  bool linearized = false;
  // multi-GPU row partition: this rank owns block rows [r0, r1); offs = 7 * row_begin
  Comm comm;
  int32_t r0 = 0, r1 = 0, e_lo = 0, e_hi = 0;
  std::vector<int32_t> row_begin;
  std::vector<int64_t> offs;
  // halo exchange (world > 1): boundary rows of all ranks, grouped by owner; this rank's share is
  // [halo_seg[rank], halo_seg[rank + 1]); halo_offs = 7 * halo_seg (doubles)
  bool use_halo = false;
  int32_t n_halo = 0, halo_slots = 0;  // boundary rows in all; slots per rank in the exchange buffer
  std::vector<int32_t> halo_seg;
  std::vector<int64_t> halo_offs;
  int32_t* d_brow = nullptr;
  double* d_halo = nullptr;
  // timing
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  // phase stamps of the LM loop (linearise | solve | update): recorded without waiting, read after
  // the trial's one host round trip (the chi2 fetch)
  hipEvent_t ev_ph[4] = {nullptr, nullptr, nullptr, nullptr};
  // per-iteration phase times (IterStats::ms_*): three event markers per LM trial, ~5.6 us of idle stream each --
  // nothing next to a 25 ms iteration, 7 % of a KITTI-00 one: measured on request (time_kernels, verbose) and
  // on systems of more than 4096 block rows, reported as 0 otherwise
  bool phase_timing = true;
  std::vector<hipEvent_t> pool;  // pairs (start, stop) for per-launch SpMV timing
  size_t pool_used = 0;
  sim3opt_kernel_times kt{};

  ~Engine() { release(); }

  void release() {
    // cached blocks are handed out again without the device-wide wait a hipFree implies
    if (stream) (void)hipStreamSynchronize(stream);
    void* ptrs[] = {d_states, d_backup, d_meas, d_ev0, d_ev1, d_hidx, d_active, d_info, d_kdelta,
                    d_rowptr, d_colidx, d_incptr, d_wrow, d_slot01, d_slot10, d_inc0, d_inc1, d_vals,
                    d_scratch, d_b, d_Minv, d_x, d_r, d_z, d_p, d_q, d_s, d_part_a, d_part_b, d_sc,
                    d_sub_first, d_sub_cnt, d_Gm, d_brow, d_halo, d_ptab};
    for (void* p : ptrs)
      if (p) dev_free(p);
    for (void* p : amg_owned)
      if (p) dev_free(p);
    amg_owned.clear();
    for (void* p : direct_owned)
      if (p) dev_free(p);
    direct_owned.clear();
    staged.release();
    if (h_sc) host_free(h_sc);
    for (hipEvent_t e : pool) event_release(e);
    if (ev_a) event_release(ev_a);
    if (ev_b) event_release(ev_b);
    for (hipEvent_t& e : ev_ph) if (e) { event_release(e); e = nullptr; }
    if (pcg_graph) (void)hipGraphExecDestroy(pcg_graph);
    if (stream) stream_release(stream);  // (synchronised above; kept for the next engine on this device)
    comm.release();
  }

  sim3::Opts mopts() const { return sim3::Opts{opt.exp_eps, opt.small_rot_half, opt.fix_small_angle_b}; }

  EdgeArgs edge_args() const {
    return EdgeArgs{e_lo, e_hi, d_ev0, d_ev1, d_meas, has_info ? d_info : nullptr,
                    has_kernel ? d_kdelta : nullptr, d_states, mopts()};
  }

  int init(const HostGraph& g, const Structure& s, std::string& err) {
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
    if (comm.world > 1 && !std::getenv("SIM3OPT_NO_HALO")) {
      std::vector<int32_t> brow_list;
      boundary_rows(nb, s.rowptr.data(), s.colidx.data(), comm.world, row_begin.data(), brow_list, halo_seg);
      n_halo = (int32_t)brow_list.size();
      // every rank's segment of the exchange buffer has the same length (the largest boundary, short
      // ones padded with -1): the exchange is then ONE in-place ncclAllGather, like the whole-vector one
      halo_slots = 0;
      for (int r = 0; r < comm.world; ++r) halo_slots = std::max(halo_slots, halo_seg[r + 1] - halo_seg[r]);
      halo_offs.resize(comm.world + 1);
      for (int r = 0; r <= comm.world; ++r) halo_offs[r] = 7 * (int64_t)halo_slots * r;
      // (worth it while the boundary is a fraction of the vector; a partition in insertion order of a
      // graph without locality has nearly every row on it: the plain all-gather is cheaper then)
      use_halo = n_halo > 0 && (int64_t)halo_slots * comm.world * 2 < nb;
      if (use_halo) {
        std::vector<int32_t> padded((size_t)halo_slots * comm.world, -1);
        for (int r = 0; r < comm.world; ++r)
          std::copy(brow_list.begin() + halo_seg[r], brow_list.begin() + halo_seg[r + 1],
                    padded.begin() + (size_t)halo_slots * r);
        HIPCHK(upload(staged, stream, d_brow, padded));
        HIPCHK(dev_malloc((void**)&d_halo, sizeof(double) * 7 * padded.size()));
        HIPCHK(hipMemset(d_halo, 0, sizeof(double) * 7 * padded.size()));
      }
      if (opt.verbose)
        std::fprintf(stderr, "sim3opt: rank %d of %d: rows [%d, %d) of %d, %d boundary rows in all (%.1f %%): %s\n",
                     comm.rank, comm.world, r0, r1, nb, n_halo, 100.0 * n_halo / std::max(1, nb),
                     use_halo ? "halo exchange" : "whole-vector all-gather");
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
    HIPCHK(dev_malloc((void**)&d_vals, sizeof(double) * 49 * (size_t)nnzb));
    HIPCHK(hipMemset(d_vals, 0, sizeof(double) * 49 * (size_t)nnzb));
    const size_t ninc = (size_t)s.incptr[nb];
    HIPCHK(dev_malloc((void**)&d_scratch, sizeof(double) * 35 * std::max<size_t>(ninc, 1)));
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

  int fetch_scalars(std::string& err) {
    HIPCHK(hipMemcpyAsync(h_sc, d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (comm.timing && comm.ev_used) return comm.drain(err);
    return SIM3OPT_OK;
  }

  // ---- timing helpers ----
  int timed_begin(std::string& err) {
    HIPCHK(hipEventRecord(ev_a, stream));
    return SIM3OPT_OK;
  }
  int timed_end(double& ms_acc, std::string& err) {
    HIPCHK(hipEventRecord(ev_b, stream));
    HIPCHK(hipEventSynchronize(ev_b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev_a, ev_b));
    ms_acc += ms;
    return SIM3OPT_OK;
  }
  int pool_get(hipEvent_t& a, hipEvent_t& b, std::string& err) {
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
  // after a stream sync: fold the recorded SpMV event pairs into the accumulators
  // (h_sc must be fresh).  Launches enqueued after the solve finished return at once; they are
  // left out of the launch count -- their few microseconds stay in the sum, so the average errs on
  // the slow side -- otherwise the per-launch figure would be flattered by up to pcg_check_every - 1
  // empty launches per solve.
  long long spmv_work_seen = 0;
  int pool_drain(std::string& err) {
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
    return SIM3OPT_OK;
  }

  // ---- aggregation multigrid ----
  template <typename T>
  int amg_up(T*& dptr, const std::vector<T>& h, std::string& err) {
    HIPCHK(dev_malloc((void**)&dptr, sizeof(T) * std::max<size_t>(h.size(), 1)));
    amg_owned.push_back(dptr);
    if (!h.empty()) HIPCHK(staged.put(dptr, h.data(), sizeof(T) * h.size(), stream));
    return SIM3OPT_OK;
  }
  int amg_alloc(double*& dptr, size_t count, std::string& err) {
    HIPCHK(dev_malloc((void**)&dptr, sizeof(double) * std::max<size_t>(count, 1)));
    amg_owned.push_back(dptr);
    HIPCHK(hipMemset(dptr, 0, sizeof(double) * std::max<size_t>(count, 1)));
    return SIM3OPT_OK;
  }

  // structure of the hierarchy (once per initialize); leaves use_amg false when the graph does
  // not coarsen (block-Jacobi is used then)
  std::vector<AmgLevelHost> amg_host;  // kept between amg_init and amg_bind
  int amg_init(const Structure& s, bool automatic, std::string& err) {
    (void)err;
    if (const char* ev = std::getenv("SIM3OPT_AMG_OMEGA")) amg_omega = std::max(0.1, std::min(0.95, std::atof(ev)));
    // cycle (measured): multiplicative on level 0, level 1 once and deeper levels three
    // times per visit; the additive level-0 form is a knob (about as fast on config 3, less robust
    // on ill-conditioned chains)
    amg_additive = false;
    // round 2: with FP32 block copies the coarse levels are cheap enough for two visits of level 1
    for (int l = 0; l <= AMG_MAX_LEVELS; ++l) amg_visits[l] = l <= 1 ? 2 : 3;
    amg_visits[0] = 1;
    if (const char* ev = std::getenv("SIM3OPT_AMG_CYCLE")) {  // e.g. "122": visits of levels 1, 2, 3...
      int last = 1;
      for (int l = 1; l <= AMG_MAX_LEVELS; ++l) {
        if ((int)std::strlen(ev) >= l && ev[l - 1] >= '1' && ev[l - 1] <= '3') last = ev[l - 1] - '0';
        amg_visits[l] = last;
      }
    }
    if (const char* ev = std::getenv("SIM3OPT_AMG_ADDITIVE")) amg_additive = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("SIM3OPT_AMG_FP32")) amg_fp32 = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("SIM3OPT_ADAPTIVE_PREC")) adaptive_prec = std::atoi(ev) != 0;
    if (!automatic) adaptive_prec = false;  // (a caller who names the multigrid gets the multigrid)
    if (const char* ev = std::getenv("SIM3OPT_AMG_PIVOT")) amg_pivot = std::atoi(ev) >= 28 ? 28 : 14;
    // measured on config 3: 1.8 into level 0 and 1.6 below cut the PCG iterations of
    // a solve from 56 to 43 (cycle 1/3) and from 29 to 25 (cycle 2/3); 2.0 (the limit for an exact
    // coarse solve) is no better
    for (int l = 0; l <= AMG_MAX_LEVELS; ++l) amg_over_l[l] = l == 0 ? 1.8 : 1.6;
    if (const char* ev = std::getenv("SIM3OPT_AMG_OVER")) {  // "a0[,a1[,a2...]]": last value repeats
      double last = 1.0;
      const char* p = ev;
      for (int l = 0; l <= AMG_MAX_LEVELS; ++l) {
        if (p && *p) {
          last = std::max(0.5, std::min(3.0, std::atof(p)));
          p = std::strchr(p, ',');
          if (p) ++p;
        }
        amg_over_l[l] = last;
      }
    }
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

  int amg_bind(const Structure& s, std::string& err) {
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
  int amg_setup(std::string& err) {
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
  void amg_prepare(double lambda) {
    const int nl = (int)amg.size();
    for (int l = 0; l < nl; ++l) {
      const AmgLevel& L = amg[l];
      const int lo = l == 0 ? r0 : 0, hi = l == 0 ? r1 : L.nb;  // level 0 is row-partitioned
      hipLaunchKernelGGL(k_jacobi, dim3(std::max(1, (hi - lo + WG - 1) / WG)), dim3(WG), 0, stream, lo, hi,
                         L.rowptr, L.vals, lambda, L.Minv, d_sc, l == 0 && amg_additive ? 1.0 : amg_omega,
                         L.diagH, L.W, l > 0 ? L.vals32 : (float*)nullptr);
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
  void spmv_mode(const AmgLevel& L, int mode, int level, const double* v, double* out,
                 const double* rvec, const double* xc = nullptr) {
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

  void amg_restrict(int l, const double* t) {  // r_{l+1} = P^T t, x_{l+1} = Minv r_{l+1}
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
  void amg_prolong(int l, const double* xc, const double* xin, double* xout) {
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
  const double* amg_coarse(int l) {
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
  double* amg_cycle(int l, double* cur, double* other) {
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
  int amg_apply(std::string& err) {
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

  // ---- exact sparse block Cholesky ----
  template <typename T>
  int direct_up(const T*& dptr, const std::vector<T>& h, std::string& err) {
    T* p = nullptr;
    HIPCHK(dev_malloc((void**)&p, sizeof(T) * std::max<size_t>(h.size(), 1)));
    direct_owned.push_back(p);
    if (!h.empty()) HIPCHK(staged.put(p, h.data(), sizeof(T) * h.size(), stream));
    dptr = p;
    return SIM3OPT_OK;
  }
  int direct_alloc(double*& dptr, size_t count, std::string& err) {
    HIPCHK(dev_malloc((void**)&dptr, sizeof(double) * std::max<size_t>(count, 1)));
    direct_owned.push_back(dptr);
    HIPCHK(hipMemset(dptr, 0, sizeof(double) * std::max<size_t>(count, 1)));
    return SIM3OPT_OK;
  }

  // plan (host, once per initialize) + buffers; leaves use_direct false when the factorisation
  // would be too expensive (the PCG takes over) unless the caller insists
  int direct_init(const Structure& s, std::string& err) {
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
    if (const char* ev = std::getenv("SIM3OPT_DIRECT_MAX_PAIRS")) max_pairs = std::atoll(ev);  // tuning knobs
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
  int direct_solve(double lambda, std::string& err) {
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

  // every rank's copy of `vec` gets the entries of the rows its own rows' blocks refer to: the boundary
  // rows only (halo exchange) where the partition has locality, the whole vector otherwise
  int exchange_rows(double* vec, std::string& err) {
    if (!use_halo) return comm.allgatherv(vec, offs, stream, err);
    const int k0 = halo_slots * comm.rank, k1 = k0 + halo_slots, nslots = halo_slots * comm.world;
    hipLaunchKernelGGL(k_halo_pack, dim3((7 * halo_slots + WG - 1) / WG), dim3(WG), 0, stream, k0, k1,
                       (const int32_t*)d_brow, (const double*)vec, d_halo);
    int rc = comm.allgatherv(d_halo, halo_offs, stream, err);
    if (rc) return rc;
    hipLaunchKernelGGL(k_halo_unpack, dim3((7 * nslots + WG - 1) / WG), dim3(WG), 0, stream, nslots, k0, k1,
                       (const int32_t*)d_brow, (const double*)d_halo, vec);
    return SIM3OPT_OK;
  }

  // ---- building blocks ----
  // scale_parts > 0: d_part_b holds that many partial sums of the trial's scale (k_scale): summed in the
  // same launch as chi2's
  int chi2(double* out, std::string& err, hipEvent_t before_fetch = nullptr, int scale_parts = 0) {
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

  // the perturbation table of the numeric Jacobians, re-evaluated when delta or the arithmetic options change
  Sim3* d_ptab = nullptr;
  double ptab_delta = 0.0;
  sim3::Opts ptab_opts{0.0, -1, -1};

  int linearize(std::string& err) {
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
    if (use_direct) {  // the factorisation's starting blocks: H in the layout of L, b permuted
      ldl.vals = d_vals;
      ldl.b = d_b;
      hipLaunchKernelGGL(k_ldl_gather, dim3(std::max(1, std::min(1024, (ldl.nL + 3) / 4))), dim3(WG), 0,
                         stream, ldl);
      HIPCHK(hipGetLastError());
    }
    linearized = true;
    amg_stale = true;
    trace_stale = true;
    kt.n_linearize += 1;
    return SIM3OPT_OK;
  }

  // SpMV variant (tuning knob, env SIM3OPT_SPMV="chunk,nt"; defaults chosen by measurement,
  // scripts/gpu_spmv_ab.py: 8 blocks per pipeline step, non-temporal block stream)
  int spmv_chunk = 8, spmv_nt = 1;

  int spmv_grid() const { return span_grid; }

  // q = (H + lambda I) v; partials of v.q in d_part_a and, with rvec, of rvec.v in d_part_b
  // With a start/stop event pair the dispatch itself is timestamped (hipExtLaunchKernelGGL):
  // no extra barrier packets, so the figure agrees with rocprofv3's kernel trace.
  void spmv_raw(double lambda, const double* v, double* q, const double* rvec, DevScalars* scp,
                hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    const int g = spmv_grid();
#define SPAN_CASE(CH, NTV)                                                                       \
  hipExtLaunchKernelGGL((k_spmv_span<CH, NTV, 0>), dim3(g), dim3(WG), 0, stream, ev0, ev1, 0, nb, \
                        d_wrow, d_rowptr, d_colidx, d_vals, v, q, lambda, d_part_a, rvec,         \
                        d_part_b, scp, (const double*)nullptr, 1, (const int32_t*)nullptr, 1.0)
#define SPAN_PLAIN(CH, NTV)                                                                     \
  hipLaunchKernelGGL((k_spmv_span<CH, NTV, 0>), dim3(g), dim3(WG), 0, stream, nb, d_wrow,       \
                     d_rowptr, d_colidx, d_vals, v, q, lambda, d_part_a, rvec, d_part_b, scp,     \
                     (const double*)nullptr, 1, (const int32_t*)nullptr, 1.0)
    if (!ev0) {  // plain launch: capturable into a hipGraph
      if (spmv_chunk <= 4) { if (spmv_nt) SPAN_PLAIN(4, true); else SPAN_PLAIN(4, false); }
      else { if (spmv_nt) SPAN_PLAIN(8, true); else SPAN_PLAIN(8, false); }
      return;
    }
    if (spmv_chunk <= 4) { if (spmv_nt) SPAN_CASE(4, true); else SPAN_CASE(4, false); }
    else { if (spmv_nt) SPAN_CASE(8, true); else SPAN_CASE(8, false); }
#undef SPAN_PLAIN
#undef SPAN_CASE
  }

  int spmv_launch(double lambda, const double* z, const double* rv, std::string& err) {  // the PCG's SpMV: w = A z, w.z (and r.z)
    hipEvent_t a = nullptr, b = nullptr;
    if (opt.time_kernels) {
      int rc = pool_get(a, b, err);
      if (rc) return rc;
    }
    spmv_raw(lambda, z, d_q, rv, d_sc, a, b);
    return SIM3OPT_OK;
  }

  // Preconditioned CG on (H + lambda I) x = b in the single-reduction form (k_pcg_step); the
  // result stays in d_x.  Two launches and one reduction point per iteration; the host only polls
  // a 100-byte struct every `pcg_check_every` iterations.
  int agree_on_fail(std::string& err) {  // multi-GPU: fail on any rank = fail on all
    hipLaunchKernelGGL(k_fail_to_double, dim3(1), dim3(1), 0, stream, d_sc);
    int rc = comm.allreduce(&d_sc->tmp_pq, 1, 1, stream, err);
    if (rc) return rc;
    hipLaunchKernelGGL(k_double_to_fail, dim3(1), dim3(1), 0, stream, d_sc);
    return SIM3OPT_OK;
  }

  int pcg(double lambda, int32_t* iters, double* rel_res, bool* ok, std::string& err) {
    if (use_direct) {  // exact step; `ok` is settled later from d_sc->fail (see optimize)
      *iters = 0;
      *rel_res = 0.0;
      *ok = true;
      return direct_solve(lambda, err);
    }
    if (use_amg && adaptive_prec) {
      // damping-dominated system?  (see adaptive_prec above)
      if (trace_stale) {
        int rc = fetch_scalars(err);
        if (rc) return rc;
        mean_diag = n > 0 ? h_sc->trace / (double)n : 0.0;
        trace_stale = false;
      }
      const double gate = bj_gate >= 0.0 ? bj_gate : 0.05 * mean_diag;
      if (mean_diag > 0.0 && lambda >= gate) {
        bool abandoned = false;
        int rc = pcg_attempt(lambda, 0, iters, rel_res, ok, nullptr, err, bj_budget, &abandoned);
        if (rc) return rc;
        if (opt.verbose >= 2)
          std::fprintf(stderr, "  lambda %.3g >= %.3g (mean |H_dd| %.3g): block-Jacobi first: %s after %d iterations\n",
                       lambda, gate, mean_diag, abandoned ? "abandoned" : "done", *iters);
        if (!abandoned) {
          ++n_bj_solves;
          if (*ok && *iters <= bj_budget / 4) bj_gate = std::min(gate, 0.5 * lambda);
          else bj_gate = std::min(gate, lambda);
          return SIM3OPT_OK;
        }
        ++n_bj_abandoned;
        bj_gate = 2.0 * lambda;  // not before the damping has doubled
      }
    }
    if (use_amg || use_chain) {
      // the block-tridiagonal factorisation (or the multigrid's coarsest-level inverse) can meet a
      // non-positive pivot when H is numerically semi-definite (cond ~1e12 in the reference's
      // as-written arithmetic): retry with block-Jacobi
      bool broke = false;
      int rc = pcg_attempt(lambda, use_amg ? 2 : 1, iters, rel_res, ok, &broke, err);
      if (rc) return rc;
      // A CG breakdown (r.z < 0, p.Ap <= 0) or a residual that is not small although the M^-1 norm
      // says so, with the over-corrected cycle: the over-correction is safe only while the (inexact)
      // coarse solves stay within (0, 2) of the exact ones -- measured on config 3: 1.8 / 1.6 always,
      // 1.9 / 1.7 not.  Before blaming the system (and making LM reject the trial), solve again with
      // the plain cycle; keep it if that was the cure.
      if (use_amg && !broke && amg_over_on && (!*ok || last_true_rel > 1e-3)) {
        if (opt.verbose)
          std::fprintf(stderr, "sim3opt: multigrid PCG broke down (ok %d, ||r||/||b|| %.1e): again without over-correction\n",
                       (int)*ok, last_true_rel);
        amg_over_on = false;
        pcg_graph_kind = -1;  // (a captured iteration has the factors baked into its launches)
        rc = pcg_attempt(lambda, 2, iters, rel_res, ok, &broke, err);
        if (rc) return rc;
        if (!*ok) {  // not the preconditioner's fault: the system is not positive definite
          amg_over_on = true;
          pcg_graph_kind = -1;
        }
      }
      if (!broke) return SIM3OPT_OK;
    }
    return pcg_attempt(lambda, 0, iters, rel_res, ok, nullptr, err);
  }

  // prec: 0 block-Jacobi, 1 chain segments, 2 aggregation multigrid
  // probe_budget > 0 (block-Jacobi tried first on a damping-dominated system): after 8 iterations the
  // reduction reached so far predicts the total; if that exceeds the budget -- or the budget runs out --
  // *abandoned is set and the caller solves again with the hierarchy
  int pcg_attempt(double lambda, int prec, int32_t* iters, double* rel_res, bool* ok,
                  bool* chain_broke, std::string& err, int probe_budget = 0, bool* abandoned = nullptr) {
    const bool use_chain = prec == 1, use_mg = prec == 2;
    const bool probe = probe_budget > 0;
    double* const zin = use_mg ? d_az : d_z;  // preconditioned residual the PCG consumes
    // r.z: from the SpMV's own pass over r -- or, with the multiplicative multigrid cycle, from the
    // cycle's last kernel, which holds r and writes z (the SpMV then skips its load of r)
    const double* const spmv_r = use_mg && !amg_additive ? nullptr : d_r;
    const int nloc = r1 - r0;
    const int gj = std::max(1, (nloc + WG - 1) / WG);
    const int gv = grid_for((nloc + 8) / 9, 4);  // 36 block rows per workgroup pass
    const int gs = spmv_grid();
    const bool multi = comm.active();
    // [w.z, r.z] summed once by k_final_sum2 (multi-GPU: then all-reduced) instead of by every
    // workgroup of the PCG step when the SpMV leaves more partials than a workgroup sums for free
    const bool pre_sum = multi || gs > MAX_GRID;
    const double* scal = pre_sum ? &d_sc->tmp_pq : nullptr;
    // automatic cap: small systems may need ~n iterations for an (almost) exact step like the
    // reference's Cholesky (chains are ill-conditioned); large ones get a truncated-Newton budget
    // (round 3: a cap of 4000 for the multigrid path was tried for the one system in twenty of the
    // as-written arithmetic on config 3 that stops at 1000 -- it stops at 4000 as well, relative residual
    // 2e-3: numerically indefinite without a detectable breakdown; the cap stays)
    int max_it = opt.pcg_max_iters > 0 ? opt.pcg_max_iters
                                       : (n <= 50000 ? std::max(100, 2 * n) : 1000);
    if (probe) max_it = std::min(max_it, probe_budget);
    const int nseg = (nloc + chain_seg - 1) / chain_seg;
    const int gc = grid_for(nseg, 4);  // chain apply: one wavefront per segment
    const double* Minv_arg = use_chain ? nullptr : d_Minv;
    int rc = SIM3OPT_OK;
    h_sc->rz[0] = h_sc->rz[1] = h_sc->alpha[0] = h_sc->alpha[1] = h_sc->rz0 = 0.0;
    h_sc->iter = 0;
    h_sc->max_iter = max_it;
    h_sc->done = h_sc->stop = h_sc->fail = 0;
    h_sc->tol2 = opt.pcg_rel_tol * opt.pcg_rel_tol;
    h_sc->lambda = lambda;
    // chi2 / scale / maxdiag live in the same struct: only the PCG fields are reset
    HIPCHK(hipMemcpyAsync(&d_sc->rz[0], &h_sc->rz[0], offsetof(DevScalars, chi2), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(&d_sc->iter, &h_sc->iter, offsetof(DevScalars, tmp_pq) - offsetof(DevScalars, iter),
                          hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(&d_sc->lambda, &h_sc->lambda, sizeof(double), hipMemcpyHostToDevice, stream));
    if (use_mg) {
      if (amg_stale) {
        rc = amg_setup(err);
        if (rc) return rc;
      }
      amg_prepare(lambda);
    } else if (use_chain) {
      hipLaunchKernelGGL(k_chain_factor, dim3(std::max(1, (nseg + 63) / 64)), dim3(64), 0, stream,
                         r0, r1, chain_seg, d_rowptr, d_vals, d_sub_first, d_sub_cnt, lambda,
                         d_Minv, d_Gm, d_sc);
    } else {
      hipLaunchKernelGGL(k_jacobi, dim3(gj), dim3(WG), 0, stream, r0, r1, d_rowptr, d_vals, lambda,
                         d_Minv, d_sc, 1.0, (const double*)nullptr, (const double*)nullptr);
    }
    hipLaunchKernelGGL(k_pcg_init, dim3(gv), dim3(WG), 0, stream, r0, r1, d_b, Minv_arg, d_x, d_r,
                       d_z, d_p, d_s);
    if (use_chain || use_mg) {
      if (multi) {
        rc = agree_on_fail(err);
        if (rc) return rc;
      }
      rc = fetch_scalars(err);  // did the factorisation succeed?
      if (rc) return rc;
      if (h_sc->fail) {
        if (opt.verbose)
          std::fprintf(stderr, "sim3opt: %s set-up met a non-positive pivot (lambda %.3g): block-Jacobi for this solve\n",
                       use_mg ? "multigrid" : "chain", lambda);
        if (chain_broke) *chain_broke = true;
        *ok = false;
        *iters = 0;
        *rel_res = 0.0;
        return SIM3OPT_OK;
      }
      if (use_chain)
        hipLaunchKernelGGL(k_chain_apply, dim3(gc), dim3(WG), 0, stream, r0, r1, chain_seg, d_Minv,
                           d_Gm, d_r, d_z, (const DevScalars*)nullptr);
      else {
        rc = amg_apply(err);
        if (rc) return rc;
      }
    }
    HIPCHK(hipGetLastError());
    if (multi && !use_mg) {  // (the multigrid cycle gathers its own operands)
      rc = exchange_rows(d_z, err);
      if (rc) return rc;
    }
    // (a multigrid iteration is ~1 ms of GPU work and its coarse launches run even after `done`:
    // poll more often)
    const int chunk = use_mg ? std::min(4, std::max(1, opt.pcg_check_every))
                             : (probe ? 8 : std::max(1, opt.pcg_check_every));
    int it = 0, par = 0;
    // Launch-bound regime (small graphs: two ~3 us kernels per iteration): replay a captured
    // hipGraph of PCG_GRAPH_ITERS iterations instead of enqueueing them one by one.  The first
    // iteration stays eager (it carries it == 0); captured steps read the counter, the damping and
    // the stopping state from DevScalars, so one instantiated graph serves every solve.
    const bool graphed = !multi && !opt.time_kernels && opt.pcg_graph && max_it > PCG_GRAPH_ITERS && !probe;
    if (graphed && (!pcg_graph || pcg_graph_kind != prec)) {
      if (pcg_graph) { (void)hipGraphExecDestroy(pcg_graph); pcg_graph = nullptr; }
      // a multigrid iteration is ~20 launches: shorter graphs waste fewer no-op launches after
      // convergence
      graph_iters = use_mg ? 4 : PCG_GRAPH_ITERS;
      hipGraph_t gr = nullptr;
      HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      for (int c = 0; c < graph_iters; ++c) {
        spmv_raw(lambda, zin, d_q, spmv_r, d_sc);
        if (pre_sum)
          hipLaunchKernelGGL(k_final_sum2, dim3(1), dim3(WG), 0, stream, d_part_a, d_part_b, gs,
                             &d_sc->tmp_pq);
        hipLaunchKernelGGL(k_pcg_step, dim3(gv), dim3(WG), 0, stream, r0, r1, (1 + c) & 1, -1,
                           scal, d_part_a, d_part_b, gs, Minv_arg,
                           (const double*)zin, d_z, d_q, d_p, d_s, d_x, d_r, d_sc);
        if (use_chain)
          hipLaunchKernelGGL(k_chain_apply, dim3(gc), dim3(WG), 0, stream, r0, r1, chain_seg,
                             d_Minv, d_Gm, d_r, d_z, (const DevScalars*)d_sc);
        if (use_mg) (void)amg_apply(err);  // single GPU here: no collectives inside
      }
      {  // (a failed launch inside the region must not leave the stream capturing)
        const hipError_t le = hipGetLastError();
        const hipError_t ce = hipStreamEndCapture(stream, &gr);
        if (le != hipSuccess || ce != hipSuccess) {
          if (gr) (void)hipGraphDestroy(gr);
          err = std::string("PCG graph capture: ") + hipGetErrorString(le != hipSuccess ? le : ce);
          return SIM3OPT_ERR_HIP;
        }
      }
      HIPCHK(hipGraphInstantiate(&pcg_graph, gr, nullptr, nullptr, 0));
      (void)hipGraphDestroy(gr);
      pcg_graph_kind = prec;
    }
    for (;;) {
      rc = fetch_scalars(err);
      if (rc) return rc;
      if (opt.time_kernels) {
        rc = pool_drain(err);
        if (rc) return rc;
      } else {
        spmv_work_seen = h_sc->n_spmv_work;
      }
      if (h_sc->done || h_sc->stop || h_sc->fail || it >= max_it) break;
      if (probe && it >= 8 && h_sc->rz0 > 0.0) {
        // squared M^-1-norm reduction after `it` iterations -> iterations to the tolerance at that rate
        const double ratio = std::fabs(h_sc->gam_last) / h_sc->rz0;
        const double need = ratio > 0.0 && ratio < 1.0 ? it * std::log(h_sc->tol2) / std::log(ratio) : 1e30;
        if (need > probe_budget) break;
      }
      if (graphed && it > 0 && par == 1 && max_it - it >= graph_iters) {
        // steps past max_iter cannot happen: the step that reaches it raises `stop`, and the
        // following launches of the replay are no-ops
        const int reps = std::max(1, std::min(chunk, max_it - it) / graph_iters);
        for (int k = 0; k < reps; ++k) HIPCHK(hipGraphLaunch(pcg_graph, stream));
        it += reps * graph_iters;
        continue;
      }
      const int todo = graphed && it == 0 ? 1 : std::min(chunk, max_it - it);
      for (int c = 0; c < todo; ++c) {
        rc = spmv_launch(lambda, zin, spmv_r, err);
        if (rc) return rc;
        if (pre_sum)  // [w.z, r.z] -> tmp_pq, tmp_rz (adjacent)
          hipLaunchKernelGGL(k_final_sum2, dim3(1), dim3(WG), 0, stream, d_part_a, d_part_b, gs,
                             &d_sc->tmp_pq);
        if (multi) {  // one 2-double all-reduce
          rc = comm.allreduce(&d_sc->tmp_pq, 2, 0, stream, err);
          if (rc) return rc;
        }
        hipLaunchKernelGGL(k_pcg_step, dim3(gv), dim3(WG), 0, stream, r0, r1, par, it, scal,
                           d_part_a, d_part_b, gs, Minv_arg, (const double*)zin, d_z, d_q, d_p, d_s,
                           d_x, d_r, d_sc);
        if (use_chain)
          hipLaunchKernelGGL(k_chain_apply, dim3(gc), dim3(WG), 0, stream, r0, r1, chain_seg,
                             d_Minv, d_Gm, d_r, d_z, (const DevScalars*)d_sc);
        if (use_mg) {
          rc = amg_apply(err);
          if (rc) return rc;
        }
        if (multi && !use_mg) {  // the next SpMV gathers z from the neighbouring ranks
          rc = exchange_rows(d_z, err);
          if (rc) return rc;
        }
        par ^= 1;
        ++it;
      }
      HIPCHK(hipGetLastError());
    }
    if (probe && abandoned && !h_sc->done && !h_sc->fail) {  // (ran out of budget or predicted to)
      *abandoned = true;
      kt.n_pcg_vec += h_sc->iter;
      *iters = h_sc->iter;
      *rel_res = h_sc->rz0 > 0 ? std::sqrt(std::fabs(h_sc->gam_last) / h_sc->rz0) : 0.0;
      *ok = true;
      return SIM3OPT_OK;
    }
    if (multi) {  // every rank updates its replica of all estimates
      rc = comm.allgatherv(d_x, offs, stream, err);
      if (rc) return rc;
      rc = agree_on_fail(err);
      if (rc) return rc;
      rc = fetch_scalars(err);
      if (rc) return rc;
    }
    last_true_rel = 0.0;
    if (use_mg && !h_sc->fail) {
      // The stopping test is in the M^-1 norm.  A multigrid cycle is symmetric by construction but
      // positive definite only within limits (over-correction, inexact coarse solves): should it
      // ever lose definiteness, r.z can vanish while r has not.  So the 2-norm of the (recursive)
      // residual is checked against ||b|| once per solve: two more small launches and one read-back.
      const int gn = grid_for(7 * (int64_t)nloc, WG);
      hipLaunchKernelGGL(k_norms2, dim3(gn), dim3(WG), 0, stream, 7 * r0, 7 * r1, d_r, d_b, d_part_a, d_part_b);
      hipLaunchKernelGGL(k_final_sum2, dim3(1), dim3(WG), 0, stream, d_part_a, d_part_b, gn, &d_sc->tmp_pq);
      HIPCHK(hipGetLastError());
      if (multi) {
        rc = comm.allreduce(&d_sc->tmp_pq, 2, 0, stream, err);
        if (rc) return rc;
      }
      rc = fetch_scalars(err);
      if (rc) return rc;
      last_true_rel = h_sc->tmp_rz > 0 ? std::sqrt(h_sc->tmp_pq / h_sc->tmp_rz) : 0.0;
      if (opt.verbose)
        std::fprintf(stderr, "sim3opt: multigrid PCG: %d iterations, ||r||_Minv ratio %.2e, ||r||_2 / ||b||_2 %.2e\n",
                     h_sc->iter, h_sc->rz0 > 0 ? std::sqrt(std::fabs(h_sc->gam_last) / h_sc->rz0) : 0.0, last_true_rel);
    }
    kt.n_pcg_vec += h_sc->iter;
    *iters = h_sc->iter;
    // r.z seen by the last executed step, i.e. of the residual BEFORE that step's update
    *rel_res = h_sc->rz0 > 0 ? std::sqrt(std::fabs(h_sc->gam_last) / h_sc->rz0) : 0.0;
    *ok = !h_sc->fail;
    return SIM3OPT_OK;
  }

  int optimize(int32_t max_iters, std::vector<sim3opt_iter_stats>& stats, std::string& err) {
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
        rc = pcg(lambda, &pit, &rres, &ok2, err);
        if (rc) return rc;
        if (phase_timing) HIPCHK(hipEventRecord(ev_ph[2], stream));
        T.pcg_iters += pit;
        T.pcg_rel_res = rres;
        if (opt.verbose >= 2)
          std::fprintf(stderr, "  trial %d: lambda %.6g, %d PCG iterations (rel %.2e)\n", qmax, lambda, pit, rres);
        double scale = 0.0;
        if (ok2) {
          hipLaunchKernelGGL(k_oplus, dim3((nv + WG - 1) / WG), dim3(WG), 0, stream, nv, d_hidx,
                             d_x, d_states, mopts(), use_direct ? (const DevScalars*)d_sc : nullptr, d_backup,
                             fail_token);
          const int ge = grid_for(7 * (int64_t)(r1 - r0), WG);
          hipLaunchKernelGGL(k_scale, dim3(ge), dim3(WG), 0, stream, 7 * r0, 7 * r1, d_x, d_b,
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
};

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
  const int dev = e->opt.device;
  e->opt = opt;
  e->opt.device = dev;
  e->comm.timing = opt.time_kernels != 0;
  return SIM3OPT_OK;
}

int engine_optimize(Engine* e, int32_t max_iters, std::vector<sim3opt_iter_stats>& stats,
                    std::string& err) {
  return e->optimize(max_iters, stats, err);
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
  return SIM3OPT_OK;
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
  if (values)
    HIPCHK(hipMemcpy(values, e->d_vals, sizeof(double) * 49 * (size_t)e->nnzb,
                     hipMemcpyDeviceToHost));
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

int engine_bench_spmv(Engine* e, int32_t reps, double* ms_mean, std::string& err) {
  if (!e->linearized) {
    err = "bench_spmv: call sim3opt_linearize (or optimize) first";
    return SIM3OPT_ERR_STATE;
  }
  // p = b as a representative dense vector
  HIPCHK(hipMemcpyAsync(e->d_p, e->d_b, sizeof(double) * (size_t)e->n, hipMemcpyDeviceToDevice,
                        e->stream));
  for (int i = 0; i < 3; ++i) e->spmv_raw(0.0, e->d_p, e->d_q, e->d_b, nullptr);
  HIPCHK(hipEventRecord(e->ev_a, e->stream));
  for (int i = 0; i < reps; ++i) e->spmv_raw(0.0, e->d_p, e->d_q, e->d_b, nullptr);
  HIPCHK(hipEventRecord(e->ev_b, e->stream));
  HIPCHK(hipEventSynchronize(e->ev_b));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e->ev_a, e->ev_b));
  *ms_mean = reps > 0 ? ms / reps : 0.0;
  return SIM3OPT_OK;
}

int engine_bench_stream(Engine* e, int32_t mode, int32_t reps, double* ms_mean, std::string& err) {
  const size_t n = (size_t)49 * (size_t)e->nnzb;
  const int g = 2048;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(k_stream_read<0>, dim3(g), dim3(WG), 0, e->stream, e->d_vals, n, e->d_q);
    else if (mode == 1) hipLaunchKernelGGL(k_stream_read<1>, dim3(g), dim3(WG), 0, e->stream, e->d_vals, n, e->d_q);
    else hipLaunchKernelGGL(k_stream_read<2>, dim3(g), dim3(WG), 0, e->stream, e->d_vals, n, e->d_q);
  };
  for (int i = 0; i < 3; ++i) launch();
  HIPCHK(hipEventRecord(e->ev_a, e->stream));
  for (int i = 0; i < reps; ++i) launch();
  HIPCHK(hipEventRecord(e->ev_b, e->stream));
  HIPCHK(hipEventSynchronize(e->ev_b));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e->ev_a, e->ev_b));
  *ms_mean = reps > 0 ? ms / reps : 0.0;
  return SIM3OPT_OK;
}

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

void engine_local_rows(const Engine* e, int32_t* begin, int32_t* end) {
  if (begin) *begin = e->r0;
  if (end) *end = e->r1;
}

int engine_preconditioner(const Engine* e) { return e->use_amg ? 2 : (e->use_chain ? 1 : 0); }

int engine_linear_solver(const Engine* e) { return e->use_direct ? 1 : 0; }

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
