// engine_impl.hpp -- the Engine class shared by the translation units of the device-resident LM:
//   engine.hip         initialisation, linearisation, chi2, the LM trial loop (OptimizationAlgorithmLevenberg::solve)
//   engine_pcg.hip     the preconditioned CG (LinearSolverEigen's role on graphs too large to factor), halo exchange
//   engine_amg.hip     the aggregation-multigrid preconditioner: set-up per linearisation / per trial, the cycle
//   engine_direct.hip  the exact sparse block Cholesky (LinearSolverEigen's role on KITTI-00-like graphs)
// Every kernel header belongs to ONE translation unit (lm_kernels.hpp -> engine.hip, pcg_kernels.hpp ->
// engine_pcg.hip, amg_kernels.hpp -> engine_amg.hip, direct_kernels.hpp -> engine_direct.hip); only the
// SpMV template (spmv_kernel.hpp) is shared.  A kernel another unit needs is reached through an Engine method.
#pragma once
// (formerly all of engine.hip) -- device-resident Levenberg-Marquardt on a Sim(3) pose graph, gfx950 (MI355X).
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

#ifndef SIM3OPT_COARSE_CH
#define SIM3OPT_COARSE_CH 8     // blocks per pipeline step of the coarse levels' passes (one-system cycle; tuning: 16)
#endif
#ifndef SIM3OPT_F32_CH
#define SIM3OPT_F32_CH 8        // blocks per pipeline step of the level-0 FP32 passes (tuning: 16)
#endif
constexpr int WG = 256;         // 4 wavefronts of 64
constexpr int KB = 4;            // right-hand sides the batched PCG solves together (engine_batch.hip)
constexpr int CHAIN_SEG_MAX = 256;  // rows per segment of the chain preconditioner (LDS of k_chain_apply)
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

#include "dev_common.hpp"
#include "direct_args.hpp"
struct EdgeArgs;  // lm_kernels.hpp

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
  // Arrays indexed by a GLOBAL position (block, incidence) of which a rank touches one contiguous range -- the
  // blocks / incidences of its own rows: H, its FP32 copy, the partitioned coarse levels, the assembly scratch --
  // are allocated for that range only (what makes N ranks hold N times the graph): the pointer the kernels index
  // is the virtual base `allocation - lo`; positions outside [lo, hi) are never dereferenced.
  // options.debug_full_arrays: the whole array, everything outside the range filled with 0xFF bytes (NaN as
  // float and as double); check_foreign_ranges() finds a write there, a read shows up as NaN in the results --
  // the test of the ranges (tests/test_distributed_gpu.py).  One rank: lo = 0, hi = total.
  struct RangedArray { void* alloc; size_t elem; int64_t total, lo, hi; };
  std::vector<RangedArray> ranged;
  template <typename T>
  int alloc_ranged(T*& virt, int64_t lo, int64_t hi, int64_t total, std::string& err) {
    total = std::max<int64_t>(total, 1);
    lo = std::max<int64_t>(0, std::min(lo, total));
    hi = std::max(lo, std::min(hi, total));
    T* p = nullptr;
    if (opt.debug_full_arrays) {
      HIPCHK(dev_malloc((void**)&p, sizeof(T) * (size_t)total));
      HIPCHK(hipMemset(p, 0xFF, sizeof(T) * (size_t)total));
      if (hi > lo) HIPCHK(hipMemset(p + lo, 0, sizeof(T) * (size_t)(hi - lo)));
      virt = p;
    } else {
      HIPCHK(dev_malloc((void**)&p, sizeof(T) * (size_t)std::max<int64_t>(hi - lo, 1)));
      HIPCHK(hipMemset(p, 0, sizeof(T) * (size_t)std::max<int64_t>(hi - lo, 1)));
      virt = reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) - sizeof(T) * (size_t)lo);
    }
    ranged.push_back({p, sizeof(T), total, lo, hi});
    return SIM3OPT_OK;
  }
  int check_foreign_ranges(std::string& err);  // (debug_full_arrays only; else a no-op)
  int64_t ranged_bytes() const {               // device bytes of the ranged arrays as allocated
    int64_t b = 0;
    for (const RangedArray& a : ranged) b += (int64_t)a.elem * (opt.debug_full_arrays ? a.total : std::max<int64_t>(a.hi - a.lo, 1));
    return b;
  }
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
    int32_t lo = 0, hi = 0;          // rows this rank's kernels work on: its own (partitioned level) or all
    int32_t own_lo = 0, own_hi = 0;  // rows whose Galerkin blocks / restricted residual this rank forms from the
    int64_t own_b0 = 0, own_b1 = 0;  //   level above (aggregates of its own rows there, or all) and their blocks
  };
  // spans (per rank) of a level's vectors / block values: set where a partitioned level meets a replicated one
  std::vector<std::vector<int64_t>> lvl_offs, lvl_blk_offs;
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
  // ---- several right-hand sides at once (engine_batch.hip): the rejected trials of an LM iteration ----
  struct BatchLevel {
    double *Minv = nullptr, *r = nullptr, *x = nullptr, *t = nullptr;  // KB systems each, strides ms / vs
    float* diag32 = nullptr;  // coarse levels: every system's damped diagonal blocks, [row][49]
    int64_t vs = 0, ms = 0;
  };
  std::vector<BatchLevel> blv;
  std::vector<void*> batch_owned;
  double *b_x = nullptr, *b_r = nullptr, *b_z = nullptr, *b_p = nullptr, *b_q = nullptr, *b_s = nullptr, *b_az = nullptr;
  double *b_Ainv = nullptr, *b_diag64 = nullptr, *b_part_a = nullptr, *b_part_b = nullptr;
  int64_t b_vs = 0, b_as = 0;
  DevScalars *d_bsc = nullptr, *h_bsc = nullptr;
  bool batch_ready = false;
  int b_nsys = KB;  // systems of the batch being solved (kernels are instantiated for 2, 3, 4)
  int64_t b_slice_blocks = 100000;  // levels with at most this many blocks run one system per grid slice
  int batch_alloc(std::string& err);
  void batch_release();
  void b_spmv_mode(int level, int mode, const double* v, double* out, const double* rvec, const double* xc);
  void b_restrict(int l, const double* t);
  double* b_coarse(int l);
  double* b_cycle(int l, double* cur, double* other);
  int pcg_batch(const double* lams, int nsys, int32_t* iters, double* rel_res, bool* capped, bool* usable,
                std::string& err);
  // most systems a batch may hold for this graph and these options (0: no batching)
  int batch_capacity() const {
    if (!use_amg || use_direct || comm.active() || !amg_fp32 || amg_additive || opt.pcg_batch == 1) return 0;
    return opt.pcg_batch > 1 ? std::min(opt.pcg_batch, KB) : KB;
  }
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
  bool last_capped = false;    // the last solve stopped at its iteration cap short of the tolerance
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
  // Row partition of a multigrid level over the ranks (level 0 = the LM system; coarse levels with more than
  // options.amg_shard_rows rows are partitioned too -- aggregates never straddle two ranks, so a rank's coarse
  // rows are the aggregates of its own fine rows).  Vectors of a partitioned level are full-length on every
  // rank; a kernel writes its own rows and reads its own rows plus the rows its blocks' columns name, which
  // exchange_level() refreshes: neighbour-only (grouped send / receive of exactly the rows the other side
  // reads) where the transport can, else the all-gather of the whole vector.
  struct LevelPart {
    int32_t lo = 0, hi = 0;               // rows of this rank
    std::vector<int32_t> row_begin;       // world + 1
    std::vector<int64_t> offs;            // 7 * row_begin (doubles): spans of the whole-vector all-gather
    std::vector<int64_t> blk_offs;        // 49 * rowptr[row_begin]: spans of the level's block values
    bool neighbour = false;               // the neighbour-only plan below is in use
    bool self_test = false;               // one rank, forced collectives: the plan sends a few rows to itself
    int32_t n_send = 0, n_recv = 0;       // rows this rank sends / receives per exchange
    int32_t *d_send = nullptr, *d_recv = nullptr;  // row lists, grouped by peer
    std::vector<int64_t> send_offs, recv_offs;     // world + 1, doubles, into the buffers
    double *d_sbuf = nullptr, *d_rbuf = nullptr;
  };
  std::vector<LevelPart> parts;  // parts[l] for l < n_sharded (world > 1 or forced collectives; else empty)
  int n_sharded = 0;             // multigrid levels [0, n_sharded) are partitioned, the rest replicated
  bool sharded(int l) const { return l < n_sharded; }
  int level_part_init(int l, int32_t nb_l, const int32_t* rowptr_l, const int32_t* colidx_l,
                      const std::vector<int32_t>& row_begin_l, std::string& err);
  // refreshes, on level l, this rank's copy of the foreign rows its own rows read (no-op on one rank)
  int exchange_level(int l, double* vec, std::string& err);
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
  std::vector<hipEvent_t> rep_pool;  // pairs around every visit of the first replicated multigrid level
  size_t rep_used = 0;
  int rep_level = 0;  // first multigrid level a partition over part_world() ranks replicates (0: no hierarchy)
  sim3opt_kernel_times kt{};

  ~Engine() { release(); }

  int device_used = -1;  // the device this engine lives on (init); release() runs under it
  void release();
  void release_under_device();

  // ranks of the row partition the multigrid hierarchy is built for: the communicator's, or -- on one rank --
  // options.amg_virtual_ranks (the hierarchy an N-rank run builds, for comparisons)
  int part_world() const { return comm.world > 1 ? comm.world : std::max(1, opt.amg_virtual_ranks); }
  sim3::Opts mopts() const { return sim3::Opts{opt.exp_eps, opt.small_rot_half, opt.fix_small_angle_b}; }

  EdgeArgs edge_args() const;

  int init(const HostGraph& g, const Structure& s, std::string& err);

  int fetch_scalars(std::string& err);

  // ---- timing helpers ----
  int timed_begin(std::string& err);
  int timed_end(double& ms_acc, std::string& err);
  int pool_get(hipEvent_t& a, hipEvent_t& b, std::string& err);
  // after a stream sync: fold the recorded SpMV event pairs into the accumulators
  // (h_sc must be fresh).  Launches enqueued after the solve finished return at once; they are
  // left out of the launch count -- their few microseconds stay in the sum, so the average errs on
  // the slow side -- otherwise the per-launch figure would be flattered by up to pcg_check_every - 1
  // empty launches per solve.
  long long spmv_work_seen = 0;
  int pool_drain(std::string& err);

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
  int amg_init(const Structure& s, bool automatic, std::string& err);

  int amg_bind(const Structure& s, std::string& err);

  // numbers of the hierarchy: once per linearisation (P = Ad(S_v) at the linearisation point)
  int amg_setup(std::string& err);

  // per trial: damped diagonal blocks, smoother inverses, dense inverse of the coarsest level
  void amg_prepare(double lambda);

  // mode 3 (coarse levels): mode 2 on v + xc[agg], the coarser level's correction prolonged on the fly
  void spmv_mode(const AmgLevel& L, int mode, int level, const double* v, double* out,
                 const double* rvec, const double* xc = nullptr);

  void amg_restrict(int l, const double* t);
  void amg_prolong(int l, const double* xc, const double* xin, double* xout);

  // Solves the level-(l+1) problem approximately (right-hand side amg[l+1].r, first iterate
  // amg[l+1].x = Minv r already there) by amg_visits[l+1] cycles; returns the buffer with the result.
  double* amg_coarse(int l);
  double* amg_coarse_body(int l);
  void amg_exchange(int l, double* vec);

  // One multigrid cycle on level l from the iterate `cur`; `other` is scratch; returns the buffer
  // that holds the new iterate (always `other`):
  //   t = r - A cur;  coarse correction;  cur += P x_c;  other = cur + Minv (r - A cur)
  double* amg_cycle(int l, double* cur, double* other);

  // d_az = M^-1 d_r; on entry d_z = Minv_0 d_r (written by the PCG step).  Multiplicative: one
  // V(1,1) (or W) cycle from that iterate.  Additive on level 0 (no fine-level matrix pass in the
  // preconditioner): M^-1 = D^-1 + P (coarse cycle) P^T.
  // Multi-GPU: level 0 is row-partitioned like the PCG (its matrix passes need the whole iterate:
  // one all-gather of d_z before, one of d_az after; the restricted residual is all-reduced), the
  // coarse levels are replicated and every rank runs the same coarse cycle.
  int amg_apply(std::string& err);

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
  int direct_init(const Structure& s, std::string& err);

  // (H + lambda I) x = b, exactly; x in d_x.  A non-positive pivot raises d_sc->fail (read by the
  // caller together with the trial's chi2: no extra round trip).
  int direct_solve(double lambda, std::string& err);
  void direct_gather();  // once per linearisation: H in the layout of L, b permuted (k_ldl_gather)
  // block-Jacobi inverses Minv = omega (D + lambda W)^-1 of rows [lo, hi) (k_jacobi; engine_pcg.hip)
  void jacobi(int lo, int hi, const int32_t* rowptr, double* vals, double lambda, double* Minv, double omega,
              const double* diagH, const double* W, float* vals32, DevScalars* sc = nullptr,
              double* diag64_out = nullptr, float* diag32_out = nullptr);
  void norms2(const double* r, const double* b, double* part_a, double* part_b, double* out2);  // engine_pcg.hip
  void dense_inverse(const double* diag64, double* Aout, DevScalars* sc);                         // engine_amg.hip

  // every rank's copy of `vec` gets the entries of the rows its own rows' blocks refer to: the boundary
  // rows only (halo exchange) where the partition has locality, the whole vector otherwise
  int exchange_rows(double* vec, std::string& err) { return exchange_level(0, vec, err); }

  // ---- building blocks ----
  // scale_parts > 0: d_part_b holds that many partial sums of the trial's scale (k_scale): summed in the
  // same launch as chi2's
  int chi2(double* out, std::string& err, hipEvent_t before_fetch = nullptr, int scale_parts = 0);

  // the perturbation table of the numeric Jacobians, re-evaluated when delta or the arithmetic options change
  Sim3* d_ptab = nullptr;
  double ptab_delta = 0.0;
  sim3::Opts ptab_opts{0.0, -1, -1};

  int linearize(std::string& err);

  // SpMV variant (tuning knob, env SIM3OPT_SPMV="chunk,nt"; defaults chosen by measurement,
  // scripts/gpu_spmv_ab.py: 8 blocks per pipeline step, non-temporal block stream)
  int spmv_chunk = 8, spmv_nt = 1;

  int spmv_grid() const { return span_grid; }

  // q = (H + lambda I) v; partials of v.q in d_part_a and, with rvec, of rvec.v in d_part_b
  // With a start/stop event pair the dispatch itself is timestamped (hipExtLaunchKernelGGL):
  // no extra barrier packets, so the figure agrees with rocprofv3's kernel trace.
  void spmv_raw(double lambda, const double* v, double* q, const double* rvec, DevScalars* scp,
                hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

  int spmv_launch(double lambda, const double* z, const double* rv, std::string& err);

  // Preconditioned CG on (H + lambda I) x = b in the single-reduction form (k_pcg_step); the
  // result stays in d_x.  Two launches and one reduction point per iteration; the host only polls
  // a 100-byte struct every `pcg_check_every` iterations.
  int agree_on_fail(std::string& err);

  int pcg(double lambda, int32_t* iters, double* rel_res, bool* ok, std::string& err);

  // prec: 0 block-Jacobi, 1 chain segments, 2 aggregation multigrid
  // probe_budget > 0 (block-Jacobi tried first on a damping-dominated system): after 8 iterations the
  // reduction reached so far predicts the total; if that exceeds the budget -- or the budget runs out --
  // *abandoned is set and the caller solves again with the hierarchy
  int pcg_attempt(double lambda, int prec, int32_t* iters, double* rel_res, bool* ok,
                  bool* chain_broke, std::string& err, int probe_budget = 0, bool* abandoned = nullptr);

  int optimize(int32_t max_iters, std::vector<sim3opt_iter_stats>& stats, std::string& err);
};

}  // namespace sim3opt
