/*
 * sim3opt.h -- C-ABI of libsim3opt: MI355X-native Sim(3) pose-graph LM optimiser.
 *
 * Drop-in boundary for the path  optimizer.initializeOptimization();
 * optimizer.optimize(100);  of the reference (kitti_surf.cpp:674-675, :1044-1045)
 * on graphs of vio::VertexSim3Expmap / vio::EdgeSim3.  Each entry point names the
 * g2o call of the reference it replaces.  Plain pointers and sizes only; the
 * library copies everything it is given (the caller keeps no pointers into it).
 *
 * Conventions crossing the boundary:
 *   Sim3 state      8 doubles [qx qy qz qw tx ty tz s]   (Eigen coeffs() order, kitti_surf.cpp:698)
 *   tangent order   [omega(3), upsilon(3), sigma]        (g2o)
 *   information     7x7 double, column-major, NULL = identity (kitti_surf.cpp:592, :637, :667)
 *   vertex ids      arbitrary int32 (g2o allows any), mapped internally
 *   errors          int status codes, no exceptions, no abort(); text via sim3opt_last_error
 *
 * All computation runs on the GPU (HIP, gfx950).  There is no CPU fallback: if no
 * HIP device is usable, sim3opt_initialize returns SIM3OPT_ERR_NO_DEVICE.
 */
#ifndef SIM3OPT_H
#define SIM3OPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sim3opt_graph sim3opt_graph;

enum {
  SIM3OPT_OK = 0,
  SIM3OPT_ERR_ARG = -1,        /* bad argument (null, unknown id, duplicate id, ...) */
  SIM3OPT_ERR_STATE = -2,      /* call order (e.g. optimize before initialize)       */
  SIM3OPT_ERR_NO_DEVICE = -3,  /* no usable HIP device                               */
  SIM3OPT_ERR_HIP = -4,        /* a HIP runtime call failed                          */
  SIM3OPT_ERR_IO = -5,         /* file could not be read / parsed                    */
  SIM3OPT_ERR_COMM = -6        /* RCCL communicator failure                          */
};

enum { SIM3OPT_KERNEL_NONE = 0, SIM3OPT_KERNEL_HUBER = 1 };

/* Solver configuration.  Replaces the reference's
 *   OptimizationAlgorithmLevenberg(BlockSolverX(LinearSolverEigen))   kitti_surf.cpp:552-558
 * and g2o's setUserLambdaInit / setMaxTrialsAfterFailure            kittiDetector.h:730, 779-782.
 * Defaults (sim3opt_options_default) are g2o's. */
typedef struct sim3opt_options {
  double tau;               /* 1e-5  lambda0 = tau * max|H_dd|                           */
  double user_lambda_init;  /* 0     > 0 overrides the tau rule                          */
  double good_step_lower;   /* 1/3                                                       */
  double good_step_upper;   /* 2/3                                                       */
  int32_t max_trials;       /* 10    maxTrialsAfterFailure                               */
  double fd_delta;          /* 1e-9  central-difference step of the numeric Jacobians (g2o's
                                        BaseBinaryEdge default; 1e-6 gives 1e-10-accurate Jacobians) */
  double exp_eps;           /* 1e-5  branch threshold of exp/log (sim3_rv.h:133)         */
  int32_t small_rot_half;   /* 0     R = I+W+W^2 (sim3_rv.h:151); 1: I+W+W^2/2           */
  int32_t fix_small_angle_b;/* 0     B coefficient as written in sim3_rv.h:166/:290 (reference
                                        behaviour); 1: exact small-theta limit               */
  int32_t dof_mask;         /* 127   bit d set = tangent component d ([w0 w1 w2 u0 u1 u2 s]) is
                                        optimised; cleared bits freeze it (0x78 = rotations frozen:
                                        the scale+translation stage, kitti_surf.cpp:1020-1024)    */
  int32_t pcg_max_iters;    /* 0 = automatic: 2n for n = 7*free vertices <= 50000, else 1000 */
  double pcg_rel_tol;       /* 1e-10 stop when ||r||_Minv <= tol * ||b||_Minv            */
  int32_t pcg_check_every;  /* 16    PCG iterations between host convergence polls (at most 4 with the
                                        multigrid preconditioner: its iterations are ~1 ms each)  */
  int32_t pcg_graph;        /* 1     replay the PCG iterations from a captured hipGraph (single GPU,
                                        time_kernels = 0); 0 = enqueue every launch               */
  int32_t preconditioner;   /* -1    0 = 7x7 block-Jacobi, 1 = block-tridiagonal chain segments,
                                       2 = aggregation multigrid (pairwise-matched aggregates,
                                       Ad(S_v)-transported prolongation, dense coarsest level),
                                       -1 = automatic: multigrid when the graph has more than 256
                                       free vertices and coarsens like a low-dimensional graph
                                       (level-1 blocks <= 0.3 x level-0 blocks: chains, chains with
                                       loops, Manhattan worlds -- not expanders), chain segments for
                                       smaller nearly pure chains (well-posed arithmetic only), else
                                       block-Jacobi                                                */
  int32_t chain_segment;    /* 256   rows per chain segment (2..256)                             */
  int32_t device;           /* -1    HIP device ordinal; -1 = current device             */
  int32_t verbose;          /* 0     1: one stderr line per LM iteration (setVerbose)    */
  int32_t time_kernels;     /* 0     1: bracket every SpMV / linearise launch with HIP events
                                        (sim3opt_get_kernel_times); small launch-gap cost    */
  int32_t linear_solver;    /* -1    how (H + lambda I) dx = b is solved (LinearSolverEigen's role,
                                        kitti_surf.cpp:553-554):
                                        1 = exact sparse block Cholesky on the GPU (nested-dissection
                                            order, level-scheduled; the reference's SimplicialLDLT),
                                        0 = preconditioned CG (see `preconditioner`),
                                       -1 = automatic: the exact factorisation when it is cheap (single
                                            GPU, at most ~3e5 7x7x7 block products per factorisation:
                                            KITTI-00 and other chain-like graphs) and no
                                            `preconditioner` was named, else the PCG                */
  /* ---- tuning of the PCG path that changes its NUMERICS (iteration counts, summation orders; the
   * solution it converges to is the same).  Read by sim3opt_initialize.  0 / negative = the default
   * rule where stated.  A SIM3OPT_* environment variable of the same name overrides the field at
   * sim3opt_initialize (debug aid); sim3opt_get_options then reports the value that was used. ---- */
  int32_t amg_cycle[4];     /* 0,..  visits of multigrid level 1, 2, 3, >= 4 per visit of the level above
                                        (1 = V, 2 = W, 3); all 0 = automatic: {2,3,3,3}, or {1,2,2,2} when level 0
                                        is partitioned over the ranks (the coarse cycle is what stays
                                        latency-bound when level 0 is sharded)  [SIM3OPT_AMG_CYCLE] */
  int32_t amg_passes[3];    /* 0,..  pairwise-matching passes on level 0, 1, >= 2 (aggregates of 2^passes rows);
                                        0 = automatic: 3, or 2 on level 0 of a graph that reaches the dense level
                                        that way                                              [SIM3OPT_AMG_PASSES] */
  int32_t amg_additive;     /* 0     1: additive level 0 (no level-0 matrix pass in the cycle) [SIM3OPT_AMG_ADDITIVE] */
  int32_t amg_fp32;         /* 1     the cycle's matrix passes stream FP32 copies of the blocks [SIM3OPT_AMG_FP32] */
  int32_t amg_pivot;        /* 14    pivot rows of the dense coarsest inverse (14 or 28)       [SIM3OPT_AMG_PIVOT] */
  int32_t amg_coarsest;     /* 256   most block rows of the dense coarsest level (8..256)      [SIM3OPT_AMG_COARSEST] */
  int32_t adaptive_prec;    /* 1     automatic preconditioner only: damping-dominated solves start with
                                        block-Jacobi (DESIGN.md 5a')                           [SIM3OPT_ADAPTIVE_PREC] */
  int32_t row_order;        /* -1    block-row order: 0 = insertion (g2o's hessianIndex), 1 = breadth-first
                                        locality order, -1 = automatic (insertion on one rank, locality order
                                        when partitioned)                                      [SIM3OPT_ROW_ORDER=insertion|bfs] */
  int32_t halo_exchange;    /* 1     partitioned runs exchange boundary rows only; 0 = whole-vector all-gather
                                                                                               [SIM3OPT_NO_HALO] */
  int32_t span_grid;        /* 0     workgroups of the span SpMV; 0 = automatic                [SIM3OPT_SPAN_GRID] */
  int32_t force_collectives;/* 0     1: run every collective of the partitioned path even with one rank
                                        (transport self-test)                                  [SIM3OPT_FORCE_COMM] */
  int32_t amg_shard_rows;   /* 4096  partitioned runs: multigrid levels with more block rows than this are
                                        partitioned by owner like level 0 (aggregates never straddle ranks), smaller
                                        ones are replicated                                  [SIM3OPT_AMG_SHARD_ROWS] */
  int32_t amg_virtual_ranks;/* 0     > 1 on ONE rank: build the hierarchy as an N-rank partition would (aggregates
                                        inside N equal row spans): what a partitioned run is compared with
                                                                                               [SIM3OPT_AMG_VIRTUAL_RANKS] */
  int32_t pcg_batch;        /* 0     most right-hand sides solved together when LM trials are rejected in a row
                                        (DESIGN.md 5d); 0 = automatic, 1 = one at a time          [SIM3OPT_PCG_BATCH] */
  double amg_omega;         /* 0.9   damping of the block-Jacobi smoother (0.1..0.95)          [SIM3OPT_AMG_OMEGA] */
  double amg_over[2];       /* 1.8, 1.6  over-correction of the coarse correction prolonged into level 0 / into
                                        deeper levels                                          [SIM3OPT_AMG_OVER=a0,a1] */
  int64_t direct_max_pairs; /* 0     most 7x7x7 block products of a factorisation the automatic rule accepts;
                                        0 = 300000 (3e7 with linear_solver = 1)               [SIM3OPT_DIRECT_MAX_PAIRS] */
  int32_t debug_full_arrays;/* 0     partitioned runs allocate the block arrays (H, its FP32 copy, the partitioned
                                        coarse levels, the assembly scratch) for the rank's own rows only; 1: whole
                                        arrays with everything outside the rank's range poisoned (0xFF = NaN) and
                                        checked after every linearisation / optimize -- a write there is an error,
                                        a read shows as NaN (the test of the ranges)          [SIM3OPT_DEBUG_FULL_ARRAYS] */
} sim3opt_options;

/* Per-iteration record (g2o G2OBatchStatistics role; bal_example.cpp:55-56). */
typedef struct sim3opt_iter_stats {
  double chi2_before;
  double chi2_after;
  double lambda;        /* after the iteration's policy update */
  double rho;           /* last gain ratio                     */
  int32_t trials;       /* LM trials used                      */
  int32_t pcg_iters;    /* PCG iterations summed over trials   */
  double pcg_rel_res;   /* last achieved relative residual     */
  double ms_linearize;  /* device time, HIP events; measured when options.time_kernels or .verbose is set
                         * or the system has more than 4096 block rows, 0 otherwise (the three event
                         * markers per trial are 7 % of a KITTI-00 iteration) */
  double ms_solve;
  double ms_update;     /* oplus + chi2 + scale                */
  int32_t pcg_capped;   /* PCG solves of this iteration that stopped at pcg_max_iters without reaching
                         * pcg_rel_tol.  LinearSolverEigen (kitti_surf.cpp:553-554) has no such state; a
                         * capped solve is an INEXACT LM step, not a failed one: CG iterates from x0 = 0 are
                         * descent directions of the damped model and satisfy x.(lambda x + b) = x.(H+lambda I)x,
                         * so g2o's gain ratio stays meaningful and decides the trial.  Exact
                         * solver: always 0 */
  int32_t reserved_;
} sim3opt_iter_stats;

/* Device time of the dominant kernels accumulated since initialize / reset
 * (HIP events on the library's own stream; used by bench.py's roofline). */
typedef struct sim3opt_kernel_times {
  double ms_spmv;      int64_t n_spmv;
  double ms_pcg_vec;   int64_t n_pcg_vec;
  double ms_linearize; int64_t n_linearize;
  double ms_chi2;      int64_t n_chi2;
  double ms_update;    int64_t n_update;
  /* multigrid cycles: device time spent on the levels a rank partition REPLICATES (every rank runs them whole:
   * the part of a PCG iteration that does not shrink with the number of ranks) and the number of visits of the
   * first such level; on one rank with options.amg_virtual_ranks = N: what an N-rank run would replicate */
  double ms_replicated_levels; int64_t n_replicated_visits;
  /* batched solves of rejected LM trials (options.pcg_batch): batches run, systems they held */
  int64_t n_batches; int64_t n_batched_solves;
} sim3opt_kernel_times;

/* Device time of the collectives of the row-partitioned path accumulated since initialize / reset
 * (HIP event pairs on the library's stream around every collective when options.time_kernels is set;
 * a pair includes the wait for the slowest rank).  bytes: payload of the whole vector / buffer. */
typedef struct sim3opt_comm_times {
  double ms_allreduce;  int64_t n_allreduce;  int64_t bytes_allreduce;
  double ms_allgather;  int64_t n_allgather;  int64_t bytes_allgather;
  /* neighbour exchanges (grouped send / receive pairs); bytes: what THIS rank sent plus what it received */
  double ms_exchange;   int64_t n_exchange;   int64_t bytes_exchange;
} sim3opt_comm_times;

int sim3opt_version(void);
void sim3opt_options_default(sim3opt_options* o);

/* g2o::SparseOptimizer ctor + setAlgorithm                         kitti_surf.cpp:552-558 */
sim3opt_graph* sim3opt_create(void);
void sim3opt_destroy(sim3opt_graph* g);
/* May be called at any time.  device, linear_solver, preconditioner and chain_segment are read by
 * sim3opt_initialize; every other field takes effect at the next call that uses it. */
int sim3opt_set_options(sim3opt_graph* g, const sim3opt_options* o);
int sim3opt_get_options(const sim3opt_graph* g, sim3opt_options* o);
const char* sim3opt_last_error(const sim3opt_graph* g);

/* new VertexSim3Expmap; setEstimate; setFixed; setId; addVertex    kitti_surf.cpp:602-620 */
int sim3opt_add_vertex(sim3opt_graph* g, int32_t id, const double state[8], int32_t fixed);
/* bulk form for graphs too large for per-element calls */
int sim3opt_add_vertices(sim3opt_graph* g, int32_t n, const int32_t* ids /*NULL: 0..n-1 appended*/,
                         const double* states /*n x 8*/, const uint8_t* fixed /*NULL: none*/);

/* new EdgeSim3; setVertex(0,v0); setVertex(1,v1); setMeasurement; information(); [setRobustKernel];
 * addEdge                                                            kitti_surf.cpp:633-638, :663-668 */
int sim3opt_add_edge(sim3opt_graph* g, int32_t id_v0, int32_t id_v1, const double meas[8],
                     const double* info77 /*NULL = I7*/, int32_t kernel, double kernel_delta);
int sim3opt_add_edges(sim3opt_graph* g, int32_t m, const int32_t* id_v0, const int32_t* id_v1,
                      const double* meas /*m x 8*/, const double* info /*NULL or m x 49*/,
                      int32_t kernel, double kernel_delta);

int32_t sim3opt_num_vertices(const sim3opt_graph* g);
int32_t sim3opt_num_edges(const sim3opt_graph* g);
/* edges()[k]: endpoint ids and measurement of the k-th edge added   (g2o OptimizableGraph::edges) */
int sim3opt_get_edge(const sim3opt_graph* g, int32_t k, int32_t* id_v0, int32_t* id_v1,
                     double meas[8]);

/* SparseOptimizer::initializeOptimization()                        kitti_surf.cpp:674
 * Builds the index mapping and the block-CSR pattern, uploads the graph to HBM. */
int sim3opt_initialize(sim3opt_graph* g);

/* SparseOptimizer::optimize(n)                                     kitti_surf.cpp:675
 * Returns iterations executed (>0), -1 if nothing to optimise, 0 on failure
 * (g2o convention; details via sim3opt_last_error). */
int sim3opt_optimize(sim3opt_graph* g, int32_t max_iters);

/* vertex(id)->estimate()                                           kitti_surf.cpp:688-689 */
int sim3opt_get_vertex(sim3opt_graph* g, int32_t id, double state[8]);
/* vertex(id)->setEstimate() after construction (warm start)        kitti_surf.cpp:1037-1038 */
int sim3opt_set_vertex(sim3opt_graph* g, int32_t id, const double state[8]);
/* all estimates in insertion order */
int sim3opt_get_vertices(sim3opt_graph* g, double* states /*n x 8*/);
int sim3opt_set_vertices(sim3opt_graph* g, const double* states /*n x 8*/);

/* computeActiveErrors(); activeRobustChi2()                        kittiDetector.h:786-787 */
int sim3opt_chi2(sim3opt_graph* g, double* chi2);

/* statistics of the last optimize() */
int32_t sim3opt_num_iterations(const sim3opt_graph* g);
int sim3opt_get_stats(const sim3opt_graph* g, int32_t iter, sim3opt_iter_stats* out);
int sim3opt_get_kernel_times(sim3opt_graph* g, sim3opt_kernel_times* out);
int sim3opt_reset_kernel_times(sim3opt_graph* g);   /* (also clears the collectives' times) */
int sim3opt_get_comm_times(sim3opt_graph* g, sim3opt_comm_times* out);

/* ---- kernel-level access (parity tests against the CPU oracle, bench roofline) ---- */
/* per-edge residuals e (m x 7), edge insertion order                EdgeSim3::computeError */
int sim3opt_edge_errors(sim3opt_graph* g, double* e_out);
/* runs the linearisation kernels once on the current estimates     BlockSolver::buildSystem */
int sim3opt_linearize(sim3opt_graph* g);
/* dimensions of the block-CSR system: free block rows, stored 7x7 blocks */
int sim3opt_system_dims(const sim3opt_graph* g, int32_t* n_block_rows, int64_t* n_blocks);
/* the block-CSR pattern alone (host only, no GPU needed, may be called before initialize): two
 * calls, arrays NULL to size them.  Row k = k-th free vertex in insertion order: its diagonal block,
 * then one block per incident edge whose other endpoint is free, sorted by (column, edge). */
int sim3opt_system_pattern(sim3opt_graph* g, int32_t* n_block_rows, int64_t* n_blocks,
                           int32_t* rowptr, int32_t* colidx);
/* copies the block-CSR Hessian (rowptr nb+1, colidx nnzb, values nnzb x 49 column-major per
 * block) and b (7 nb) to the host; block row k = k-th free vertex in insertion order.  A partitioned run
 * assembles and holds the blocks (and the entries of b) of this rank's rows only (sim3opt_local_rows): the
 * other rows' blocks read zero */
int sim3opt_get_system(sim3opt_graph* g, int32_t* rowptr, int32_t* colidx, double* values,
                       double* b);
/* (a graph that is row-partitioned over several ranks numbers its block rows in locality order
 * instead: sim3opt_partition_plan(..., locality = 1, vertex_of_row, ...) gives the mapping) */
/* solves (H + lambda I) x = b with the block-Jacobi PCG on the last linearisation */
int sim3opt_solve(sim3opt_graph* g, double lambda, double* x /*7 nb*/, int32_t* iters,
                  double* rel_res);
/* Preconditioner the PCG of this (initialized) graph uses: 0 block-Jacobi, 1 chain segments,
 * 2 aggregation multigrid (what `preconditioner = -1` resolved to); negative = error code. */
int sim3opt_preconditioner_in_use(const sim3opt_graph* g);
/* Linear solver of this (initialized) graph: 1 exact sparse block Cholesky, 0 PCG (what
 * `linear_solver = -1` resolved to); negative = error code. */
int sim3opt_linear_solver_in_use(const sim3opt_graph* g);
/* What the automatic multigrid choices resolved to on this (initialized) graph: levels of the hierarchy
 * (0: none), how many of them are partitioned over the ranks (0 on one rank), visits of levels 1, 2, 3, >= 4 per
 * visit of the level above (options.amg_cycle = 0 picks {2,3,3,3}, or {1,2,2,2} on a partitioned run).  Any
 * pointer may be NULL. */
int sim3opt_amg_in_use(const sim3opt_graph* g, int32_t* n_levels, int32_t* n_partitioned, int32_t visits[4]);
/* Device memory of the block arrays of this (initialized) graph on this rank -- H, its FP32 copy, the coarse levels'
 * blocks, the assembly scratch: bytes[0] as allocated (a partitioned run holds its own rows only), bytes[1] what one
 * rank holding the whole graph allocates for them.  (Vectors, edges and index arrays -- about a tenth of the total --
 * are replicated and not counted.) */
int sim3opt_device_bytes(const sim3opt_graph* g, int64_t bytes[2]);
/* Plan of the exact sparse block Cholesky (LinearSolverEigen's role, kitti_surf.cpp:553-554) for this
 * graph: host only, no GPU needed, may be called before initialize.  Block column j of L is block row
 * perm[j] of the system (nested-dissection order); its stored 7x7 blocks are colptr[j]..colptr[j+1]
 * (diagonal first, rows lrow[] ascending).  Block s of L starts from the sum of the system's blocks
 * src[srcptr[s]..srcptr[s+1]) (indices into the block-CSR values of sim3opt_get_system) and subtracts
 * L[pa[k]] L[pb[k]]^T for k in pairptr[s]..pairptr[s+1].  Schedule: group q runs levels
 * gptr[q]..gptr[q+1], level l is columns lcolp[l]..lcolp[l+1]; groups but the last are independent.
 * Work split of the kernel: level l runs in rounds rptr[l]..rptr[l+1]; round q is 18 ints at
 * cells[18 q]: wavefront w owns blocks cells[18 q + w]..cells[18 q + w + 1] (at most 8) and the
 * products cells[18 q + 9 + w]..cells[18 q + 9 + w + 1] (4 wavefronts in the bottom groups, 8 in
 * the last one).
 * dims = {columns, blocks of L, block products per factorisation, elimination-tree height, groups,
 * levels, entries of src, rounds}.  Two calls: arrays NULL to size them, then filled.  max_pairs <= 0:
 * the automatic limit.  SIM3OPT_ERR_STATE when a factorisation needs more block products than that. */
int sim3opt_direct_plan(sim3opt_graph* g, int64_t max_pairs, int64_t dims[8], int32_t* perm,
                        int32_t* colptr, int32_t* lrow, int32_t* srcptr, int32_t* src,
                        int32_t* pairptr, int32_t* pa, int32_t* pb, int32_t* gptr, int32_t* lcolp,
                        int32_t* rptr, int32_t* cells);
/* Structure of the multigrid hierarchy `preconditioner = 2` would use for this graph (host only, no
 * GPU needed, may be called before initialize): *n_levels levels; rows[l] / blocks[l] = block rows
 * and stored 7x7 blocks of level l (up to `capacity` levels are written); aggregate_of_row (may be
 * NULL) receives, for each level-0 block row (free vertex in insertion order), its level-1 row.
 * SIM3OPT_ERR_STATE when the graph does not coarsen (block-Jacobi is used then). */
int sim3opt_amg_hierarchy(sim3opt_graph* g, int32_t capacity, int32_t* n_levels, int32_t* rows,
                          int64_t* blocks, int32_t* aggregate_of_row);

/* ---- row-partitioned multi-GPU (one process per GPU, RCCL over xGMI) ----
 * Every rank adds the SAME full graph; rank r then owns a contiguous range of block rows (equal
 * length), linearises the edges incident to them, streams its rows in the SpMV and keeps
 * a replica of all vertex estimates.  Collectives per PCG iteration (single-reduction CG): ONE
 * in-place all-gather of the preconditioned residual and ONE 2-double all-reduce; with the multigrid
 * preconditioner a second all-gather and the all-reduce of the restricted level-1 residual
 * (DESIGN.md section 7 lists sizes for N = 2 / 4 / 8); per LM trial: one all-gather of the step, one
 * 2-double all-reduce.  All ranks return identical results.  Call between create and initialize.
 * unique_id is the 128-byte ncclUniqueId produced by sim3opt_comm_unique_id on rank 0 and broadcast
 * by the caller (torch.distributed / MPI). */
int sim3opt_comm_unique_id(uint8_t id_out[128]);
int sim3opt_comm_init(sim3opt_graph* g, int32_t rank, int32_t world, const uint8_t unique_id[128]);
/* Same partitioned path over user-supplied host collectives (MPI, gloo, ...): operands are staged
 * through pinned host memory.  op: 0 = sum, 1 = max.  offsets has world+1 entries in doubles; rank r
 * owns buf[offsets[r] .. offsets[r+1]) on entry and the whole buf must be filled on return.
 * Callbacks return 0 on success. */
typedef int (*sim3opt_allreduce_fn)(void* ctx, double* buf, int32_t n, int32_t op);
typedef int (*sim3opt_allgatherv_fn)(void* ctx, double* buf, const int64_t* offsets, int32_t rank,
                                     int32_t world);
int sim3opt_comm_init_callbacks(sim3opt_graph* g, int32_t rank, int32_t world,
                                sim3opt_allreduce_fn allreduce, sim3opt_allgatherv_fn allgatherv,
                                void* ctx);
/* Optional third callback: the neighbour exchange of the partitioned path (halo rows of a level go to the
 * ranks that read them and to nobody else).  send[send_offsets[p] .. send_offsets[p+1]) goes to rank p,
 * recv[recv_offsets[p] .. recv_offsets[p+1]) must hold what rank p sent to this rank on return (offsets
 * in doubles, world+1 entries each; most spans are empty: a slab has two neighbours).  Without it the
 * library falls back to the all-gather of the whole vector.  Call after sim3opt_comm_init_callbacks. */
typedef int (*sim3opt_alltoallv_fn)(void* ctx, const double* send, const int64_t* send_offsets, double* recv,
                                    const int64_t* recv_offsets, int32_t rank, int32_t world);
int sim3opt_comm_set_alltoallv(sim3opt_graph* g, sim3opt_alltoallv_fn alltoallv);
/* Plan of the per-iteration exchange for `n_block_rows` rows over `world` ranks (host only): fills
 * row_begin (world+1, may be NULL) with the equal-length rank partition and returns 1 when the
 * in-place equal-count ncclAllGather applies (always, for this partition -- trailing ranks may be
 * short or empty), 0 otherwise, negative on bad arguments.  *count = doubles each rank contributes,
 * *padded_len = doubles every exchanged vector is allocated with (world * count >= 7 n_block_rows;
 * the tail is padding no kernel reads). */
int sim3opt_comm_allgather_plan(int32_t n_block_rows, int32_t world, int32_t* row_begin,
                                int64_t* count, int64_t* padded_len);
/* host-side partition plans (no GPU needed), world+1 entries each:
 *   _equal : the RANK partition -- equal-length row spans, so the per-iteration exchange is one
 *            in-place ncclAllGather (pose-graph rows have near-uniform block counts)
 *   plain  : spans balanced by stored 7x7 blocks (used for the SpMV's per-wavefront spans) */
int sim3opt_partition_rows_equal(int32_t n_block_rows, int32_t world, int32_t* row_begin);
int sim3opt_partition_rows(int32_t n_block_rows, const int32_t* rowptr, int32_t world,
                           int32_t* row_begin /*world+1*/);
/* Row order and halo of the partition over `world` ranks (host only, may be called before
 * initialize).  locality = 1: the block rows in the breadth-first locality order the partitioned path
 * gives a graph with world > 1 (contiguous rank spans are slabs of the graph); 0: insertion order (what
 * one rank uses: g2o's hessianIndex).  vertex_of_row (n_block_rows entries, may be NULL): vertex index
 * (insertion order) of every block row; row_begin: world + 1; boundary_rows_of_rank[r]: rows of rank r
 * with a neighbour on another rank -- what it sends per exchange of the partitioned PCG;
 * *cut_edges: edges whose endpoints two ranks own (both linearise them). */
int sim3opt_partition_plan(sim3opt_graph* g, int32_t world, int32_t locality, int32_t* vertex_of_row,
                           int32_t* row_begin, int32_t* boundary_rows_of_rank, int64_t* cut_edges);
/* Neighbour-only exchange plan of rank `rank` on level 0 of the partition over `world` ranks (host only;
 * locality order): rows it sends (its own rows another rank's rows reference, grouped by that rank:
 * send_seg has world + 1 entries) and rows it receives (grouped by owner).  Two calls: rows NULL to get
 * the counts.  By the symmetry of the pattern, rank p's receive group for q equals q's send group for p. */
int sim3opt_halo_plan(sim3opt_graph* g, int32_t world, int32_t rank, int32_t* n_send, int32_t* n_recv,
                      int32_t* send_rows, int32_t* send_seg, int32_t* recv_rows, int32_t* recv_seg);
/* block-row range [begin, end) this graph's rank owns (valid after initialize) */
int sim3opt_local_rows(const sim3opt_graph* g, int32_t* begin, int32_t* end);

/* Returns the device blocks the library keeps for re-use (graphs that are re-initialised after growing
 * by an edge find their predecessor's buffers, csrc/devmem.cpp; at most 1 GB) to the HIP runtime, together
 * with the idle streams, events and pinned host blocks (<= 64 MB) it recycles the same way.
 * Called automatically when the last sim3opt_graph / sim3opt_ba handle of the process is destroyed;
 * call it yourself before allocating large device buffers of your own next to a live handle. */
void sim3opt_release_device_cache(void);

/* ---- reference-format I/O (host C++; the callers either side of the path) ---- */
/* Builds the graph of testDirectSim3Optimization                   kitti_surf.cpp:562-670
 * from <dir>/cc.txt, <dir>/framePoses.txt (or framePoses_kf.txt), <dir>/loopConstraints.txt. */
int sim3opt_load_kitti_direct(sim3opt_graph* g, const char* dir, int32_t use_one_constraint);
/* Consistency graph against numbers the reference wrote itself: vertices = the KITTI ground-truth
 * poses (<dir>/gt_kf.txt, or 00.txt; 3x4 Pc2w rows, kitti_surf.cpp:1164-1190) as S_iw = (Rw2c, tw2c, 1);
 * edges = line 1 of every <dir>/loopConstraints.txt record, i.e. DCM2Euler(Pw2c[f2] Pw2c[f1]^-1) + its
 * translation as the reference's detector printed them from the same ground truth
 * (kittiDetector.h:1051-1060), v0 = frame 1, v1 = frame 2, scale 1.  Every residual of this graph
 * vanishes to the 8 decimals of the file iff the Euler convention, compose, inverse and edge
 * orientation used by sim3opt_load_kitti_direct are the reference's.  Host only. */
int sim3opt_load_kitti_gt_loops(sim3opt_graph* g, const char* dir);
/* Writes "kfid s tx ty tz qx qy qz qw" rows (S_wi of each estimate)  kitti_surf.cpp:678-701;
 * precision: 17 significant digits (the reference prints 6). image_ids may be NULL. */
int sim3opt_write_poses(sim3opt_graph* g, const char* path, const int32_t* image_ids);

/* ---- interchange formats and map re-anchoring ----
 * KeyFrame .bin reader: LoadComboKeyFrame                            drawPTAMPoints.cpp:33-84
 * Two-call pattern: with capacity < *n_obs only the header (id, Rw2c row-major, twinc, n_obs) is
 * returned; point_ids / points_w (n x 3) / obs_uv (n x 2) are filled when capacity >= n_obs. */
int sim3opt_read_keyframe_bin(const char* path, int32_t* kf_id, double Rw2c[9], double twinc[3],
                              int32_t* n_obs, uint32_t* point_ids, double* points_w, double* obs_uv,
                              int32_t capacity);
/* figureKITTIBA's re-anchoring                                        drawPTAMPoints.cpp:416-429
 * points[k] <- S_new(f)^-1 * (R_old(f) points[k] + t_old(f)), f = keyframe of the point's LAST
 * observation in (obs_frame, obs_point) order; unobserved points keep their coordinates.
 * old_Rt: n_frames x 12 (R row-major, then t); new_states: n_frames x 8 (S_iw).  Runs on the GPU. */
int sim3opt_reanchor_points(int32_t n_frames, const double* old_Rt, const double* new_states,
                            int32_t n_points, double* points, int32_t n_obs,
                            const int32_t* obs_frame, const int32_t* obs_point, int32_t device);
/* BAL problem file (ceres-solver format) handed from figureKITTIBA to ba_demo:
 * SaveBALFile                                                          drawPTAMPoints.cpp:218-283
 * Rw2c: n_cams x 9 row-major, tw2c: n_cams x 3, points: n_points x 3, observations as (camera,
 * point, u, v).  Point ids must be exactly 0..n_points-1 (SIM3OPT_ERR_ARG otherwise; the reference
 * exits).  Host only. */
int sim3opt_write_bal(const char* path, int32_t n_cams, const double* Rw2c, const double* tw2c,
                      const double f_k1_k2[3], int32_t n_points, const double* points,
                      int32_t n_obs, const int32_t* obs_cam, const int32_t* obs_point,
                      const double* obs_uv);
/* g2o text export (VERTEX_SIM3:EXPMAP / EDGE_SIM3:EXPMAP / FIX) of a graph with ids 0..n-1 and
 * identity information, for re-running it in stock g2o */
int sim3opt_write_g2o(sim3opt_graph* g, const char* path);

/* ---- bundle adjustment hand-off: the reference's ba_demo on the GPU ----
 * bal_example.cpp:44-243: g2o::VertexSE3Expmap cameras (T_w2c), g2o::VertexSBAPointXYZ points
 * (marginalised), g2o::EdgeProjectXYZ2UV with ONE fixed g2o::CameraParameters(f, pp, 0) (:90-97),
 * information I / pixel_noise^2 (:147), g2o::RobustKernelHuber(2.5) (:149-153), Levenberg-Marquardt
 * over BlockSolver_6_3 + LinearSolverEigen (:76-88), optimize(maxIterations) (:213).  No vertex is
 * fixed (the reference fixes none); the damping carries the gauge, as it does there.
 * Device pipeline: sim3opt_amd/csrc/ba.hip.  No CPU fallback: SIM3OPT_ERR_NO_DEVICE without a GPU. */
typedef struct sim3opt_ba sim3opt_ba;

typedef struct sim3opt_ba_options {
  double huber_delta;       /* RobustKernelHuber delta; 0 = no robust kernel      default 2.5  :151 */
  double pixel_noise;       /* information = I / pixel_noise^2                     default 1.0  :62  */
  double tau;               /* lambda_0 = tau * max diag(H)                        default 1e-5      */
  double user_lambda_init;  /* > 0: used instead                                   default 0         */
  int32_t max_trials;       /* LM trials per iteration                             default 10        */
  int32_t pcg_max_iters;    /* reduced camera system; 0 = automatic                default 0         */
  double pcg_rel_tol;       /* |r|_M / |r0|_M of the reduced system                default 1e-12     */
  int32_t linear_solver;    /* reduced camera system: -1 automatic (exact block Cholesky unless one
                               factorisation needs > 8 M block products), 1 exact or fail, 0 block-Jacobi
                               PCG                                                 default -1   :76-80 */
  int32_t device;           /* HIP device ordinal, -1 = current                    default -1        */
  int32_t verbose;          /* one line per LM iteration on stderr                 default 0    :72  */
} sim3opt_ba_options;

void sim3opt_ba_options_default(sim3opt_ba_options* o);
sim3opt_ba* sim3opt_ba_create(void);
void sim3opt_ba_destroy(sim3opt_ba* b);
const char* sim3opt_ba_last_error(const sim3opt_ba* b);
int sim3opt_ba_set_options(sim3opt_ba* b, const sim3opt_ba_options* o);
/* cameras: n_cams x 7 [qx qy qz qw tx ty tz] of T_w2c (the SE3Quat of :170-172); points n x 3;
 * observations (camera index, point index, u, v) as the BAL rows (:134-158).  Indices out of range
 * -> SIM3OPT_ERR_ARG (the reference asserts).  focal / cx / cy: the fixed CameraParameters. */
int sim3opt_ba_set_problem(sim3opt_ba* b, int32_t n_cams, const double* cam_qt, int32_t n_points,
                           const double* points, int32_t n_obs, const int32_t* obs_cam,
                           const int32_t* obs_point, const double* obs_uv, double focal, double cx,
                           double cy);
/* OptimizableGraph::Vertex::setFixed on cameras (n_cams flags; ba_demo itself fixes none).  A fixed
 * camera keeps its estimate and leaves the linear system. */
int sim3opt_ba_set_fixed_cameras(sim3opt_ba* b, const uint8_t* fixed);
/* the BAL file ba_demo takes as argv[1] (:104-189; per-camera f, k1, k2 are read and ignored as
 * there: the projection uses the fixed focal / cx / cy) */
int sim3opt_ba_read_bal(sim3opt_ba* b, const char* path, double focal, double cx, double cy);
int sim3opt_ba_dims(const sim3opt_ba* b, int32_t* n_cams, int32_t* n_points, int32_t* n_obs);
/* robustified chi2 of the current estimates (SparseOptimizer::activeRobustChi2) */
int sim3opt_ba_chi2(sim3opt_ba* b, double* chi2);
/* LM iterations performed (g2o's return convention: 0 on failure, -1 for an empty problem) */
int sim3opt_ba_optimize(sim3opt_ba* b, int32_t max_iters);
int sim3opt_ba_get_cameras(const sim3opt_ba* b, double* cam_qt /* n_cams x 7 */);
int sim3opt_ba_get_points(const sim3opt_ba* b, double* points /* n_points x 3 */);
int32_t sim3opt_ba_num_iterations(const sim3opt_ba* b);
int sim3opt_ba_get_stats(const sim3opt_ba* b, int32_t iter, sim3opt_iter_stats* out);
/* "% SE3 optimization result: kf id, tcinw, rc2w(qxyzw)" rows                        :223-238 */
int sim3opt_ba_write_poses(const sim3opt_ba* b, const char* path);

/* ---- stepwise optimisation, stage 1 (host C++) ----
 * "scale_dlt" of testStepwiseSim3Optimization                        kitti_surf.cpp:887-933
 * Null vector of the edge equations s_C x[v0] - x[v1] = 0 (the reference: last column of V of
 * Eigen::JacobiSVD), divided by its first entry, written into the scale of every vertex estimate.
 * Needs dense vertex ids 0..n-1 and n <= 4096.  sigma_ratio (optional) receives an estimate of
 * sigma_min / sigma_max (the reference warns below 5e-4).  Stage 2 / 3 are optimize() runs with
 * dof_mask = 0x78 (rotations frozen) / 127, each warm-started from the previous stage. */
int sim3opt_stepwise_scale_init(sim3opt_graph* g, double* sigma_ratio);

/* ---- evaluation harness (host C++) ----
 * estimateSimilarityTransform = Eigen::umeyama(query, train, true)   kitti_surf.cpp:1091-1161
 * followed by the RMSE / max deviation of the aligned positions      kitti_surf.cpp:1446-1463.
 * S is the 4x4 row-major similarity [cR t; 0 1] mapping query to train coordinates. */
int sim3opt_align_trajectory(int32_t n, const double* query_xyz /*n x 3*/,
                             const double* train_xyz /*n x 3*/, int32_t with_scale, double S[16],
                             double* rmse, double* max_dev);

#ifdef __cplusplus
}
#endif
#endif /* SIM3OPT_H */
