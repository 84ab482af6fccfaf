/*
 * sim3_oracle.h -- CPU restatement of the reference's Sim(3) pose-graph LM path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sim3opt_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in g2o @
 * 8564e1e and vio_g2o @ HEAD (reference build.sh:41-45, :92), which are not
 * under /root/reference and cannot be fetched, and the reference ships no
 * tests, golden vectors or stored optimiser outputs for this path
 * (SURVEY.md section 8c).  This file restates the published g2o algorithm
 * (LM + numeric Jacobians + exact sparse Cholesky) and follows the in-tree
 * formula authority sim3_rv.h for exp/log.  What pins it is listed in
 * tests/test_oracle.py: the reference's input data files, the by-construction
 * zero residual of odometry edges (kitti_surf.cpp:653-666), the loop scale
 * ln(5.32393351) (loopConstraints.txt record 1), group identities, closed-form
 * Jacobians, dense numpy linear algebra, an independent numpy restatement of
 * the Sim(3) formulae (sim3opt_amd/sim3np.py, used only to prepare data), and --
 * for the LM trace itself -- an independent numpy / scipy Levenberg-Marquardt
 * (tests/golden/make_lm_golden.py: own central differences, scipy assembly,
 * SuperLU solve, own lambda policy) whose lock-step and free-run records on both
 * KITTI-00 graphs are committed as tests/golden/kitti_lm_golden.json.
 */
#ifndef SIM3_ORACLE_H
#define SIM3_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Sim(3) element: x -> s*R(q)*x + t.  q in Eigen coeffs() order (x,y,z,w),
 * as printed by the reference (kitti_surf.cpp:698).  8 doubles, 64 bytes. */
typedef struct {
  double q[4];
  double t[3];
  double s;
} or_sim3;

/* All constants of the restated g2o behaviour in one place. */
typedef struct {
  double tau;               /* 1e-5   lambda0 = tau * max diag(H)            */
  double user_lambda_init;  /* 0      >0 overrides tau rule                   */
  double good_step_lower;   /* 1/3                                            */
  double good_step_upper;   /* 2/3                                            */
  int    max_trials;        /* 10     maxTrialsAfterFailure                   */
  double fd_delta;          /* 1e-9   central-difference step                 */
  double exp_eps;           /* 1e-5   branch threshold of exp/log             */
  int    small_rot_half;    /* 0: R=I+W+W^2 (sim3_rv.h:151); 1: I+W+W^2/2     */
  int    fix_small_angle_b; /* 0: B as written (sim3_rv.h:166, :290); 1: exact
                               small-theta limit ((s2/2-s+1)e^s-1)/s^3          */
  int    dof_mask;          /* 127    bit d set = tangent component d is free; cleared bits
                               zero that Jacobian column (frozen rotation: 0x78)   */
  int    threads;           /* 1      OpenMP threads for per-edge loops       */
} or_options;

/* One record per LM iteration. */
typedef struct {
  double chi2_before;  /* chi2 at iteration start                      */
  double chi2_after;   /* chi2 kept at iteration end                   */
  double lambda;       /* lambda AFTER the iteration's policy update   */
  double rho;          /* last gain ratio                              */
  int    trials;       /* LM trials used (qmax)                        */
  int    solve_ok;     /* last linear solve succeeded (SPD)            */
  double t_linearize;  /* seconds                                      */
  double t_solve;
  double t_update;
} or_iter;

enum { OR_KERNEL_NONE = 0, OR_KERNEL_HUBER = 1 };

void or_options_default(or_options *o);

/* group operations -- sim3_rv.h:125-190 (exp), :242-320 (ln), :199-220 */
void or_sim3_exp(const double xi[7], const or_options *o, or_sim3 *out); /* xi=[omega,upsilon,sigma] */
void or_sim3_log(const or_sim3 *S, const or_options *o, double xi[7]);
void or_sim3_mul(const or_sim3 *a, const or_sim3 *b, or_sim3 *out);
void or_sim3_inv(const or_sim3 *a, or_sim3 *out);
void or_quat_from_R(const double R[9] /*row-major*/, double q[4]);
void or_R_from_quat(const double q[4], double R[9] /*row-major*/);
void or_euler_rpy_to_R(double r, double p, double y, double R[9]); /* kittiDetector.h:225-243 */

/* EdgeSim3::computeError: e = log(C * S0 * S1^-1) */
void or_edge_error(const or_sim3 *C, const or_sim3 *S0, const or_sim3 *S1,
                   const or_options *o, double e[7]);
/* BaseBinaryEdge numeric linearizeOplus; A,B column-major 7x7 */
void or_edge_jacobians(const or_sim3 *C, const or_sim3 *S0, const or_sim3 *S1,
                       const or_options *o, double A[49], double B[49]);

/* bulk per-edge evaluation (for kernel-level parity tests) */
void or_all_errors(int ne, const int *v0, const int *v1, const double *meas,
                   const double *states, const or_options *o, double *e_out /*ne x 7*/);
void or_all_jacobians(int ne, const int *v0, const int *v1, const double *meas,
                      const double *states, const or_options *o,
                      double *A_out /*ne x 49*/, double *B_out /*ne x 49*/);

/* chi2 = sum rho(e^T Omega e) */
double or_chi2(int nv, const double *states, int ne, const int *v0, const int *v1,
               const double *meas, const double *info, int kernel, double kdelta,
               const or_options *o);

/* Dense normal equations for small graphs: H (n x n, column-major, full
 * symmetric) and b (n), n = 7*(#free vertices); free vertices take block
 * indices in ascending id order.  Returns n. */
int or_build_dense(int nv, const double *states, const unsigned char *fixed, int ne,
                   const int *v0, const int *v1, const double *meas, const double *info,
                   int kernel, double kdelta, const or_options *o, double *H, double *b);

/* Solve (H + lambda I) x = b by the same sparse LDL^T used in or_optimize.
 * Returns 1 on success, 0 if not positive definite. x has 7*#free entries. */
int or_solve_once(int nv, const double *states, const unsigned char *fixed, int ne,
                  const int *v0, const int *v1, const double *meas, const double *info,
                  int kernel, double kdelta, const or_options *o, double lambda,
                  double *x, double *b_out);

/* SparseOptimizer::optimize(max_iters) with OptimizationAlgorithmLevenberg.
 * states (nv x 8) updated in place.  trace must hold max_iters records.
 * Returns the number of iterations executed (g2o convention), 0 on failure,
 * -1 if there is nothing to optimise. */
int or_optimize(int nv, double *states, const unsigned char *fixed, int ne,
                const int *v0, const int *v1, const double *meas, const double *info,
                int kernel, double kdelta, int max_iters, const or_options *o,
                or_iter *trace);

/* number of nonzeros in L of the last factorisation (fill diagnostic) */
long or_last_lnz(void);

#ifdef __cplusplus
}
#endif
#endif
