/*
 * sim3_oracle.c -- CPU restatement of the reference's Sim(3) pose-graph LM path.
 * TEST INFRASTRUCTURE ONLY (see sim3_oracle.h header comment; parity unpinned).
 *
 * What it restates, with the reference location each part follows:
 *   exp / log / inverse / compose ..... sim3_rv.h:125-190, :242-320, :199-220
 *                                       (tangent re-ordered to g2o's [omega, upsilon, sigma])
 *   edge residual ...................... g2o::EdgeSim3::computeError, set up at
 *                                       kitti_surf.cpp:633-638, :663-668
 *   numeric Jacobians, quadratic form .. g2o BaseBinaryEdge
 *   LM policy .......................... g2o OptimizationAlgorithmLevenberg, instantiated at
 *                                       kitti_surf.cpp:552-558, run at :674-675
 *   exact sparse Cholesky .............. g2o LinearSolverEigen (SimplicialLDLT + fill-reducing
 *                                       ordering), kitti_surf.cpp:553-554
 */
#include "sim3_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void or_options_default(or_options *o) {
  o->tau = 1e-5;
  o->user_lambda_init = 0.0;
  o->good_step_lower = 1.0 / 3.0;
  o->good_step_upper = 2.0 / 3.0;
  o->max_trials = 10;
  o->fd_delta = 1e-9;
  o->exp_eps = 1e-5;
  o->small_rot_half = 0;
  o->fix_small_angle_b = 0;
  o->dof_mask = 127;
  o->threads = 1;
}

/* ------------------------------------------------------------------ */
/* 3x3 helpers (row-major)                                             */
/* ------------------------------------------------------------------ */
static void skew3(const double w[3], double W[9]) { /* sim3_rv.h:38-50 */
  W[0] = 0;     W[1] = -w[2]; W[2] = w[1];
  W[3] = w[2];  W[4] = 0;     W[5] = -w[0];
  W[6] = -w[1]; W[7] = w[0];  W[8] = 0;
}

static void mul33(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += A[3 * i + k] * B[3 * k + j];
      C[3 * i + j] = acc;
    }
}

/* partial-pivot LU solve of a 3x3 system (TooN::LU / Eigen lu().solve role, sim3_rv.h:305-307) */
static void solve33(const double Win[9], const double tin[3], double x[3]) {
  double M[3][4];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i][j] = Win[3 * i + j];
    M[i][3] = tin[i];
  }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 4; ++j) { double tmp = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = tmp; }
    for (int r = c + 1; r < 3; ++r) {
      double f = M[r][c] / M[c][c];
      for (int j = c; j < 4; ++j) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 2; i >= 0; --i) {
    double acc = M[i][3];
    for (int j = i + 1; j < 3; ++j) acc -= M[i][j] * x[j];
    x[i] = acc / M[i][i];
  }
}

/* Eigen-convention matrix -> quaternion (trace branch, else largest diagonal) */
void or_quat_from_R(const double R[9], double q[4]) {
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
}

void or_R_from_quat(const double q[4], double R[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

static void quat_rot(const double q[4], const double v[3], double o[3]) {
  /* v + w*uv + qv x uv,  uv = 2 * (qv x v) */
  double ux = 2 * (q[1] * v[2] - q[2] * v[1]);
  double uy = 2 * (q[2] * v[0] - q[0] * v[2]);
  double uz = 2 * (q[0] * v[1] - q[1] * v[0]);
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

/* kittiDetector.h:225-243  R = Rz(yaw) Ry(pitch) Rx(roll) */
void or_euler_rpy_to_R(double r, double p, double y, double R[9]) {
  double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), ch = cos(y), sh = sin(y);
  R[0] = cp * ch; R[1] = sp * sr * ch - cr * sh; R[2] = cr * sp * ch + sh * sr;
  R[3] = cp * sh; R[4] = sr * sp * sh + cr * ch; R[5] = cr * sp * sh - sr * ch;
  R[6] = -sp;     R[7] = sr * cp;                R[8] = cr * cp;
}

/* ------------------------------------------------------------------ */
/* Sim(3) group                                                        */
/* ------------------------------------------------------------------ */
/* A, B, C of W = A*Omega + B*Omega^2 + C*I   (sim3_rv.h:143-181 / :261-303) */
static void w_coeffs(double sigma, double s, double theta, int small_theta, double eps, int fixb,
                     double *A, double *B, double *C) {
  if (fabs(sigma) < eps) {
    *C = 1.0;
    if (small_theta) {
      *A = 1.0 / 2.0;
      *B = 1.0 / 6.0;
    } else {
      double th2 = theta * theta;
      *A = (1 - cos(theta)) / th2;
      *B = (theta - sin(theta)) / (th2 * theta);
    }
  } else {
    *C = (s - 1) / sigma;
    if (small_theta) {
      double sg2 = sigma * sigma;
      *A = ((sigma - 1) * s + 1) / sg2;
      /* as written in sim3_rv.h:166 / :290 (not the small-theta limit); fixb selects the limit */
      *B = ((0.5 * sg2 - sigma + 1) * s - (fixb ? 1.0 : 0.0)) / (sg2 * sigma);
    } else {
      double a = s * sin(theta), b = s * cos(theta);
      double th2 = theta * theta, c = th2 + sigma * sigma;
      *A = (a * sigma + (1 - b) * theta) / (theta * c);
      *B = (*C - ((b - 1) * sigma + a * theta) / c) * 1.0 / th2;
    }
  }
}

void or_sim3_exp(const double xi[7], const or_options *o, or_sim3 *out) {
  const double *omega = xi, *ups = xi + 3;
  double sigma = xi[6], eps = o->exp_eps;
  double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], W[9];
  skew3(omega, Om);
  mul33(Om, Om, Om2);
  double s = exp(sigma);
  int small = theta < eps;
  double A, B, C;
  w_coeffs(sigma, s, theta, small, eps, o->fix_small_angle_b, &A, &B, &C);
  if (small) {
    double h = o->small_rot_half ? 0.5 : 1.0;
    for (int i = 0; i < 9; ++i) R[i] = Om[i] + h * Om2[i];
  } else {
    double k1 = sin(theta) / theta, k2 = (1 - cos(theta)) / (theta * theta);
    for (int i = 0; i < 9; ++i) R[i] = k1 * Om[i] + k2 * Om2[i];
  }
  R[0] += 1; R[4] += 1; R[8] += 1;
  for (int i = 0; i < 9; ++i) W[i] = A * Om[i] + B * Om2[i];
  W[0] += C; W[4] += C; W[8] += C;
  for (int i = 0; i < 3; ++i) out->t[i] = W[3 * i] * ups[0] + W[3 * i + 1] * ups[1] + W[3 * i + 2] * ups[2];
  or_quat_from_R(R, out->q);
  out->s = s;
}

void or_sim3_log(const or_sim3 *S, const or_options *o, double xi[7]) {
  double eps = o->exp_eps;
  double s = S->s, sigma = log(s);
  double R[9];
  or_R_from_quat(S->q, R);
  double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]}; /* sim3_rv.h:51-54 */
  double omega[3], theta = 0;
  int small = d > 1 - eps;
  if (small) {
    for (int i = 0; i < 3; ++i) omega[i] = 0.5 * dR[i];
  } else {
    theta = acos(d);
    double k = theta / (2 * sqrt(1 - d * d));
    for (int i = 0; i < 3; ++i) omega[i] = k * dR[i];
  }
  double A, B, C;
  w_coeffs(sigma, s, theta, small, eps, o->fix_small_angle_b, &A, &B, &C);
  double Om[9], Om2[9], W[9];
  skew3(omega, Om);
  mul33(Om, Om, Om2);
  for (int i = 0; i < 9; ++i) W[i] = A * Om[i] + B * Om2[i];
  W[0] += C; W[4] += C; W[8] += C;
  double ups[3];
  solve33(W, S->t, ups);
  xi[0] = omega[0]; xi[1] = omega[1]; xi[2] = omega[2];
  xi[3] = ups[0];   xi[4] = ups[1];   xi[5] = ups[2];
  xi[6] = sigma;
}

void or_sim3_mul(const or_sim3 *a, const or_sim3 *b, or_sim3 *out) { /* sim3_rv.h:214-220 */
  or_sim3 r;
  double rt[3];
  quat_mul(a->q, b->q, r.q);
  quat_rot(a->q, b->t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = a->s * rt[i] + a->t[i];
  r.s = a->s * b->s;
  *out = r;
}

void or_sim3_inv(const or_sim3 *a, or_sim3 *out) { /* sim3_rv.h:199-203 */
  or_sim3 r;
  r.q[0] = -a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = a->q[3];
  double tmp[3] = {(-1.0 / a->s) * a->t[0], (-1.0 / a->s) * a->t[1], (-1.0 / a->s) * a->t[2]};
  quat_rot(r.q, tmp, r.t);
  r.s = 1.0 / a->s;
  *out = r;
}

/* ------------------------------------------------------------------ */
/* Edge                                                                */
/* ------------------------------------------------------------------ */
void or_edge_error(const or_sim3 *C, const or_sim3 *S0, const or_sim3 *S1, const or_options *o,
                   double e[7]) {
  or_sim3 S1i, CS0, E;
  or_sim3_inv(S1, &S1i);
  or_sim3_mul(C, S0, &CS0);
  or_sim3_mul(&CS0, &S1i, &E);
  or_sim3_log(&E, o, e);
}

static void numeric_block(const or_sim3 *C, const or_sim3 *S0, const or_sim3 *S1, int which,
                          const or_options *o, double J[49]) {
  double delta = o->fd_delta, scalar = 1.0 / (2 * delta);
  double add[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int d = 0; d < 7; ++d) {
    double ep[7], em[7];
    or_sim3 P, Sp;
    add[d] = delta;
    or_sim3_exp(add, o, &P);
    or_sim3_mul(&P, which == 0 ? S0 : S1, &Sp);
    or_edge_error(C, which == 0 ? &Sp : S0, which == 0 ? S1 : &Sp, o, ep);
    add[d] = -delta;
    or_sim3_exp(add, o, &P);
    or_sim3_mul(&P, which == 0 ? S0 : S1, &Sp);
    or_edge_error(C, which == 0 ? &Sp : S0, which == 0 ? S1 : &Sp, o, em);
    add[d] = 0.0;
    for (int r = 0; r < 7; ++r) J[7 * d + r] = ((o->dof_mask >> d) & 1) ? scalar * (ep[r] - em[r]) : 0.0;
  }
}

void or_edge_jacobians(const or_sim3 *C, const or_sim3 *S0, const or_sim3 *S1, const or_options *o,
                       double A[49], double B[49]) {
  numeric_block(C, S0, S1, 0, o, A);
  numeric_block(C, S0, S1, 1, o, B);
}

static void set_threads(const or_options *o) {
#ifdef _OPENMP
  omp_set_num_threads(o->threads > 0 ? o->threads : 1);
#else
  (void)o;
#endif
}

void or_all_errors(int ne, const int *v0, const int *v1, const double *meas, const double *states,
                   const or_options *o, double *e_out) {
  set_threads(o);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < ne; ++k)
    or_edge_error((const or_sim3 *)(meas + 8 * (size_t)k), (const or_sim3 *)(states + 8 * (size_t)v0[k]),
                  (const or_sim3 *)(states + 8 * (size_t)v1[k]), o, e_out + 7 * (size_t)k);
}

void or_all_jacobians(int ne, const int *v0, const int *v1, const double *meas,
                      const double *states, const or_options *o, double *A_out, double *B_out) {
  set_threads(o);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < ne; ++k)
    or_edge_jacobians((const or_sim3 *)(meas + 8 * (size_t)k),
                      (const or_sim3 *)(states + 8 * (size_t)v0[k]),
                      (const or_sim3 *)(states + 8 * (size_t)v1[k]), o, A_out + 49 * (size_t)k,
                      B_out + 49 * (size_t)k);
}

/* e^T Omega e ; Omega column-major 7x7 or NULL for identity */
static double quad_form(const double e[7], const double *Om) {
  double acc = 0;
  if (!Om) {
    for (int i = 0; i < 7; ++i) acc += e[i] * e[i];
    return acc;
  }
  for (int c = 0; c < 7; ++c) {
    double col = 0;
    for (int r = 0; r < 7; ++r) col += e[r] * Om[7 * c + r];
    acc += col * e[c];
  }
  return acc;
}

/* g2o RobustKernelHuber::robustify */
static void robustify(int kernel, double kdelta, double e2, double rho[3]) {
  if (kernel == OR_KERNEL_HUBER) {
    double dsqr = kdelta * kdelta;
    if (e2 <= dsqr) {
      rho[0] = e2; rho[1] = 1.0; rho[2] = 0.0;
    } else {
      double sq = sqrt(e2);
      rho[0] = 2 * sq * kdelta - dsqr;
      rho[1] = kdelta / sq;
      rho[2] = -0.5 * rho[1] / e2;
    }
  } else {
    rho[0] = e2; rho[1] = 1.0; rho[2] = 0.0;
  }
}

double or_chi2(int nv, const double *states, int ne, const int *v0, const int *v1,
               const double *meas, const double *info, int kernel, double kdelta,
               const or_options *o) {
  (void)nv;
  set_threads(o);
  double *per = (double *)malloc(sizeof(double) * (size_t)(ne > 0 ? ne : 1));
#pragma omp parallel for schedule(static)
  for (int k = 0; k < ne; ++k) {
    double e[7], rho[3];
    or_edge_error((const or_sim3 *)(meas + 8 * (size_t)k), (const or_sim3 *)(states + 8 * (size_t)v0[k]),
                  (const or_sim3 *)(states + 8 * (size_t)v1[k]), o, e);
    robustify(kernel, kdelta, quad_form(e, info ? info + 49 * (size_t)k : NULL), rho);
    per[k] = rho[0];
  }
  double sum = 0; /* serial, fixed order */
  for (int k = 0; k < ne; ++k) sum += per[k];
  free(per);
  return sum;
}

/* ------------------------------------------------------------------ */
/* Per-edge quadratic form (g2o BaseBinaryEdge::constructQuadraticForm) */
/* ------------------------------------------------------------------ */
typedef struct {
  double H00[49], H11[49], H01[49]; /* column-major */
  double b0[7], b1[7];
} edge_quad;

static void edge_quadratic(const double A[49], const double B[49], const double e[7],
                           const double *Om, int kernel, double kdelta, edge_quad *out) {
  double w = 1.0;
  if (kernel != OR_KERNEL_NONE) {
    double rho[3];
    robustify(kernel, kdelta, quad_form(e, Om), rho);
    w = rho[1];
  }
  /* OA = w*Omega*A, OB = w*Omega*B, Oe = -w*Omega*e */
  double OA[49], OB[49], Oe[7];
  for (int c = 0; c < 7; ++c)
    for (int r = 0; r < 7; ++r) {
      double sa = 0, sb = 0;
      if (Om) {
        for (int k = 0; k < 7; ++k) {
          sa += Om[7 * k + r] * A[7 * c + k];
          sb += Om[7 * k + r] * B[7 * c + k];
        }
      } else {
        sa = A[7 * c + r];
        sb = B[7 * c + r];
      }
      OA[7 * c + r] = w * sa;
      OB[7 * c + r] = w * sb;
    }
  for (int r = 0; r < 7; ++r) {
    double se = 0;
    if (Om) for (int k = 0; k < 7; ++k) se += Om[7 * k + r] * e[k];
    else se = e[r];
    Oe[r] = -w * se;
  }
  for (int c = 0; c < 7; ++c)
    for (int r = 0; r < 7; ++r) {
      double h00 = 0, h11 = 0, h01 = 0;
      for (int k = 0; k < 7; ++k) {
        h00 += A[7 * r + k] * OA[7 * c + k];
        h11 += B[7 * r + k] * OB[7 * c + k];
        h01 += A[7 * r + k] * OB[7 * c + k];
      }
      out->H00[7 * c + r] = h00;
      out->H11[7 * c + r] = h11;
      out->H01[7 * c + r] = h01;
    }
  for (int r = 0; r < 7; ++r) {
    double s0 = 0, s1 = 0;
    for (int k = 0; k < 7; ++k) {
      s0 += A[7 * r + k] * Oe[k];
      s1 += B[7 * r + k] * Oe[k];
    }
    out->b0[r] = s0;
    out->b1[r] = s1;
  }
}

/* ------------------------------------------------------------------ */
/* Block minimum-degree ordering (fill-reducing; AMD's role in Eigen)  */
/* ------------------------------------------------------------------ */
typedef struct { int *a; int n, cap; } ivec;

static void ivec_push(ivec *v, int x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? 2 * v->cap : 8;
    v->a = (int *)realloc(v->a, sizeof(int) * (size_t)v->cap);
  }
  v->a[v->n++] = x;
}

static int cmp_int(const void *x, const void *y) {
  int a = *(const int *)x, b = *(const int *)y;
  return (a > b) - (a < b);
}

typedef struct { int deg, v; } hent;
typedef struct { hent *h; int n, cap; } heap;

static int hless(hent a, hent b) { return a.deg < b.deg || (a.deg == b.deg && a.v < b.v); }
static void heap_push(heap *H, hent e) {
  if (H->n == H->cap) {
    H->cap = H->cap ? 2 * H->cap : 64;
    H->h = (hent *)realloc(H->h, sizeof(hent) * (size_t)H->cap);
  }
  int i = H->n++;
  H->h[i] = e;
  while (i > 0) {
    int p = (i - 1) / 2;
    if (!hless(H->h[i], H->h[p])) break;
    hent t = H->h[i]; H->h[i] = H->h[p]; H->h[p] = t;
    i = p;
  }
}
static hent heap_pop(heap *H) {
  hent top = H->h[0];
  H->h[0] = H->h[--H->n];
  int i = 0;
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < H->n && hless(H->h[l], H->h[m])) m = l;
    if (r < H->n && hless(H->h[r], H->h[m])) m = r;
    if (m == i) break;
    hent t = H->h[i]; H->h[i] = H->h[m]; H->h[m] = t;
    i = m;
  }
  return top;
}

/* order[k] = block eliminated k-th */
static void min_degree_order(int nb, int npairs, const int *pa, const int *pb, int *order) {
  ivec *adj = (ivec *)calloc((size_t)(nb > 0 ? nb : 1), sizeof(ivec));
  for (int k = 0; k < npairs; ++k)
    if (pa[k] != pb[k]) { ivec_push(&adj[pa[k]], pb[k]); ivec_push(&adj[pb[k]], pa[k]); }
  for (int v = 0; v < nb; ++v) { /* sort + unique */
    qsort(adj[v].a, (size_t)adj[v].n, sizeof(int), cmp_int);
    int m = 0;
    for (int i = 0; i < adj[v].n; ++i)
      if (m == 0 || adj[v].a[m - 1] != adj[v].a[i]) adj[v].a[m++] = adj[v].a[i];
    adj[v].n = m;
  }
  char *gone = (char *)calloc((size_t)(nb > 0 ? nb : 1), 1);
  heap H = {0, 0, 0};
  for (int v = 0; v < nb; ++v) { hent e = {adj[v].n, v}; heap_push(&H, e); }
  int done = 0;
  int *tmp = NULL, tmpcap = 0;
  while (done < nb) {
    hent e = heap_pop(&H);
    int v = e.v;
    if (gone[v] || e.deg != adj[v].n) continue; /* stale */
    gone[v] = 1;
    order[done++] = v;
    ivec Nv = adj[v];
    for (int iu = 0; iu < Nv.n; ++iu) {
      int u = Nv.a[iu];
      ivec *au = &adj[u];
      int need = au->n + Nv.n;
      if (need > tmpcap) { tmpcap = 2 * need; tmp = (int *)realloc(tmp, sizeof(int) * (size_t)tmpcap); }
      int i = 0, j = 0, m = 0;
      while (i < au->n || j < Nv.n) { /* sorted union minus {u, v} */
        int x;
        if (j >= Nv.n || (i < au->n && au->a[i] < Nv.a[j])) x = au->a[i++];
        else if (i >= au->n || Nv.a[j] < au->a[i]) x = Nv.a[j++];
        else { x = au->a[i]; ++i; ++j; }
        if (x != u && x != v) tmp[m++] = x;
      }
      if (m > au->cap) { au->cap = m; au->a = (int *)realloc(au->a, sizeof(int) * (size_t)m); }
      memcpy(au->a, tmp, sizeof(int) * (size_t)m);
      au->n = m;
      hent ne = {m, u};
      heap_push(&H, ne);
    }
    free(adj[v].a);
    adj[v].a = NULL; adj[v].n = adj[v].cap = 0;
  }
  for (int v = 0; v < nb; ++v) free(adj[v].a);
  free(adj); free(gone); free(H.h); free(tmp);
}

/* ------------------------------------------------------------------ */
/* Sparse system: block pattern -> permuted scalar CSC (upper) + LDL^T */
/* ------------------------------------------------------------------ */
typedef struct {
  int nb, n;          /* free blocks, scalars */
  int *hidx;          /* vertex -> free block index or -1 */
  int *binv;          /* free block -> permuted block position */
  /* permuted block CSC (upper incl. diagonal, rows ascending, diagonal last) */
  int *bcp, *bri;
  /* scalar CSC of the upper triangle */
  long *Ap; int *Ai; double *Ax;
  /* per edge: slot of the off-diagonal block in (bcp,bri) or -1; transposed flag */
  long *eslot; char *etrans;
  /* LDL^T */
  int *Parent, *Lnz; long *Lp; int *Li; double *Lx, *D, *Y; int *Pattern, *Flag;
  double *Awork; /* Ax + lambda on the diagonal */
  double *b;     /* rhs in ORIGINAL free order */
  long lnz;
} sys_t;

static long g_last_lnz = 0;
long or_last_lnz(void) { return g_last_lnz; }

typedef struct { int c, r; } bpair;
static int cmp_bpair(const void *x, const void *y) {
  const bpair *a = (const bpair *)x, *b = (const bpair *)y;
  if (a->c != b->c) return (a->c > b->c) - (a->c < b->c);
  return (a->r > b->r) - (a->r < b->r);
}

static long block_slot(const sys_t *S, int r, int c) { /* binary search in column c */
  int lo = S->bcp[c], hi = S->bcp[c + 1] - 1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    if (S->bri[mid] == r) return mid;
    if (S->bri[mid] < r) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/* scalar position of element (i,k) of block slot p (rows i, column k of block column c) */
static long scalar_pos(const sys_t *S, int c, long p, int i, int k) {
  int a = (int)(p - S->bcp[c]);
  return S->Ap[7 * c + k] + 7 * a + i;
}

static void sys_free(sys_t *S) {
  free(S->hidx); free(S->binv); free(S->bcp); free(S->bri); free(S->Ap); free(S->Ai);
  free(S->Ax); free(S->eslot); free(S->etrans); free(S->Parent); free(S->Lnz); free(S->Lp);
  free(S->Li); free(S->Lx); free(S->D); free(S->Y); free(S->Pattern); free(S->Flag);
  free(S->Awork); free(S->b);
  memset(S, 0, sizeof(*S));
}

static void sys_build_structure(sys_t *S, int nv, const unsigned char *fixed, int ne,
                                const int *v0, const int *v1) {
  memset(S, 0, sizeof(*S));
  S->hidx = (int *)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
  int nb = 0;
  for (int v = 0; v < nv; ++v) S->hidx[v] = (fixed && fixed[v]) ? -1 : nb++;
  S->nb = nb; S->n = 7 * nb;
  int n = S->n;
  /* ordering */
  int *pa = (int *)malloc(sizeof(int) * (size_t)(ne > 0 ? ne : 1));
  int *pb = (int *)malloc(sizeof(int) * (size_t)(ne > 0 ? ne : 1));
  int np = 0;
  for (int k = 0; k < ne; ++k) {
    int a = S->hidx[v0[k]], b = S->hidx[v1[k]];
    if (a >= 0 && b >= 0 && a != b) { pa[np] = a; pb[np] = b; ++np; }
  }
  int *order = (int *)malloc(sizeof(int) * (size_t)(nb > 0 ? nb : 1));
  min_degree_order(nb, np, pa, pb, order);
  S->binv = (int *)malloc(sizeof(int) * (size_t)(nb > 0 ? nb : 1));
  for (int k = 0; k < nb; ++k) S->binv[order[k]] = k;
  free(order);
  /* permuted upper block pattern */
  bpair *bp = (bpair *)malloc(sizeof(bpair) * (size_t)(np + nb + 1));
  int m = 0;
  for (int k = 0; k < nb; ++k) { bp[m].c = k; bp[m].r = k; ++m; }
  for (int k = 0; k < np; ++k) {
    int a = S->binv[pa[k]], b = S->binv[pb[k]];
    bp[m].c = a > b ? a : b; bp[m].r = a > b ? b : a; ++m;
  }
  free(pa); free(pb);
  qsort(bp, (size_t)m, sizeof(bpair), cmp_bpair);
  int u = 0;
  for (int k = 0; k < m; ++k)
    if (u == 0 || bp[u - 1].c != bp[k].c || bp[u - 1].r != bp[k].r) bp[u++] = bp[k];
  S->bcp = (int *)calloc((size_t)nb + 1, sizeof(int));
  S->bri = (int *)malloc(sizeof(int) * (size_t)(u > 0 ? u : 1));
  for (int k = 0; k < u; ++k) { S->bcp[bp[k].c + 1]++; S->bri[k] = bp[k].r; }
  for (int c = 0; c < nb; ++c) S->bcp[c + 1] += S->bcp[c];
  free(bp);
  /* scalar CSC */
  S->Ap = (long *)malloc(sizeof(long) * ((size_t)n + 1));
  long nnz = 0;
  for (int c = 0; c < nb; ++c) {
    int nblk = S->bcp[c + 1] - S->bcp[c]; /* includes the diagonal block (last) */
    for (int k = 0; k < 7; ++k) { S->Ap[7 * c + k] = nnz; nnz += 7L * (nblk - 1) + (k + 1); }
  }
  S->Ap[n] = nnz;
  S->Ai = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  S->Ax = (double *)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(double));
  S->Awork = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  for (int c = 0; c < nb; ++c) {
    int nblk = S->bcp[c + 1] - S->bcp[c];
    for (int k = 0; k < 7; ++k) {
      long p = S->Ap[7 * c + k];
      for (int a = 0; a < nblk - 1; ++a)
        for (int i = 0; i < 7; ++i) S->Ai[p++] = 7 * S->bri[S->bcp[c] + a] + i;
      for (int i = 0; i <= k; ++i) S->Ai[p++] = 7 * c + i;
    }
  }
  /* per-edge slots */
  S->eslot = (long *)malloc(sizeof(long) * (size_t)(ne > 0 ? ne : 1));
  S->etrans = (char *)malloc((size_t)(ne > 0 ? ne : 1));
  for (int k = 0; k < ne; ++k) {
    int a = S->hidx[v0[k]], b = S->hidx[v1[k]];
    S->eslot[k] = -1; S->etrans[k] = 0;
    if (a >= 0 && b >= 0 && a != b) {
      int p0 = S->binv[a], p1 = S->binv[b];
      if (p0 < p1) { S->eslot[k] = block_slot(S, p0, p1); S->etrans[k] = 0; }
      else { S->eslot[k] = block_slot(S, p1, p0); S->etrans[k] = 1; }
    }
  }
  /* symbolic LDL^T: elimination tree and column counts (up-looking) */
  S->Parent = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  S->Lnz = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  S->Flag = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  S->Pattern = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  S->Lp = (long *)malloc(sizeof(long) * ((size_t)n + 1));
  for (int k = 0; k < n; ++k) {
    S->Parent[k] = -1; S->Flag[k] = k; S->Lnz[k] = 0;
    for (long p = S->Ap[k]; p < S->Ap[k + 1]; ++p) {
      int i = S->Ai[p];
      if (i < k)
        for (; S->Flag[i] != k; i = S->Parent[i]) {
          if (S->Parent[i] == -1) S->Parent[i] = k;
          S->Lnz[i]++;
          S->Flag[i] = k;
        }
    }
  }
  S->Lp[0] = 0;
  for (int k = 0; k < n; ++k) S->Lp[k + 1] = S->Lp[k] + S->Lnz[k];
  S->lnz = S->Lp[n];
  g_last_lnz = S->lnz;
  S->Li = (int *)malloc(sizeof(int) * (size_t)(S->lnz > 0 ? S->lnz : 1));
  S->Lx = (double *)malloc(sizeof(double) * (size_t)(S->lnz > 0 ? S->lnz : 1));
  S->D = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  S->Y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  S->b = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
}

/* numeric LDL^T of Awork; returns 1 if all pivots are positive */
static int sys_factor(sys_t *S) {
  int n = S->n;
  const long *Ap = S->Ap; const int *Ai = S->Ai; const double *Ax = S->Awork;
  for (int k = 0; k < n; ++k) {
    int top = n;
    S->Y[k] = 0.0; S->Flag[k] = k; S->Lnz[k] = 0;
    for (long p = Ap[k]; p < Ap[k + 1]; ++p) {
      int i = Ai[p];
      if (i <= k) {
        S->Y[i] += Ax[p];
        int len = 0;
        for (; S->Flag[i] != k; i = S->Parent[i]) { S->Pattern[len++] = i; S->Flag[i] = k; }
        while (len > 0) S->Pattern[--top] = S->Pattern[--len];
      }
    }
    double dk = S->Y[k];
    S->Y[k] = 0.0;
    for (; top < n; ++top) {
      int i = S->Pattern[top];
      double yi = S->Y[i];
      S->Y[i] = 0.0;
      long p2 = S->Lp[i] + S->Lnz[i];
      for (long p = S->Lp[i]; p < p2; ++p) S->Y[S->Li[p]] -= S->Lx[p] * yi;
      double lki = yi / S->D[i];
      dk -= lki * yi;
      S->Li[p2] = k;
      S->Lx[p2] = lki;
      S->Lnz[i]++;
    }
    S->D[k] = dk;
    if (!(dk > 0.0)) return 0; /* not positive definite (or NaN) */
  }
  return 1;
}

/* x (ORIGINAL free order) = (P^T L D L^T P)^-1 b */
static void sys_solve(const sys_t *S, double *x) {
  int n = S->n, nb = S->nb;
  double *y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int blk = 0; blk < nb; ++blk)
    for (int i = 0; i < 7; ++i) y[7 * S->binv[blk] + i] = S->b[7 * blk + i];
  for (int j = 0; j < n; ++j) {
    long p2 = S->Lp[j] + S->Lnz[j];
    for (long p = S->Lp[j]; p < p2; ++p) y[S->Li[p]] -= S->Lx[p] * y[j];
  }
  for (int j = 0; j < n; ++j) y[j] /= S->D[j];
  for (int j = n - 1; j >= 0; --j) {
    long p2 = S->Lp[j] + S->Lnz[j];
    for (long p = S->Lp[j]; p < p2; ++p) y[j] -= S->Lx[p] * y[S->Li[p]];
  }
  for (int blk = 0; blk < nb; ++blk)
    for (int i = 0; i < 7; ++i) x[7 * blk + i] = y[7 * S->binv[blk] + i];
  free(y);
}

/* errors + Jacobians + quadratic forms for every edge, then serial assembly */
static void sys_linearize(sys_t *S, const double *states, int ne, const int *v0, const int *v1,
                          const double *meas, const double *info, int kernel, double kdelta,
                          const or_options *o, edge_quad *eq) {
  set_threads(o);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < ne; ++k) {
    int a = S->hidx[v0[k]], b = S->hidx[v1[k]];
    if (a < 0 && b < 0) continue;
    const or_sim3 *C = (const or_sim3 *)(meas + 8 * (size_t)k);
    const or_sim3 *S0 = (const or_sim3 *)(states + 8 * (size_t)v0[k]);
    const or_sim3 *S1 = (const or_sim3 *)(states + 8 * (size_t)v1[k]);
    double e[7], A[49], B[49];
    or_edge_error(C, S0, S1, o, e);
    memset(A, 0, sizeof(A)); memset(B, 0, sizeof(B));
    if (a >= 0) numeric_block(C, S0, S1, 0, o, A); /* fixed endpoints are skipped */
    if (b >= 0) numeric_block(C, S0, S1, 1, o, B);
    edge_quadratic(A, B, e, info ? info + 49 * (size_t)k : NULL, kernel, kdelta, &eq[k]);
  }
  memset(S->Ax, 0, sizeof(double) * (size_t)S->Ap[S->n]);
  memset(S->b, 0, sizeof(double) * (size_t)S->n);
  for (int k = 0; k < ne; ++k) {
    int a = S->hidx[v0[k]], b = S->hidx[v1[k]];
    if (a >= 0) {
      int c = S->binv[a];
      long p = S->bcp[c + 1] - 1; /* diagonal slot */
      for (int kk = 0; kk < 7; ++kk)
        for (int i = 0; i <= kk; ++i) S->Ax[scalar_pos(S, c, p, i, kk)] += eq[k].H00[7 * kk + i];
      for (int i = 0; i < 7; ++i) S->b[7 * a + i] += eq[k].b0[i];
    }
    if (b >= 0) {
      int c = S->binv[b];
      long p = S->bcp[c + 1] - 1;
      for (int kk = 0; kk < 7; ++kk)
        for (int i = 0; i <= kk; ++i) S->Ax[scalar_pos(S, c, p, i, kk)] += eq[k].H11[7 * kk + i];
      for (int i = 0; i < 7; ++i) S->b[7 * b + i] += eq[k].b1[i];
    }
    if (S->eslot[k] >= 0) {
      int p0 = S->binv[a], p1 = S->binv[b];
      int c = p0 < p1 ? p1 : p0;
      for (int kk = 0; kk < 7; ++kk)
        for (int i = 0; i < 7; ++i) {
          double v = S->etrans[k] ? eq[k].H01[7 * i + kk] : eq[k].H01[7 * kk + i];
          S->Ax[scalar_pos(S, c, S->eslot[k], i, kk)] += v;
        }
    }
  }
}

static double sys_max_diag(const sys_t *S) {
  double m = 0;
  for (int j = 0; j < S->n; ++j) {
    double d = fabs(S->Ax[S->Ap[j + 1] - 1]); /* diagonal is the last entry of each column */
    if (d > m) m = d;
  }
  return m;
}

static int sys_solve_lambda(sys_t *S, double lambda, double *x) {
  memcpy(S->Awork, S->Ax, sizeof(double) * (size_t)S->Ap[S->n]);
  for (int j = 0; j < S->n; ++j) S->Awork[S->Ap[j + 1] - 1] += lambda;
  if (!sys_factor(S)) return 0;
  sys_solve(S, x);
  return 1;
}

/* ------------------------------------------------------------------ */
int or_build_dense(int nv, const double *states, const unsigned char *fixed, int ne,
                   const int *v0, const int *v1, const double *meas, const double *info,
                   int kernel, double kdelta, const or_options *o, double *H, double *b) {
  int *hidx = (int *)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
  int nb = 0;
  for (int v = 0; v < nv; ++v) hidx[v] = (fixed && fixed[v]) ? -1 : nb++;
  int n = 7 * nb;
  memset(H, 0, sizeof(double) * (size_t)n * (size_t)n);
  memset(b, 0, sizeof(double) * (size_t)n);
  for (int k = 0; k < ne; ++k) {
    int a = hidx[v0[k]], bb = hidx[v1[k]];
    if (a < 0 && bb < 0) continue;
    const or_sim3 *C = (const or_sim3 *)(meas + 8 * (size_t)k);
    const or_sim3 *S0 = (const or_sim3 *)(states + 8 * (size_t)v0[k]);
    const or_sim3 *S1 = (const or_sim3 *)(states + 8 * (size_t)v1[k]);
    double e[7], A[49], B[49];
    edge_quad q;
    or_edge_error(C, S0, S1, o, e);
    memset(A, 0, sizeof(A)); memset(B, 0, sizeof(B));
    if (a >= 0) numeric_block(C, S0, S1, 0, o, A);
    if (bb >= 0) numeric_block(C, S0, S1, 1, o, B);
    edge_quadratic(A, B, e, info ? info + 49 * (size_t)k : NULL, kernel, kdelta, &q);
    for (int c = 0; c < 7; ++c)
      for (int r = 0; r < 7; ++r) {
        if (a >= 0) H[(size_t)(7 * a + c) * n + 7 * a + r] += q.H00[7 * c + r];
        if (bb >= 0) H[(size_t)(7 * bb + c) * n + 7 * bb + r] += q.H11[7 * c + r];
        if (a >= 0 && bb >= 0 && a != bb) {
          H[(size_t)(7 * bb + c) * n + 7 * a + r] += q.H01[7 * c + r];
          H[(size_t)(7 * a + r) * n + 7 * bb + c] += q.H01[7 * c + r];
        }
      }
    for (int r = 0; r < 7; ++r) {
      if (a >= 0) b[7 * a + r] += q.b0[r];
      if (bb >= 0) b[7 * bb + r] += q.b1[r];
    }
  }
  free(hidx);
  return n;
}

int or_solve_once(int nv, const double *states, const unsigned char *fixed, int ne,
                  const int *v0, const int *v1, const double *meas, const double *info,
                  int kernel, double kdelta, const or_options *o, double lambda, double *x,
                  double *b_out) {
  sys_t S;
  sys_build_structure(&S, nv, fixed, ne, v0, v1);
  edge_quad *eq = (edge_quad *)malloc(sizeof(edge_quad) * (size_t)(ne > 0 ? ne : 1));
  sys_linearize(&S, states, ne, v0, v1, meas, info, kernel, kdelta, o, eq);
  int ok = sys_solve_lambda(&S, lambda, x);
  if (b_out) memcpy(b_out, S.b, sizeof(double) * (size_t)S.n);
  free(eq);
  sys_free(&S);
  return ok;
}

int or_optimize(int nv, double *states, const unsigned char *fixed, int ne, const int *v0,
                const int *v1, const double *meas, const double *info, int kernel, double kdelta,
                int max_iters, const or_options *o, or_iter *trace) {
  sys_t S;
  sys_build_structure(&S, nv, fixed, ne, v0, v1);
  if (S.nb == 0 || ne == 0) { sys_free(&S); return -1; }
  int n = S.n;
  edge_quad *eq = (edge_quad *)malloc(sizeof(edge_quad) * (size_t)ne);
  double *x = (double *)malloc(sizeof(double) * (size_t)n);
  double *backup = (double *)malloc(sizeof(double) * 8 * (size_t)nv);
  double lambda = 0, ni = 2;
  int iters = 0, ok = 1;
  for (int it = 0; it < max_iters && ok; ++it) {
    or_iter *T = &trace[it];
    memset(T, 0, sizeof(*T));
    double t0 = now_s();
    double currentChi = or_chi2(nv, states, ne, v0, v1, meas, info, kernel, kdelta, o);
    double tempChi = currentChi;
    T->chi2_before = currentChi;
    sys_linearize(&S, states, ne, v0, v1, meas, info, kernel, kdelta, o, eq);
    double t1 = now_s();
    T->t_linearize = t1 - t0;
    if (it == 0) {
      lambda = o->user_lambda_init > 0 ? o->user_lambda_init : o->tau * sys_max_diag(&S);
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
      memcpy(backup, states, sizeof(double) * 8 * (size_t)nv); /* push */
      double ts = now_s();
      int ok2 = sys_solve_lambda(&S, lambda, x);
      if (!ok2) memset(x, 0, sizeof(double) * (size_t)n);
      double tu = now_s();
      T->t_solve += tu - ts;
      for (int v = 0; v < nv; ++v) { /* update: S <- exp(dx) * S */
        int h = S.hidx[v];
        if (h < 0) continue;
        or_sim3 P, R;
        or_sim3_exp(x + 7 * h, o, &P);
        or_sim3_mul(&P, (const or_sim3 *)(states + 8 * (size_t)v), &R);
        memcpy(states + 8 * (size_t)v, &R, sizeof(R));
      }
      tempChi = or_chi2(nv, states, ne, v0, v1, meas, info, kernel, kdelta, o);
      if (!ok2) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < n; ++j) scale += x[j] * (lambda * x[j] + S.b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1.0 - pow(2 * rho - 1, 3);
        alpha = alpha < o->good_step_upper ? alpha : o->good_step_upper;
        double sf = o->good_step_lower > alpha ? o->good_step_lower : alpha;
        lambda *= sf;
        ni = 2;
        currentChi = tempChi; /* discardTop */
      } else {
        lambda *= ni;
        ni *= 2;
        memcpy(states, backup, sizeof(double) * 8 * (size_t)nv); /* pop */
      }
      T->solve_ok = ok2;
      T->t_update += now_s() - tu;
      qmax++;
    } while (rho < 0 && qmax < o->max_trials);
    T->chi2_after = currentChi;
    T->lambda = lambda;
    T->rho = rho;
    T->trials = qmax;
    ++iters;
    if (qmax == o->max_trials || rho == 0 || !isfinite(lambda)) ok = 0; /* Terminate */
  }
  free(eq); free(x); free(backup);
  sys_free(&S);
  return iters;
}
