/* oracle_sanitizer_driver.c -- the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizers.py):
 * a small loop-closed chain with noisy measurements, parallel edges, dense information and a Huber kernel through every
 * entry point of oracle/sim3_oracle.h.  Prints the chi2 trace; any sanitizer report ends the process with an error. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sim3_oracle.h"

static unsigned long long lcg = 88172645463325252ull;
static double urand(void) {
  lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
}

int main(void) {
  enum { NV = 60 };
  or_options o;
  or_options_default(&o);
  o.fix_small_angle_b = 1;
  double *gt = malloc(sizeof(double) * 8 * NV), *st = malloc(sizeof(double) * 8 * NV);
  unsigned char fixed[NV];
  memset(fixed, 0, sizeof fixed);
  fixed[0] = 1;
  or_sim3 cur = {{0, 0, 0, 1}, {0, 0, 0}, 1.0};
  for (int i = 0; i < NV; ++i) {
    memcpy(gt + 8 * i, &cur, sizeof cur);
    double xi[7] = {0.05 * urand(), 0.05 * urand(), 0.3 + 0.1 * urand(), 1.0, 0.1 * urand(), 0.05 * urand(), 0.02 * urand()};
    or_sim3 step, nxt;
    or_sim3_exp(xi, &o, &step);
    or_sim3_mul(&step, &cur, &nxt);
    cur = nxt;
  }
  int ne = 0, cap = 3 * NV;
  int *v0 = malloc(sizeof(int) * cap), *v1 = malloc(sizeof(int) * cap);
  double *meas = malloc(sizeof(double) * 8 * cap), *info = malloc(sizeof(double) * 49 * cap);
  for (int pass = 0; pass < 3; ++pass)
    for (int i = 1; i < NV; ++i) {
      int a = i, b = pass == 0 ? i - 1 : (pass == 1 ? (i * 7) % NV : i - 1);  /* odometry, loops, parallel edges */
      if (a == b || (pass == 2 && i % 9)) continue;
      or_sim3 Sa, Sb, Sbi, T, N, C;
      memcpy(&Sa, gt + 8 * a, sizeof Sa);
      memcpy(&Sb, gt + 8 * b, sizeof Sb);
      or_sim3_inv(&Sa, &Sbi);            /* C = noise * S_b * S_a^-1: zero residual without the noise */
      or_sim3_mul(&Sb, &Sbi, &T);
      double n[7];
      for (int d = 0; d < 7; ++d) n[d] = 0.01 * urand();
      or_sim3_exp(n, &o, &N);
      or_sim3_mul(&N, &T, &C);
      v0[ne] = a; v1[ne] = b;
      memcpy(meas + 8 * ne, &C, sizeof C);
      for (int r = 0; r < 7; ++r)
        for (int c = 0; c < 7; ++c) info[49 * ne + 7 * c + r] = (r == c ? 1.0 + 0.1 * r : 0.0) + (r + c == 6 && r != c ? 0.05 : 0.0);
      ++ne;
    }
  /* dead-reckoned start with drift */
  memcpy(st, gt, sizeof(double) * 8 * NV);
  for (int i = 1; i < NV; ++i) {
    double xi[7] = {0.002 * i * urand(), 0, 0, 0.01 * i * urand(), 0.01 * i * urand(), 0, 0.001 * i};
    or_sim3 D, S, R;
    or_sim3_exp(xi, &o, &D);
    memcpy(&S, st + 8 * i, sizeof S);
    or_sim3_mul(&D, &S, &R);
    memcpy(st + 8 * i, &R, sizeof R);
  }
  double *e = malloc(sizeof(double) * 7 * ne), *A = malloc(sizeof(double) * 49 * ne), *B = malloc(sizeof(double) * 49 * ne);
  or_all_errors(ne, v0, v1, meas, st, &o, e);
  or_all_jacobians(ne, v0, v1, meas, st, &o, A, B);
  const int n = 7 * (NV - 1);
  double *H = malloc(sizeof(double) * (size_t)n * n), *b = malloc(sizeof(double) * n), *x = malloc(sizeof(double) * n);
  if (or_build_dense(NV, st, fixed, ne, v0, v1, meas, info, 1, 0.5, &o, H, b) != n) return 2;
  if (!or_solve_once(NV, st, fixed, ne, v0, v1, meas, info, 1, 0.5, &o, 1e-3, x, b)) return 3;
  for (int variant = 0; variant < 3; ++variant) {
    double *s2 = malloc(sizeof(double) * 8 * NV);
    memcpy(s2, st, sizeof(double) * 8 * NV);
    or_iter tr[12];
    o.fix_small_angle_b = variant != 2;
    const int it = or_optimize(NV, s2, fixed, ne, v0, v1, meas, variant ? info : NULL, variant ? 1 : 0, 0.5, 12, &o, tr);
    printf("variant %d: %d iterations, chi2 %.6g -> %.6g, fill %ld\n", variant, it, tr[0].chi2_before,
           it > 0 ? tr[it - 1].chi2_after : -1.0, or_last_lnz());
    if (it <= 0 || !(tr[it - 1].chi2_after <= tr[0].chi2_before)) return 4;
    free(s2);
  }
  /* nothing to optimise: every vertex fixed */
  memset(fixed, 1, sizeof fixed);
  or_iter tr1[2];
  printf("all fixed: %d\n", or_optimize(NV, st, fixed, ne, v0, v1, meas, NULL, 0, 0.0, 2, &o, tr1));
  free(gt); free(st); free(v0); free(v1); free(meas); free(info); free(e); free(A); free(B); free(H); free(b); free(x);
  printf("ok\n");
  return 0;
}
