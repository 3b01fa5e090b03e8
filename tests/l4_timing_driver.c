/* tests/l4_timing_driver.c -- what LD_PRELOAD=libm4ri_amd.so does to M4RI's own L4 routines (our own client
 * code against M4RI's public API; linked against the interposable reference build like dropin_driver.c):
 * times mzd_trsm_upper_left, mzd_ple, mzd_pluq, mzd_solve_left, mzd_echelonize, mzd_inv_m4ri, mzd_trtri_upper and
 * mzd_transpose at one size.  Run it with and without the preload;
 * the internal mzd_addmul / _mzd_addmul calls of those routines then run on the GPU or on the CPU. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <m4ri/m4ri.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv) {
  const rci_t n = argc > 1 ? atoi(argv[1]) : 16384;
  srandom(17);
  printf("l4_timing_driver n=%d, m4ri_amd interposed: %s\n", n, dlsym(RTLD_DEFAULT, "m4ri_amd_get_stats") ? "yes" : "no");
  mzd_t *U = mzd_init(n, n), *B = mzd_init(n, n), *A = mzd_init(n, n);
  mzd_randomize(U);
  mzd_randomize(B);
  mzd_randomize(A);
  for (rci_t i = 0; i < n; ++i) {
    mzd_row(U, i)[i / 64] |= (word)1 << (i % 64);
    for (wi_t w = 0; w < i / 64; ++w) mzd_row(U, i)[w] = 0;
    if (i % 64) mzd_row(U, i)[i / 64] &= ~(((word)1 << (i % 64)) - 1);
  }
  const int only_new = argc > 2 && argv[2][0] == 'n'; /* "new": only mzd_trtri_upper and mzd_transpose (the others take minutes on the CPU) */
  double t;
  word f9 = 0;
  mzd_t *X = mzd_copy(NULL, B), *A2 = mzd_copy(NULL, A), *A4 = mzd_copy(NULL, A), *A3 = mzd_copy(NULL, A), *Y = mzd_copy(NULL, B), *A5 = mzd_copy(NULL, A), *Ai = A;
  mzp_t *P = mzp_init(n), *Q = mzp_init(n);
  rci_t r;
  if (!only_new) {
  t = now();
  mzd_trsm_upper_left(U, X, 0);
  printf("  mzd_trsm_upper_left %d x %d : %.3f s\n", n, n, now() - t);
  t = now();
  r = mzd_ple(A2, P, Q, 0);
  printf("  mzd_ple             %d x %d : %.3f s (rank %d)\n", n, n, now() - t, r);
  for (rci_t i = 0; i < n; ++i) mzd_row(A4, i)[(n / 3) / 64] = 0; /* 64 empty columns: pivots move, the column step has work */
  t = now();
  r = mzd_pluq(A4, P, Q, 0);
  printf("  mzd_pluq            %d x %d : %.3f s (rank %d)\n", n, n, now() - t, r);
  t = now();
  int st = mzd_solve_left(A3, Y, 0, 0);
  printf("  mzd_solve_left      %d x %d : %.3f s (status %d)\n", n, n, now() - t, st);
  for (rci_t i = 0; i < n; ++i) mzd_row(A5, i)[(n / 5) / 64] = 0;
  t = now();
  r = mzd_echelonize(A5, 1);
  printf("  mzd_echelonize full %d x %d : %.3f s (rank %d)\n", n, n, now() - t, r);
  /* an invertible matrix the way bench/bench_invert.c:18-40 builds one: unit lower triangular times unit upper triangular
   * (a random matrix of this size is singular more often than not, and mzd_inv_m4ri is meant for the other case) */
  mzd_t *Lm = mzd_copy(NULL, B);
  for (rci_t i = 0; i < n; ++i) {
    for (wi_t w = i / 64 + 1; w < Lm->width; ++w) mzd_row(Lm, i)[w] = 0;
    mzd_row(Lm, i)[i / 64] &= ((word)1 << (i % 64)) - 1;
    mzd_row(Lm, i)[i / 64] |= (word)1 << (i % 64);
  }
  mzd_t *LU = mzd_mul(NULL, Lm, U, 0);
  t = now();
  Ai = mzd_inv_m4ri(NULL, LU, 0);
  printf("  mzd_inv_m4ri        %d x %d : %.3f s (invertible input)\n", n, n, now() - t);
  t = now();
  mzd_t *As = mzd_inv_m4ri(NULL, A, 0);
  printf("  mzd_inv_m4ri        %d x %d : %.3f s (singular input: the augmented elimination's leftovers)\n", n, n, now() - t);
  for (rci_t i = 0; i < n; ++i)
    for (wi_t w = 0; w < As->width; ++w) f9 = f9 * 1099511628211ull ^ mzd_row(As, i)[w];
  }
  mzd_t *Ui = mzd_copy(NULL, U);
  t = now();
  mzd_trtri_upper(Ui);
  printf("  mzd_trtri_upper     %d x %d : %.3f s\n", n, n, now() - t);
  t = now();
  mzd_t *At = mzd_transpose(NULL, A);
  printf("  mzd_transpose       %d x %d : %.3f s\n", n, n, now() - t);
  /* fingerprints so that the two runs can be compared */
  word f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0, f8 = 0;
  for (rci_t i = 0; i < n; ++i)
    for (wi_t w = 0; w < X->width; ++w) { f1 = f1 * 1099511628211ull ^ mzd_row(X, i)[w]; f2 = f2 * 1099511628211ull ^ mzd_row(A2, i)[w];
      f3 = f3 * 1099511628211ull ^ mzd_row(A4, i)[w]; f4 = f4 * 1099511628211ull ^ mzd_row(Y, i)[w];
      f5 = f5 * 1099511628211ull ^ mzd_row(A5, i)[w]; f6 = f6 * 1099511628211ull ^ mzd_row(Ai, i)[w];
      f7 = f7 * 1099511628211ull ^ mzd_row(Ui, i)[w]; f8 = f8 * 1099511628211ull ^ mzd_row(At, i)[w]; }
  printf("  fingerprints: trsm %016llx ple %016llx pluq %016llx solve %016llx echelon %016llx inverse %016llx trtri %016llx transpose %016llx inverse(singular) %016llx\n",
         (unsigned long long)f1, (unsigned long long)f2, (unsigned long long)f3, (unsigned long long)f4, (unsigned long long)f5, (unsigned long long)f6,
         (unsigned long long)f7, (unsigned long long)f8, (unsigned long long)f9);
  return 0;
}
