/* tests/dropin_driver.c -- a plain M4RI client program (our own code, written against M4RI's public
 * API only) used to demonstrate the drop-in boundary:
 *
 *     LD_PRELOAD=m4ri_amd/libm4ri_amd.so  oracle/_ref/dropin_driver
 *
 * It is linked against the interposable reference build (oracle/_ref/libm4ri_plt.so).  With the
 * preload, every mzd_mul / mzd_addmul / mzd_mul_m4rm / mzd_addmul_m4rm / _mzd_addmul call below
 * binds to libm4ri_amd.so (GPU) while mzd_init, mzd_randomize, mzd_mul_naive, mzd_add, mzd_equal,
 * mzd_free ... stay the reference's (CPU).  Each case checks the interposed product against
 * mzd_mul_naive -- the same differential style as the reference's tests/test_multiplication.c,
 * whose shape list (test_multiplication.c:251-286) this restates.  Without the preload it is an
 * ordinary self-test of the reference. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <m4ri/m4ri.h>
#include <stdio.h>
#include <stdlib.h>

static int mul_case(rci_t m, rci_t l, rci_t n, int k, int cutoff) {
  mzd_t *A = mzd_init(m, l), *B = mzd_init(l, n);
  mzd_randomize(A);
  mzd_randomize(B);
  mzd_t *C = mzd_mul(NULL, A, B, cutoff);       /* C == NULL: result must be mzd_free()-able */
  mzd_t *D = mzd_mul_m4rm(NULL, A, B, k);
  mzd_t *E = mzd_mul_naive(NULL, A, B);
  int bad = !mzd_equal(C, E) || !mzd_equal(D, E);
  /* accumulate on top: C += A*B twice must give C back */
  mzd_t *F = mzd_copy(NULL, C);
  mzd_addmul(F, A, B, cutoff);
  mzd_addmul_m4rm(F, A, B, k);
  bad |= !mzd_equal(F, C);
  _mzd_addmul(F, A, B, cutoff > 0 ? cutoff : 64);
  mzd_t *Z = mzd_init(m, n);
  bad |= !mzd_equal(F, Z);
  printf("  mul %5d x %5d x %5d  k=%d cutoff=%4d : %s\n", m, l, n, k, cutoff, bad ? "FAILED" : "ok");
  mzd_free(A); mzd_free(B); mzd_free(C); mzd_free(D); mzd_free(E); mzd_free(F); mzd_free(Z);
  return bad;
}

static int window_case(rci_t M, rci_t N, rci_t m, rci_t n) {
  /* products written into a window with non-zero excess must leave the parent's other bits alone
   * (the property tests/test_smallops.c checks with mzd_check_pattern) */
  const word pattern = 0x03030303030303ull;
  mzd_t *P = mzd_init(M, N);
  for (rci_t i = 0; i < M; ++i)
    for (wi_t j = 0; j < P->width; ++j) mzd_row(P, i)[j] = pattern;
  rci_t k = m < n ? m : n;
  mzd_t *A = mzd_init(m, k), *B = mzd_init(k, n);
  mzd_randomize(A);
  mzd_randomize(B);
  mzd_t *W = mzd_init_window(P, 0, 0, m, n);
  mzd_mul(W, A, B, 0);
  mzd_t *E = mzd_mul_naive(NULL, A, B);
  int bad = !mzd_equal(W, E);
  for (rci_t i = 0; i < M; ++i)
    for (wi_t j = 0; j < P->width; ++j) {
      word got = mzd_row(P, i)[j], keep = ~(word)0;
      if (i < m && j < W->width - 1) continue;                    /* inside the window            */
      if (i < m && j == W->width - 1) keep = ~W->high_bitmask;    /* its last word: excess bits   */
      if ((got & keep) != (pattern & keep)) bad = 1;
    }
  printf("  window %4d x %4d in %4d x %4d : %s\n", m, n, M, N, bad ? "FAILED" : "ok");
  mzd_free_window(W); mzd_free(A); mzd_free(B); mzd_free(E); mzd_free(P);
  return bad;
}

/* The reference's own L4 routines: their internal mzd_addmul / _mzd_addmul calls on WINDOWS
 * (triangular.c:100,348,439,503, ple.c:126, solve.c:89) go through the PLT and land on the GPU too.
 * Checked by the defining identities with the reference's naive CPU product. */
static long long interposed_products(void) {
  typedef int (*stats_fn)(void *);
  stats_fn st = (stats_fn)dlsym(RTLD_DEFAULT, "m4ri_amd_get_stats");
  struct { int levels, leaf_launches; long long leaf_products; int lm, ll, ln, r; double a, b, c, d, cum_ms; long long cum_launches; } s; /* m4ri_amd_stats */
  return (st && st(&s) == 0) ? s.leaf_launches : -1;
}

static int l4_case(rci_t n, rci_t m, int cutoff) {
  int bad = 0;
  /* U: random unit upper triangular, L = U^T-shaped lower triangular */
  mzd_t *U = mzd_init(n, n), *L = mzd_init(n, n);
  mzd_randomize(U);
  mzd_randomize(L);
  for (rci_t i = 0; i < n; ++i) {
    for (rci_t j = 0; j < i; ++j) mzd_write_bit(U, i, j, 0);
    for (rci_t j = i + 1; j < n; ++j) mzd_write_bit(L, i, j, 0);
    mzd_write_bit(U, i, i, 1);
    mzd_write_bit(L, i, i, 1);
  }
  mzd_t *B0 = mzd_init(n, m);
  mzd_randomize(B0);
  mzd_t *X = mzd_copy(NULL, B0);
  mzd_trsm_upper_left(U, X, cutoff);                 /* U X = B0 */
  mzd_t *Chk = mzd_mul_naive(NULL, U, X);
  bad |= !mzd_equal(Chk, B0);
  mzd_free(Chk);
  mzd_copy(X, B0);
  mzd_trsm_lower_left(L, X, cutoff);                 /* L X = B0 */
  Chk = mzd_mul_naive(NULL, L, X);
  bad |= !mzd_equal(Chk, B0);
  mzd_free(Chk);
  /* solve A X = B0 with A = L*U (invertible): PLE + both TRSMs inside */
  mzd_t *A = mzd_mul_naive(NULL, L, U), *A0 = mzd_copy(NULL, A);
  mzd_copy(X, B0);
  bad |= mzd_solve_left(A, X, cutoff, 0) != 0;
  Chk = mzd_mul_naive(NULL, A0, X);
  bad |= !mzd_equal(Chk, B0);
  mzd_free(Chk);
  /* PLE of a rank-deficient matrix: rank must match naive Gaussian elimination */
  mzd_t *R = mzd_init(n, n);
  mzd_randomize(R);
  for (rci_t i = n / 2; i < n; ++i) mzd_copy_row(R, i, R, i - n / 2);   /* rank <= n/2 */
  mzd_t *R2 = mzd_copy(NULL, R);
  mzp_t *P = mzp_init(n), *Q = mzp_init(n);
  const rci_t r1 = mzd_ple(R, P, Q, cutoff), r2 = mzd_echelonize_naive(R2, 0);
  bad |= r1 != r2;
  printf("  L4 n=%d m=%d cutoff=%d (trsm upper/lower, solve_left, ple rank %d/%d) : %s\n", n, m, cutoff, r1, r2, bad ? "FAILED" : "ok");
  mzp_free(P); mzp_free(Q);
  mzd_free(U); mzd_free(L); mzd_free(B0); mzd_free(X); mzd_free(A); mzd_free(A0); mzd_free(R); mzd_free(R2);
  return bad;
}

int main(void) {
  int status = 0;
  srandom(17);
  typedef int (*stats_fn)(void *);
  stats_fn st = (stats_fn)dlsym(RTLD_DEFAULT, "m4ri_amd_get_stats");
  printf("dropin_driver: m4ri_amd interposed: %s\n", st ? "yes" : "no (plain reference run)");
  static const int cases[][5] = {
      {1, 1, 1, 0, 1024}, {1, 128, 128, 0, 0}, {3, 131, 257, 0, 0}, {64, 64, 64, 0, 64}, {128, 128, 128, 0, 64},
      {21, 171, 31, 0, 63}, {21, 171, 31, 0, 131}, {193, 65, 65, 8, 64}, {1025, 1025, 1025, 3, 256},
      {2048, 2048, 4096, 0, 1024}, {4096, 3528, 4096, 0, 1024}, {1024, 1025, 1, 0, 1024}, {1000, 1000, 1000, 0, 256},
      {1000, 10, 20, 0, 64}, {1710, 1290, 1000, 0, 256}, {1290, 1710, 200, 0, 64}, {1290, 1710, 2000, 0, 256},
      {1290, 1290, 2000, 0, 64}, {1000, 210, 200, 0, 64}};
  for (unsigned i = 0; i < sizeof(cases) / sizeof(cases[0]); ++i)
    status += mul_case(cases[i][0], cases[i][1], cases[i][2], cases[i][3], cases[i][4]);
  status += window_case(64, 64, 10, 10);
  status += window_case(100, 100, 64, 64);
  status += window_case(1024, 1024, 513, 511);
  status += window_case(1024, 1024, 512, 768 + 30);
  status += window_case(2048, 2048, 1024, 1024);
  status += l4_case(2048, 1000, 0);
  status += l4_case(3000, 3000, 512);
  if (interposed_products() >= 0) printf("dropin_driver: the L4 cases ended on an interposed product (%lld leaf launch(es))\n", interposed_products());
  if (st) {
    struct { int levels, leaf_launches; long long leaf_products; int lm, ll, ln, r; double a, b, c, d, cum_ms; long long cum_launches; } s; /* m4ri_amd_stats */
    if (st(&s) == 0) printf("dropin_driver: last interposed call used %d leaf launch(es)\n", s.leaf_launches);
  }
  printf(status ? "dropin_driver: FAILED\n" : "dropin_driver: ALL OK\n");
  return status != 0;
}
