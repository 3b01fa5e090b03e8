/* oracle/gf2_oracle.c -- TEST INFRASTRUCTURE ONLY (see gf2_oracle.h).
 *
 * CPU restatement of the reference's mzd_mul path.  Every function names the reference lines whose
 * OBSERVABLE behaviour it reproduces (result bits, window/excess rules, fatal errors); the code is
 * our own and deliberately simple -- one lookup table at a time, direct table indexing instead of
 * the codebook's ord/inc arrays, no SSE2, no caches.  Tuning knobs (k, cutoff) change the order of
 * operations only, never a bit of the result, exactly as in the reference.
 */
#include "gf2_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RADIX 64
#define FFFF (~(gf2o_word)0)

static void die(const char *msg) { /* misc.c:36-42 m4ri_die: print + abort */
  fprintf(stderr, "gf2_oracle: %s\n", msg);
  abort();
}

/* __M4RI_LEFT_BITMASK(n), misc.h:272: the n lowest bits; n == 0 means a full word */
static gf2o_word left_mask(int n) { return FFFF >> ((RADIX - n) % RADIX); }

static gf2o_word *row_of(const gf2o_mat *M, int64_t r) { return M->data + r * M->rowstride; }

/* ---- allocation / views: mzd.c:142-185 --------------------------------------------------- */
gf2o_mat *gf2o_init(int32_t r, int32_t c) {
  gf2o_mat *A = (gf2o_mat *)calloc(1, sizeof(gf2o_mat));
  if (!A) die("out of memory");
  A->nrows        = r;
  A->ncols        = c;
  A->width        = c > 0 ? (c - 1) / RADIX + 1 : 0;
  A->rowstride    = (A->width & 1) ? A->width + 1 : A->width; /* even stride, mzd.c:147-148 */
  A->high_bitmask = left_mask(c % RADIX);
  A->flags        = (A->high_bitmask != FFFF) ? GF2O_FLAG_EXCESS : 0;
  if (r && c) {
    size_t bytes = (size_t)r * (size_t)A->rowstride * sizeof(gf2o_word);
    if (posix_memalign((void **)&A->data, 64, bytes)) die("out of memory");
    memset(A->data, 0, bytes);
  }
  return A;
}

gf2o_mat *gf2o_init_window(gf2o_mat *M, int32_t lowr, int32_t lowc, int32_t highr, int32_t highc) {
  if (lowc % RADIX) die("window must start on a word boundary"); /* assert at mzd.c:161 */
  gf2o_mat *W = (gf2o_mat *)calloc(1, sizeof(gf2o_mat));
  if (!W) die("out of memory");
  int32_t nrows = highr - lowr;
  if (M->nrows - lowr < nrows) nrows = M->nrows - lowr;
  W->nrows        = nrows;
  W->ncols        = highc - lowc;
  W->rowstride    = M->rowstride;
  W->width        = (W->ncols + RADIX - 1) / RADIX;
  W->high_bitmask = left_mask(W->ncols % RADIX);
  W->flags        = GF2O_FLAG_WINDOW | ((W->ncols % RADIX) ? GF2O_FLAG_EXCESS : 0);
  W->data         = M->data + (int64_t)lowr * M->rowstride + lowc / RADIX;
  return W;
}

void gf2o_free(gf2o_mat *A) {
  if (!A) return;
  if (!(A->flags & GF2O_FLAG_WINDOW)) free(A->data);
  free(A);
}

/* ---- deterministic fill: mzd.c:1282-1292 fill order with splitmix64 -------------------------- */
uint64_t gf2o_splitmix_next(uint64_t *state) {
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void gf2o_fill_splitmix(gf2o_mat *A, uint64_t seed) {
  if (A->width == 0) return;
  uint64_t st = seed;
  for (int64_t i = 0; i < A->nrows; ++i) {
    gf2o_word *row = row_of(A, i);
    for (int64_t j = 0; j + 1 < A->width; ++j) row[j] = gf2o_splitmix_next(&st);
    gf2o_word r = gf2o_splitmix_next(&st);
    row[A->width - 1] ^= (row[A->width - 1] ^ r) & A->high_bitmask;
  }
}

/* ---- element-wise helpers --------------------------------------------------------------------- */
/* _mzd_add, mzd.c:1471-1583: C = A ^ B over min rows; last word of C keeps its bits outside
 * C->high_bitmask; in-place and mixed-stride operands allowed. */
gf2o_mat *gf2o_add(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B) {
  int64_t nrows = A->nrows < B->nrows ? A->nrows : B->nrows;
  if (C->nrows < nrows) nrows = C->nrows;
  const int64_t w = A->width;
  if (w == 0) return C;
  for (int64_t i = 0; i < nrows; ++i) {
    const gf2o_word *a = row_of(A, i), *b = row_of(B, i);
    gf2o_word *c = row_of(C, i);
    for (int64_t j = 0; j + 1 < w; ++j) c[j] = a[j] ^ b[j];
    c[w - 1] ^= (a[w - 1] ^ b[w - 1] ^ c[w - 1]) & C->high_bitmask;
  }
  return C;
}

/* mzd_copy, mzd.c:1363-1382: masked copy, allocates when N == NULL */
gf2o_mat *gf2o_copy(gf2o_mat *N, const gf2o_mat *P) {
  if (N == P) return N;
  if (!N) N = gf2o_init(P->nrows, P->ncols);
  else if (N->nrows < P->nrows || N->ncols < P->ncols) die("copy: target matrix is too small");
  if (P->width == 0) return N;
  for (int64_t i = 0; i < P->nrows; ++i) {
    const gf2o_word *p = row_of(P, i);
    gf2o_word *n = row_of(N, i);
    for (int64_t j = 0; j + 1 < P->width; ++j) n[j] = p[j];
    n[P->width - 1] = (n[P->width - 1] & ~P->high_bitmask) | (p[P->width - 1] & P->high_bitmask);
  }
  return N;
}

/* mzd_set_ui(A, 0), mzd.c:1294-1302: clears valid bits only */
void gf2o_set_zero(gf2o_mat *A) {
  if (A->width == 0) return;
  for (int64_t i = 0; i < A->nrows; ++i) {
    gf2o_word *r = row_of(A, i);
    for (int64_t j = 0; j + 1 < A->width; ++j) r[j] = 0;
    r[A->width - 1] &= ~A->high_bitmask;
  }
}

/* mzd_equal, mzd.c:1314-1331: valid bits only */
int gf2o_equal(const gf2o_mat *A, const gf2o_mat *B) {
  if (A->nrows != B->nrows || A->ncols != B->ncols) return 0;
  if (A == B || A->width == 0) return 1;
  for (int64_t i = 0; i < A->nrows; ++i) {
    const gf2o_word *a = row_of(A, i), *b = row_of(B, i);
    for (int64_t j = 0; j + 1 < A->width; ++j)
      if (a[j] != b[j]) return 0;
    if ((a[A->width - 1] ^ b[A->width - 1]) & A->high_bitmask) return 0;
  }
  return 1;
}

uint64_t gf2o_fingerprint(const gf2o_mat *A) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (int64_t i = 0; i < A->nrows; ++i) {
    const gf2o_word *a = row_of(A, i);
    for (int64_t j = 0; j < A->width; ++j) {
      gf2o_word v = a[j];
      if (j == A->width - 1) v &= A->high_bitmask;
      for (int b = 0; b < 8; ++b) {
        h ^= (v >> (8 * b)) & 0xff;
        h *= 0x100000001b3ull;
      }
    }
  }
  return h;
}

/* mzd_read_bits, mzd.h:892-901: n <= 64 bits starting at column y, column y lands on bit 0 */
static gf2o_word read_bits(const gf2o_mat *M, int64_t x, int64_t y, int n) {
  const int spot     = (int)(y % RADIX);
  const int64_t blk  = y / RADIX;
  const int spill    = spot + n - RADIX;
  const gf2o_word *r = row_of(M, x);
  gf2o_word t = (spill <= 0) ? r[blk] << -spill : (r[blk + 1] << (RADIX - spill)) | (r[blk] >> spill);
  return t >> (RADIX - n);
}

/* row XOR of the valid words of B's row j into C's row i; B's last word is masked so C's bits at
 * columns >= ncols are never disturbed (mzd_combine / mzd.h:993-1048 on whole rows). */
static void xor_row(gf2o_mat *C, int64_t i, const gf2o_mat *B, int64_t j) {
  gf2o_word *c = row_of(C, i);
  const gf2o_word *b = row_of(B, j);
  for (int64_t w = 0; w + 1 < B->width; ++w) c[w] ^= b[w];
  c[B->width - 1] ^= b[B->width - 1] & B->high_bitmask;
}

/* ---- definitional product: _mzd_mul_va, mzd.c:1256-1268 --------------------------------------- */
gf2o_mat *gf2o_mul_naive(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int clear) {
  if (A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols) die("mul_naive: wrong dimensions");
  if (clear) gf2o_set_zero(C);
  if (B->width == 0) return C;
  for (int64_t i = 0; i < A->nrows; ++i)
    for (int64_t j = 0; j < A->ncols; ++j)
      if ((row_of(A, i)[j / RADIX] >> (j % RADIX)) & 1) xor_row(C, i, B, j);
  return C;
}

/* ---- Method of the Four Russians: _mzd_mul_m4rm, brilliantrussian.c:1032-1190 ---------------- */
/* One table of 2^k combinations of k consecutive rows of B at a time (the reference keeps eight;
 * same sums).  T[x] = XOR of rows r+b for the set bits b of x, built in Gray-code order so each
 * entry is one row XOR (mzd_make_table, :163-211; the doc table in graycode.h is stale, the code's
 * ord[] is the reflected Gray code i^(i>>1)).  Table rows carry zero excess bits (:165-168,:206),
 * which is what preserves the excess bits of a windowed C.  Bit b of the index is column r+b of A
 * (mzd_read_bits puts column r on bit 0). */
gf2o_mat *gf2o_mul_m4rm(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int k, int clear) {
  if (A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols) die("mul_m4rm: wrong dimensions");
  if (clear) gf2o_set_zero(C);
  if (C->nrows == 0 || C->ncols == 0 || A->ncols == 0) return C;
  if (k <= 0) k = 8; /* :1075-1089 picks k from cache sizes; any k gives the same bits */
  if (k > 8) k = 8;
  const int64_t w = B->width, l = A->ncols;
  gf2o_word *T = (gf2o_word *)malloc(((size_t)1 << k) * (size_t)w * sizeof(gf2o_word));
  if (!T) die("out of memory");
  for (int64_t r = 0; r < l; r += k) {
    const int kk = (l - r < k) ? (int)(l - r) : k; /* tails :1157-1178 */
    memset(T, 0, (size_t)w * sizeof(gf2o_word));
    unsigned prev = 0;
    for (unsigned i = 1; i < (1u << kk); ++i) {
      const unsigned g = i ^ (i >> 1);
      const int bit    = __builtin_ctz(i); /* gray(i) ^ gray(i-1) == 1 << ctz(i) */
      const gf2o_word *b = row_of(B, r + bit);
      gf2o_word *t = T + (size_t)g * w, *tp = T + (size_t)prev * w;
      for (int64_t j = 0; j + 1 < w; ++j) t[j] = tp[j] ^ b[j];
      t[w - 1] = tp[w - 1] ^ (b[w - 1] & B->high_bitmask);
      prev = g;
    }
    for (int64_t j = 0; j < A->nrows; ++j) {
      const gf2o_word x = read_bits(A, j, r, kk);
      const gf2o_word *t = T + (size_t)x * w;
      gf2o_word *c = row_of(C, j);
      for (int64_t q = 0; q < w; ++q) c[q] ^= t[q];
    }
  }
  free(T);
  return C;
}

/* ---- Strassen-Winograd, Bodrato's sequence ------------------------------------------------------ */
static int closer(int64_t a, int cutoff) { return 3 * a < 4 * (int64_t)cutoff; } /* strassen.c:39 */

static gf2o_mat *win(const gf2o_mat *M, int64_t r0, int64_t c0, int64_t r1, int64_t c1) {
  return gf2o_init_window((gf2o_mat *)M, (int32_t)r0, (int32_t)c0, (int32_t)r1, (int32_t)c1);
}

/* split sizes, strassen.c:69-80: halves are word-aligned and even down the whole recursion */
static void split_sizes(int64_t m, int64_t k, int64_t n, int cutoff, int64_t *mmm, int64_t *kkk, int64_t *nnn) {
  int64_t mult = RADIX, width = (m < n ? m : n);
  if (k < width) width = k;
  width /= 2;
  while (width > cutoff) { width /= 2; mult *= 2; }
  *mmm = (((m - m % mult) / RADIX) >> 1) * RADIX;
  *kkk = (((k - k % mult) / RADIX) >> 1) * RADIX;
  *nnn = (((n - n % mult) / RADIX) >> 1) * RADIX;
}

/* remainder strips after the even block, strassen.c:170-204 (mul) and :488-522 (addmul) */
static void peel(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int64_t mmm, int64_t kkk, int64_t nnn, int clear) {
  const int64_t m = A->nrows, k = A->ncols, n = B->ncols;
  nnn *= 2; mmm *= 2; kkk *= 2;
  if (n > nnn) { /* last columns: full A times the right strip of B */
    gf2o_mat *Bl = win(B, 0, nnn, k, n), *Cl = win(C, 0, nnn, m, n);
    gf2o_mul_m4rm(Cl, A, Bl, 0, clear);
    gf2o_free(Bl); gf2o_free(Cl);
  }
  if (m > mmm) { /* last rows */
    gf2o_mat *Al = win(A, mmm, 0, m, k), *Bf = win(B, 0, 0, k, nnn), *Cl = win(C, mmm, 0, m, nnn);
    gf2o_mul_m4rm(Cl, Al, Bf, 0, clear);
    gf2o_free(Al); gf2o_free(Bf); gf2o_free(Cl);
  }
  if (k > kkk) { /* last inner slab: always accumulates into the bulk */
    gf2o_mat *Al = win(A, 0, kkk, mmm, k), *Bl = win(B, kkk, 0, k, nnn), *Cb = win(C, 0, 0, mmm, nnn);
    gf2o_mul_m4rm(Cb, Al, Bl, 0, 0);
    gf2o_free(Al); gf2o_free(Bl); gf2o_free(Cb);
  }
}

/* _mzd_mul_even, strassen.c:41-208 */
gf2o_mat *gf2o_mul_even(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff) {
  if (C->nrows == 0 || C->ncols == 0) return C;
  const int64_t m = A->nrows, k = A->ncols, n = B->ncols;
  if (closer(m, cutoff) || closer(k, cutoff) || closer(n, cutoff)) /* leaf :51-67 */
    return gf2o_mul_m4rm(C, A, B, 0, 1);

  int64_t mmm, kkk, nnn;
  split_sizes(m, k, n, cutoff, &mmm, &kkk, &nnn);

  gf2o_mat *A11 = win(A, 0, 0, mmm, kkk), *A12 = win(A, 0, kkk, mmm, 2 * kkk);
  gf2o_mat *A21 = win(A, mmm, 0, 2 * mmm, kkk), *A22 = win(A, mmm, kkk, 2 * mmm, 2 * kkk);
  gf2o_mat *B11 = win(B, 0, 0, kkk, nnn), *B12 = win(B, 0, nnn, kkk, 2 * nnn);
  gf2o_mat *B21 = win(B, kkk, 0, 2 * kkk, nnn), *B22 = win(B, kkk, nnn, 2 * kkk, 2 * nnn);
  gf2o_mat *C11 = win(C, 0, 0, mmm, nnn), *C12 = win(C, 0, nnn, mmm, 2 * nnn);
  gf2o_mat *C21 = win(C, mmm, 0, 2 * mmm, nnn), *C22 = win(C, mmm, nnn, 2 * mmm, 2 * nnn);

  /* 7 products, 15 additions, strassen.c:108-150 */
  gf2o_mat *X = gf2o_init((int32_t)mmm, (int32_t)kkk); /* Wmk */
  gf2o_mat *Y = gf2o_init((int32_t)kkk, (int32_t)nnn); /* Wkn */
  gf2o_add(Y, B22, B12);
  gf2o_add(X, A22, A12);
  gf2o_mul_even(C21, X, Y, cutoff);
  gf2o_add(X, A22, A21);
  gf2o_add(Y, B22, B21);
  gf2o_mul_even(C22, X, Y, cutoff);
  gf2o_add(Y, Y, B12);
  gf2o_add(X, X, A12);
  gf2o_mul_even(C11, X, Y, cutoff);
  gf2o_add(X, X, A11);
  gf2o_mul_even(C12, X, B12, cutoff);
  gf2o_add(C12, C12, C22);
  gf2o_free(X);
  X = gf2o_init((int32_t)mmm, (int32_t)nnn); /* :137: a fresh product A12*B21 */
  gf2o_mul_even(X, A12, B21, cutoff);
  gf2o_add(C11, C11, X);
  gf2o_add(C12, C11, C12);
  gf2o_add(C11, C21, C11);
  gf2o_add(Y, Y, B11);
  gf2o_mul_even(C21, A21, Y, cutoff);
  gf2o_free(Y);
  gf2o_add(C21, C11, C21);
  gf2o_add(C22, C22, C11);
  gf2o_mul_even(C11, A11, B11, cutoff);
  gf2o_add(C11, C11, X);
  gf2o_free(X);

  gf2o_free(A11); gf2o_free(A12); gf2o_free(A21); gf2o_free(A22);
  gf2o_free(B11); gf2o_free(B12); gf2o_free(B21); gf2o_free(B22);
  gf2o_free(C11); gf2o_free(C12); gf2o_free(C21); gf2o_free(C22);

  peel(C, A, B, mmm, kkk, nnn, 1);
  return C;
}

/* _mzd_addmul_even, strassen.c:367-526 */
gf2o_mat *gf2o_addmul_even(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff) {
  if (C->nrows == 0 || C->ncols == 0) return C;
  const int64_t m = A->nrows, k = A->ncols, n = B->ncols;
  if (closer(m, cutoff) || closer(k, cutoff) || closer(n, cutoff)) /* leaf :377-394 */
    return gf2o_mul_m4rm(C, A, B, 0, 0);

  int64_t mmm, kkk, nnn;
  split_sizes(m, k, n, cutoff, &mmm, &kkk, &nnn);

  gf2o_mat *A11 = win(A, 0, 0, mmm, kkk), *A12 = win(A, 0, kkk, mmm, 2 * kkk);
  gf2o_mat *A21 = win(A, mmm, 0, 2 * mmm, kkk), *A22 = win(A, mmm, kkk, 2 * mmm, 2 * kkk);
  gf2o_mat *B11 = win(B, 0, 0, kkk, nnn), *B12 = win(B, 0, nnn, kkk, 2 * nnn);
  gf2o_mat *B21 = win(B, kkk, 0, 2 * kkk, nnn), *B22 = win(B, kkk, nnn, 2 * kkk, 2 * nnn);
  gf2o_mat *C11 = win(C, 0, 0, mmm, nnn), *C12 = win(C, 0, nnn, mmm, 2 * nnn);
  gf2o_mat *C21 = win(C, mmm, 0, 2 * mmm, nnn), *C22 = win(C, mmm, nnn, 2 * mmm, 2 * nnn);

  /* 14 additions, 2 products + 5 accumulating products, strassen.c:436-466 */
  gf2o_mat *S = gf2o_init((int32_t)mmm, (int32_t)kkk);
  gf2o_mat *T = gf2o_init((int32_t)kkk, (int32_t)nnn);
  gf2o_mat *U = gf2o_init((int32_t)mmm, (int32_t)nnn);
  gf2o_add(S, A22, A21);
  gf2o_add(T, B22, B21);
  gf2o_mul_even(U, S, T, cutoff);
  gf2o_add(C22, U, C22);
  gf2o_add(C12, U, C12);
  gf2o_mul_even(U, A12, B21, cutoff);
  gf2o_add(C11, U, C11);
  gf2o_addmul_even(C11, A11, B11, cutoff);
  gf2o_add(S, S, A12);
  gf2o_add(T, T, B12);
  gf2o_addmul_even(U, S, T, cutoff);
  gf2o_add(C12, C12, U);
  gf2o_add(S, A11, S);
  gf2o_addmul_even(C12, S, B12, cutoff);
  gf2o_add(T, B11, T);
  gf2o_addmul_even(C21, A21, T, cutoff);
  gf2o_add(S, A22, A12);
  gf2o_add(T, B22, B12);
  gf2o_addmul_even(U, S, T, cutoff);
  gf2o_add(C21, C21, U);
  gf2o_add(C22, C22, U);
  gf2o_free(S); gf2o_free(T); gf2o_free(U);

  gf2o_free(A11); gf2o_free(A12); gf2o_free(A21); gf2o_free(A22);
  gf2o_free(B11); gf2o_free(B12); gf2o_free(B21); gf2o_free(B22);
  gf2o_free(C11); gf2o_free(C12); gf2o_free(C21); gf2o_free(C22);

  peel(C, A, B, mmm, kkk, nnn, 0);
  return C;
}

/* cutoff normalisation shared by mzd_mul / mzd_addmul, strassen.c:348-354 / :679-685.  The
 * reference's default is MIN(sqrt(4*L3), 4096) (strassen.h:133-135) = 4096 for any L3 >= 4 MiB. */
static int norm_cutoff(int cutoff, const char *who) {
  if (cutoff < 0) { fprintf(stderr, "gf2_oracle: %s: cutoff must be >= 0.\n", who); abort(); }
  if (cutoff == 0) cutoff = 4096;
  cutoff = cutoff / RADIX * RADIX;
  if (cutoff < RADIX) cutoff = RADIX;
  return cutoff;
}

/* mzd_mul, strassen.c:345-365.  A == B runs _mzd_sqr_even there (:210-343); its result is that of
 * the general routine with B := A, which is what this does. */
gf2o_mat *gf2o_mul(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff) {
  if (A->ncols != B->nrows) die("mul: A ncols need to match B nrows");
  cutoff = norm_cutoff(cutoff, "mul");
  if (!C) C = gf2o_init(A->nrows, B->ncols);
  else if (C->nrows != A->nrows || C->ncols != B->ncols) die("mul: C has wrong dimensions");
  return gf2o_mul_even(C, A, B, cutoff);
}

/* mzd_addmul, strassen.c:675-700 */
gf2o_mat *gf2o_addmul(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff) {
  if (A->ncols != B->nrows) die("addmul: A ncols need to match B nrows");
  cutoff = norm_cutoff(cutoff, "addmul");
  if (!C) C = gf2o_init(A->nrows, B->ncols);
  else if (C->nrows != A->nrows || C->ncols != B->ncols) die("addmul: C has wrong dimensions");
  if (A->nrows == 0 || A->ncols == 0 || B->ncols == 0) return C;
  return gf2o_addmul_even(C, A, B, cutoff);
}


/* ---- triangular solves (m4ri/triangular.c:406-455, :467-514) ----------------------------------------
 * X_i = B_i + sum_{k<i} L[i,k] X_k (lower) / sum_{k>i} U[i,k] X_k (upper), rows added whole with the last
 * word under B's column mask (triangular.c:421, :482).  The reference does this only for <= 64 rows and
 * recurses / uses tables above; the result does not depend on that. */
static int o_bit(const gf2o_mat *M, int64_t r, int64_t c) { return (int)((M->data[r * M->rowstride + c / 64] >> (c % 64)) & 1); }

static void o_row_add(gf2o_mat *B, int64_t dst, int64_t src) {
  gf2o_word *d = B->data + dst * B->rowstride;
  const gf2o_word *s = B->data + src * B->rowstride;
  for (int64_t j = 0; j + 1 < B->width; ++j) d[j] ^= s[j];
  if (B->width) d[B->width - 1] ^= s[B->width - 1] & B->high_bitmask;
}

void gf2o_trsm_lower_left(const gf2o_mat *L, gf2o_mat *B) {
  for (int64_t i = 1; i < B->nrows; ++i)
    for (int64_t k = 0; k < i; ++k)
      if (o_bit(L, i, k)) o_row_add(B, i, k);
}

void gf2o_trsm_upper_left(const gf2o_mat *U, gf2o_mat *B) {
  for (int64_t i = (int64_t)B->nrows - 2; i >= 0; --i)
    for (int64_t k = i + 1; k < B->nrows; ++k)
      if (o_bit(U, i, k)) o_row_add(B, i, k);
}


/* ---- PLE (m4ri/ple_russian.c:380-617, restated column by column) ------------------------------------- */
static void o_row_swap(gf2o_mat *A, int64_t a, int64_t b) {
  if (a == b) return;
  gf2o_word *x = A->data + a * A->rowstride, *y = A->data + b * A->rowstride;
  for (int64_t j = 0; j + 1 < A->width; ++j) { gf2o_word t = x[j]; x[j] = y[j]; y[j] = t; }
  if (A->width) {  /* last word under the column mask (mzd.h: mzd_row_swap masks the last word) */
    const gf2o_word m = A->high_bitmask, t = (x[A->width - 1] ^ y[A->width - 1]) & m;
    x[A->width - 1] ^= t; y[A->width - 1] ^= t;
  }
}

/* dst ^= src from column `col` to the end (mzd.h: mzd_row_add_offset) */
static void o_row_add_from(gf2o_mat *A, int64_t dst, int64_t src, int64_t col) {
  if (col >= A->ncols) return;
  gf2o_word *d = A->data + dst * A->rowstride;
  const gf2o_word *s = A->data + src * A->rowstride;
  int64_t w = col / 64;
  gf2o_word first = ~(gf2o_word)0 << (col % 64);
  for (; w < A->width; ++w) {
    gf2o_word v = s[w] & first;
    if (w == A->width - 1) v &= A->high_bitmask;
    d[w] ^= v;
    first = ~(gf2o_word)0;
  }
}

static void o_col_swap_in_row(gf2o_mat *A, int64_t r, int64_t a, int64_t b) {
  gf2o_word *row = A->data + r * A->rowstride;
  const gf2o_word x = ((row[a / 64] >> (a % 64)) ^ (row[b / 64] >> (b % 64))) & 1;
  row[a / 64] ^= x << (a % 64);
  row[b / 64] ^= x << (b % 64);
}

/* mzd_first_zero_row (mzd.c:1826-1841) of the window rows [row0, row0 + R) x columns [c0, c1), c0 on a word boundary:
 * one past the last row with a set bit.  (For a one-word window the reference ORs the word unmasked; every caller
 * here has zero excess bits or takes the base case either way, so the mask is always applied.) */
static int64_t o_first_zero_row(const gf2o_mat *A, int64_t row0, int64_t R, int64_t c0, int64_t c1) {
  const int64_t w0 = c0 / 64, w1 = (c1 + 63) / 64;
  const gf2o_word last = (c1 % 64) ? (~(gf2o_word)0 >> (64 - c1 % 64)) : ~(gf2o_word)0;
  for (int64_t i = R - 1; i >= 0; --i) {
    const gf2o_word *row = A->data + (row0 + i) * A->rowstride;
    gf2o_word any = 0;
    for (int64_t w = w0; w < w1; ++w) any |= (w == w1 - 1) ? (row[w] & last) : row[w];
    if (any) return i + 1;
  }
  return 0;
}

/* Columns [c0, c1) eliminated one by one -- what _mzd_ple_russian (ple_russian.c:380-617) computes on that window when
 * the columns before it are done: the t-th pivot found goes to Q[c0 + t]; *rank is the running row count. */
static int64_t o_ple_columns(gf2o_mat *A, int32_t *P, int32_t *Q, int64_t *rank, int64_t c0, int64_t c1) {
  const int64_t nrows = A->nrows, first = *rank;
  for (int64_t c = c0; c < c1 && *rank < nrows; ++c) {
    int64_t piv = -1;
    for (int64_t i = *rank; i < nrows; ++i)
      if (o_bit(A, i, c)) { piv = i; break; }            /* ple_russian.c:141-159: first row with the bit set */
    if (piv < 0) continue;
    P[*rank] = (int32_t)piv;                             /* :162-166 */
    Q[c0 + (*rank - first)] = (int32_t)c;
    o_row_swap(A, piv, *rank);
    for (int64_t r = *rank + 1; r < nrows; ++r)
      if (o_bit(A, r, c)) o_row_add_from(A, r, *rank, c + 1);  /* :150, :180-183: the multiplier stays at column c */
    ++*rank;
  }
  return *rank - first;
}

/* compressing L (ple_russian.c:596-602, ple.c:151): row r takes the column swaps (j, Q[j]) for j = 0 .. min(r, rank - 1) */
static void o_compress_l(gf2o_mat *A, const int32_t *Q, int64_t rank) {
  for (int64_t r = 0; r < A->nrows; ++r) {
    const int64_t last = r < rank - 1 ? r : rank - 1;
    for (int64_t j = 0; j <= last; ++j)
      if (Q[j] != j) o_col_swap_in_row(A, r, j, Q[j]);
  }
}

int32_t gf2o_ple(gf2o_mat *A, int32_t *P, int32_t *Q) {
  for (int64_t i = 0; i < A->nrows; ++i) P[i] = (int32_t)i;  /* ple_russian.c:412-414 */
  for (int64_t j = 0; j < A->ncols; ++j) Q[j] = (int32_t)j;
  int64_t rank = 0;
  o_ple_columns(A, P, Q, &rank, 0, A->ncols);
  o_compress_l(A, Q, rank);
  return (int32_t)rank;
}

/* The block recursion of _mzd_ple (ple.c:62-171) on the window rows [*rank, *rank + R) x columns [c0, c1).  The
 * decomposed matrix, P and the first `rank` entries of Q do not depend on it (they are fixed by the pivoting rule),
 * but the entries of Q BEHIND the rank do: a node resets its part of Q to the identity (:68), a window without a set
 * bit returns at once (:69), small windows take the column-by-column base case (:74-81), and the others split their
 * columns in halves (:96), recurse on the left half and on the Schur complement of the right half, and copy the right
 * half's pivots down behind the left half's (:146) -- the right half's own entries stay where they were. */
static int64_t o_ple_rec(gf2o_mat *A, int32_t *P, int32_t *Q, int64_t *rank, int64_t R, int64_t c0, int64_t c1, int64_t cutoff) {
  const int64_t ncols = c1 - c0, width = (ncols + 63) / 64;
  const int64_t e = o_first_zero_row(A, *rank, R, c0, c1);   /* :66 */
  for (int64_t c = c0; c < c1; ++c) Q[c] = (int32_t)c;        /* :68 */
  if (!e) return 0;                                           /* :69 */
  if (ncols <= 64 || width * R <= cutoff) return o_ple_columns(A, P, Q, rank, c0, c1);  /* :74-81 */
  const int64_t n1 = (((ncols - 1) / 64 + 1) >> 1) * 64;      /* :96 */
  const int64_t r1 = o_ple_rec(A, P, Q, rank, e, c0, c0 + n1, cutoff);       /* :105: A0 = the first e rows */
  const int64_t r2 = o_ple_rec(A, P, Q, rank, e - r1, c0 + n1, c1, cutoff);  /* :135: A11 = rows r1 .. e */
  for (int64_t t = 0; t < r2; ++t) Q[c0 + r1 + t] = Q[c0 + n1 + t];          /* :146 */
  return r1 + r2;
}

int32_t gf2o_ple_recursive(gf2o_mat *A, int32_t *P, int32_t *Q, int64_t cutoff) {
  for (int64_t i = 0; i < A->nrows; ++i) P[i] = (int32_t)i;
  int64_t rank = 0;
  o_ple_rec(A, P, Q, &rank, A->nrows, 0, A->ncols, cutoff);
  o_compress_l(A, Q, rank);
  return (int32_t)rank;
}

static void o_apply_q_tri(gf2o_mat *A, int64_t rows, const int32_t *Q) {
  for (int64_t row = 0; row < rows; ++row)
    for (int64_t i = row + 1; i < A->ncols; ++i)            /* mzp.c:285-288: rows [.., min(row_bound, i)) take swap i */
      if (Q[i] != i) o_col_swap_in_row(A, row, i, Q[i]);
}

/* ---- PLUQ (m4ri/ple.c:50-60, ple_russian.c:625-629, m4ri/mzp.c:279-293) ---------------------------------------- */
int32_t gf2o_pluq(gf2o_mat *A, int32_t *P, int32_t *Q) {     /* _mzd_pluq_russian */
  const int32_t r = gf2o_ple(A, P, Q);
  o_apply_q_tri(A, A->nrows, Q);
  return r;
}

int32_t gf2o_pluq_recursive(gf2o_mat *A, int32_t *P, int32_t *Q, int64_t cutoff) {  /* _mzd_pluq */
  const int32_t r = gf2o_ple_recursive(A, P, Q, cutoff);
  o_apply_q_tri(A, (r && r < A->nrows) ? r : A->nrows, Q);  /* ple.c:52-58: the window A0 of the first r rows, or all of A */
  return r;
}


/* ---- elimination table primitives (m4ri/brilliantrussian.c:163-211, :213-601) -------------------------- */
void gf2o_make_table(const gf2o_mat *M, int32_t r, int32_t c, int k, gf2o_mat *T, int32_t *L) {
  const int64_t home = c / 64, wide = M->width - home;
  const gf2o_word mask_end = M->high_bitmask, pure_begin = ~(gf2o_word)0 << (c % 64);
  const gf2o_word mask_begin = (wide != 1) ? pure_begin : (pure_begin & mask_end);  /* :165-168 */
  L[0] = 0;
  for (int32_t i = 1; i < (1 << k); ++i) {
    int b = 0;
    while (!((i >> b) & 1)) ++b;                 /* inc[i-1]: the bit in which Gray code i differs from i-1 */
    L[i ^ (i >> 1)] = i;                         /* ord[i] (graycode.c:31-62: reflected Gray code) */
    if (r + b >= M->nrows || wide <= 0) continue; /* :181 */
    gf2o_word *ti = T->data + (int64_t)i * T->rowstride + home;
    const gf2o_word *ti1 = T->data + (int64_t)(i - 1) * T->rowstride + home;
    const gf2o_word *m = M->data + (int64_t)(r + b) * M->rowstride + home;
    for (int64_t j = 0; j < wide; ++j) {
      gf2o_word x = m[j] ^ ti1[j];
      if (j == 0) x &= mask_begin;
      else if (j == wide - 1) x &= mask_end;
      ti[j] = x;
    }
  }
}

void gf2o_process_rows(gf2o_mat *M, int32_t startrow, int32_t stoprow, int32_t startcol, int k, int nt, const gf2o_mat *const *T,
                       const int32_t *const *L) {
  int kb[6];
  if (nt == 1) kb[0] = k;
  else if (nt == 2) { kb[0] = k / 2; kb[1] = k - k / 2; }   /* :357-358 */
  else {
    const int rem = k % nt;                                 /* :392-398, :438-445, :488-494, :544-552 */
    for (int i = 0; i < nt; ++i) kb[i] = k / nt + ((i < nt - 1 && rem >= nt - 1 - i) ? 1 : 0);
  }
  const int64_t block = startcol / 64;
  for (int64_t r = startrow; r < stoprow; ++r) {
    gf2o_word *row = M->data + r * M->rowstride;
    /* mzd_read_bits(M, r, startcol, k), mzd.h:892-901 */
    const int spot = startcol % 64, spill = spot + k - 64;
    gf2o_word bits = spill <= 0 ? (row[block] << -spill) : ((row[block + 1] << (64 - spill)) | (row[block] >> spill));
    bits >>= (64 - k);
    int32_t x[6];
    for (int t = 0; t < nt; ++t) {
      const gf2o_word bm = kb[t] >= 64 ? ~(gf2o_word)0 : (((gf2o_word)1 << kb[t]) - 1);
      x[t] = L[t][bits & bm];
      bits = kb[t] >= 64 ? 0 : bits >> kb[t];
    }
    for (int t = 0; t < nt; ++t) {
      const gf2o_word *tr = T[t]->data + (int64_t)x[t] * T[t]->rowstride;
      for (int64_t j = block; j < M->width; ++j) row[j] ^= tr[j];
    }
  }
}


/* ---- right-hand triangular solves (m4ri/triangular.c:41-130, :301-393): X T = B, column by column ------------ */
static void o_flip(gf2o_mat *M, int64_t r, int64_t c) { M->data[r * M->rowstride + c / 64] ^= (gf2o_word)1 << (c % 64); }

void gf2o_trsm_upper_right(const gf2o_mat *U, gf2o_mat *B) {  /* x_j = b_j + sum_{i<j} x_i U[i,j] */
  for (int64_t r = 0; r < B->nrows; ++r)
    for (int64_t j = 1; j < B->ncols; ++j) {
      int p = 0;
      for (int64_t i = 0; i < j; ++i) p ^= o_bit(B, r, i) & o_bit(U, i, j);
      if (p) o_flip(B, r, j);
    }
}

void gf2o_trsm_lower_right(const gf2o_mat *L, gf2o_mat *B) {  /* x_j = b_j + sum_{i>j} x_i L[i,j], j descending (triangular.c:366-393) */
  for (int64_t r = 0; r < B->nrows; ++r)
    for (int64_t j = (int64_t)B->ncols - 2; j >= 0; --j) {
      int p = 0;
      for (int64_t i = j + 1; i < B->ncols; ++i) p ^= o_bit(B, r, i) & o_bit(L, i, j);
      if (p) o_flip(B, r, j);
    }
}


/* ---- echelon forms (m4ri/echelonform.c:29-139, m4ri/brilliantrussian.c:603-841) --------------------------------
 * What mzd_echelonize_m4ri / mzd_echelonize_pluq / mzd_echelonize leave in A, stated without their schedules: columns
 * left to right, the pivot of a column is the first row at or below the current rank with the bit set
 * (_mzd_gauss_submatrix{,_full}, brilliantrussian.c:48-121; mzd_find_pivot, mzd.c:1661-1776; ple_russian.c:141-159), it is
 * swapped up to the rank's row, the rows below are cleared in that column -- and, for the reduced form (full), the rows
 * above as well.  The strips of 6k columns, the Gray-code tables and the PLUQ detour only change the order in which these
 * additions happen: each row ends as itself plus the one combination of pivot rows that clears its pivot columns. */
int32_t gf2o_echelonize(gf2o_mat *A, int full) {
  int64_t rank = 0;
  for (int64_t c = 0; c < A->ncols && rank < A->nrows; ++c) {
    int64_t piv = -1;
    for (int64_t i = rank; i < A->nrows; ++i)
      if (o_bit(A, i, c)) { piv = i; break; }
    if (piv < 0) continue;
    o_row_swap(A, piv, rank);
    for (int64_t r = full ? 0 : rank + 1; r < A->nrows; ++r)
      if (r != rank && o_bit(A, r, c)) o_row_add_from(A, r, rank, c);
    ++rank;
  }
  return (int32_t)rank;
}

/* mzd_apply_p_right / mzd_apply_p_right_trans (mzp.c:193-260): the column transpositions (i, P[i]) applied to every row,
 * i descending for A * P, ascending for A * P^T. */
void gf2o_apply_p_right(gf2o_mat *A, const int32_t *P, int64_t length, int trans) {
  if (length > A->ncols) length = A->ncols;
  for (int64_t r = 0; r < A->nrows; ++r)
    for (int64_t t = 0; t < length; ++t) {
      const int64_t i = trans ? t : length - 1 - t;
      if (P[i] != i) o_col_swap_in_row(A, r, i, P[i]);
    }
}


/* ---- the drivers over PLUQ (m4ri/mzp.c:65-81, m4ri/solve.c:30-191, m4ri/brilliantrussian.c:971-997) ------------------ */
void gf2o_apply_p_left(gf2o_mat *A, const int32_t *P, int64_t length, int trans) {  /* mzp.c:65-81 */
  if (A->ncols == 0) return;
  if (length > A->nrows) length = A->nrows;
  for (int64_t t = 0; t < length; ++t) {
    const int64_t i = trans ? length - 1 - t : t;
    o_row_swap(A, i, P[i]);
  }
}

static int o_is_zero(const gf2o_mat *M) {
  for (int64_t r = 0; r < M->nrows; ++r)
    for (int64_t w = 0; w < M->width; ++w)
      if (M->data[r * M->rowstride + w] & (w == M->width - 1 ? M->high_bitmask : ~(gf2o_word)0)) return 0;
  return 1;
}

static void o_zero_rows(gf2o_mat *M, int64_t r0, int64_t r1) {
  for (int64_t r = r0; r < r1; ++r)
    for (int64_t w = 0; w < M->width; ++w)
      M->data[r * M->rowstride + w] &= (w == M->width - 1) ? ~M->high_bitmask : 0;
}

int gf2o_pluq_solve_left(const gf2o_mat *A, int32_t rank, const int32_t *P, const int32_t *Q, gf2o_mat *B, int check) {  /* solve.c:57-121 */
  int retval = 0;
  gf2o_apply_p_left(B, P, A->nrows, 0);                                 /* :72 */
  gf2o_mat *Y1 = gf2o_init_window(B, 0, 0, rank, B->ncols);             /* :76-77 */
  gf2o_trsm_lower_left(A, Y1);                                          /* :78: only the bits (i, k), k < i < rank, of A are read */
  if (check) {                                                          /* :81-98 */
    gf2o_mat *H  = gf2o_init_window((gf2o_mat *)A, rank, 0, A->nrows, rank);
    gf2o_mat *Y2 = gf2o_init_window(B, rank, 0, A->nrows, B->ncols);
    if (A->nrows < B->nrows) o_zero_rows(B, A->nrows, B->nrows);
    if (A->nrows > rank && rank > 0 && B->ncols > 0) gf2o_addmul(Y2, H, Y1, 0);
    if (!o_is_zero(Y2)) retval = -1;
    gf2o_free(H);
    gf2o_free(Y2);
  }
  gf2o_trsm_upper_left(A, Y1);                                          /* :100 */
  gf2o_free(Y1);
  if (!check) o_zero_rows(B, rank, B->nrows);                           /* :104-114 */
  gf2o_apply_p_left(B, Q, A->ncols, 1);                                 /* :116 */
  return retval;
}

int gf2o_solve_left(gf2o_mat *A, gf2o_mat *B, int check) {                /* solve.c:123-152 */
  if (check && B->nrows > A->nrows) {
    gf2o_mat *Bpad = gf2o_init_window(B, A->nrows + 1 <= B->nrows ? A->nrows + 1 : B->nrows, 0, B->nrows, B->ncols);  /* :125: one row late */
    const int z = o_is_zero(Bpad);
    gf2o_free(Bpad);
    if (!z) return -1;
  }
  int32_t *P = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->nrows + 1)), *Q = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->ncols + 1));
  const int32_t rank = gf2o_pluq_recursive(A, P, Q, GF2O_PLE_CUTOFF);   /* :143 */
  const int r = gf2o_pluq_solve_left(A, rank, P, Q, B, check);
  free(P);
  free(Q);
  return r;
}

/* mzd_kernel_left_pluq (solve.c:154-191).  A <- its PLUQ; returns the rank; when it is below ncols, R (ncols x (ncols -
 * rank), zeroed by the caller) <- the kernel basis. */
int32_t gf2o_kernel_left_pluq(gf2o_mat *A, gf2o_mat *R) {
  int32_t *P = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->nrows + 1)), *Q = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->ncols + 1));
  const int32_t r = gf2o_pluq_recursive(A, P, Q, GF2O_PLE_CUTOFF);
  if (r < A->ncols && R) {
    gf2o_mat *RU = gf2o_init_window(R, 0, 0, r, R->ncols);
    for (int64_t i = 0; i < r; ++i)                                     /* :170-175 */
      for (int64_t j = 0; j < R->ncols; ++j)
        if (o_bit(A, i, r + j)) o_flip(R, i, j);
    gf2o_trsm_upper_left(A, RU);                                        /* :177 */
    gf2o_free(RU);
    for (int64_t i = 0; i < R->ncols; ++i) o_flip(R, r + i, i);         /* :179 */
    gf2o_apply_p_left(R, Q, A->ncols, 1);                               /* :180 */
  }
  free(P);
  free(Q);
  return r;
}

/* mzd_transpose (mzd.c:1118-1139): DST <- A^T, bit by bit (the reference's block schedule, mzd.c:700-1100, only decides
 * in which order the bits move).  DST's bits beyond its columns are kept. */
void gf2o_transpose(gf2o_mat *DST, const gf2o_mat *A) {
  if (DST->nrows != A->ncols || DST->ncols != A->nrows) die("mzd_transpose: Wrong size for return matrix.\n");
  for (int64_t i = 0; i < DST->nrows; ++i)
    for (int64_t w = 0; w < DST->width; ++w) {
      gf2o_word v = 0;
      const int64_t jn = (w == DST->width - 1 && DST->ncols % 64) ? DST->ncols % 64 : 64;
      for (int64_t j = 0; j < jn; ++j) v |= (gf2o_word)o_bit(A, w * 64 + j, i) << j;
      gf2o_word *d = DST->data + i * DST->rowstride + w;
      const gf2o_word m = (w == DST->width - 1) ? DST->high_bitmask : ~(gf2o_word)0;
      *d = (*d & ~m) | (v & m);
    }
}

/* mzd_trtri_upper / mzd_trtri_upper_russian (triangular.c:518-547, triangular_russian.c:378-470): A <- A^-1 in place for
 * a unit upper triangular A.  The reference's base step, _mzd_trtri_upper_submatrix (triangular_russian.c:378-382),
 * applied to the whole matrix: columns left to right, row i added from column i + 1 on to every row above it that has a
 * bit in column i.  The reference applies it inside k-bit blocks and reaches the rows above a block through tables,
 * and above 2 * L3 bits it halves and closes the off-diagonal block with two TRSMs; the inverse being unique, those
 * change the order of the row additions, not the result.  Diagonal and lower triangle: never read, never written. */
void gf2o_trtri_upper(gf2o_mat *A) {
  if (A->nrows != A->ncols) die("mzd_trtri_upper: matrix must be square.\n");
  for (int64_t i = 1; i < A->nrows; ++i)
    for (int64_t j = 0; j < i; ++j)
      if (o_bit(A, j, i)) o_row_add_from(A, j, i, i + 1);
}

/* mzd_inv_m4ri (brilliantrussian.c:971-997): the right block of the reduced row echelon form of [A | 0 | I] */
void gf2o_inv(gf2o_mat *B, const gf2o_mat *A) {
  const int64_t n = A->nrows, nr = 64 * A->width;
  gf2o_mat *C = gf2o_init((int32_t)n, (int32_t)(2 * nr));
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t w = 0; w < A->width; ++w)
      C->data[i * C->rowstride + w] = A->data[i * A->rowstride + w] & (w == A->width - 1 ? A->high_bitmask : ~(gf2o_word)0);
    o_flip(C, i, nr + i);
  }
  gf2o_echelonize(C, 1);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t w = 0; w < B->width; ++w) {
      const gf2o_word v = C->data[i * C->rowstride + A->width + w];
      gf2o_word *d = B->data + i * B->rowstride + w;
      if (w == B->width - 1) *d = (*d & ~B->high_bitmask) | (v & B->high_bitmask);
      else *d = v;
    }
  gf2o_free(C);
}
