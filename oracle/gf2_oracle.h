/* oracle/gf2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of M4RI's dense GF(2) multiply path
 * (mzd_mul -> Strassen-Winograd -> M4RM leaf).  It exists so the HIP path can be checked bit for
 * bit; nothing under m4ri_amd/ may include, link or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Pinning: tests/test_oracle_vs_reference.py runs every function here against the real reference
 * built from /root/reference into oracle/_ref/ (oracle/Makefile) on the shapes of the reference's
 * own tests/test_multiplication.c and tests/test_smallops.c, and tests/golden/ holds fixtures
 * generated from that reference build (tests/golden/make_golden.py).
 *
 * The matrix descriptor is layout-identical to the reference's mzd_t
 * (/root/reference m4ri/mzd.h:68-99: nrows@0, ncols@4, width@8, rowstride@16, flags@24,
 * high_bitmask@48, data@56, sizeof == 64) so one ctypes structure serves oracle, reference and
 * product alike.
 */
#ifndef GF2_ORACLE_H
#define GF2_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t gf2o_word;

typedef struct gf2o_mat {
  int32_t nrows;
  int32_t ncols;
  int64_t width;        /* words holding valid bits: ceil(ncols/64)            */
  int64_t rowstride;    /* words between rows                                  */
  uint8_t flags;        /* 0x2: ncols%64 != 0 ("non-zero excess"), 0x4: window */
  uint8_t pad[23];
  gf2o_word high_bitmask; /* valid bits of word width-1                        */
  gf2o_word *data;
} gf2o_mat;

#define GF2O_FLAG_EXCESS 0x2
#define GF2O_FLAG_WINDOW 0x4

/* allocation / views (mzd.c:142-185) */
gf2o_mat *gf2o_init(int32_t r, int32_t c);
gf2o_mat *gf2o_init_window(gf2o_mat *M, int32_t lowr, int32_t lowc, int32_t highr, int32_t highc);
void gf2o_free(gf2o_mat *A);

/* fill: exactly `width` PRNG words per row, row-major, last one masked into the row
 * (mzd.c:1282-1292 mzd_randomize_custom); PRNG = splitmix64 seeded with `seed`. */
void gf2o_fill_splitmix(gf2o_mat *A, uint64_t seed);
uint64_t gf2o_splitmix_next(uint64_t *state);

/* element-wise helpers: mzd.c:1471 (_mzd_add), :1363 (mzd_copy), :1294 (mzd_set_ui(.,0)),
 * :1314 (mzd_equal: valid bits only) */
gf2o_mat *gf2o_add(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B);
gf2o_mat *gf2o_copy(gf2o_mat *N, const gf2o_mat *P);
void gf2o_set_zero(gf2o_mat *A);
int gf2o_equal(const gf2o_mat *A, const gf2o_mat *B);

/* products.  `clear` != 0: C = A*B, else C ^= A*B.  All return C. */
gf2o_mat *gf2o_mul_naive(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int clear);       /* mzd.c:1256-1268 */
gf2o_mat *gf2o_mul_m4rm(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int k, int clear);  /* brilliantrussian.c:1032-1190 */
gf2o_mat *gf2o_mul_even(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff);        /* strassen.c:41-208 */
gf2o_mat *gf2o_addmul_even(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff);     /* strassen.c:367-526 */
gf2o_mat *gf2o_mul(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff);             /* strassen.c:345-365 */
gf2o_mat *gf2o_addmul(gf2o_mat *C, const gf2o_mat *A, const gf2o_mat *B, int cutoff);          /* strassen.c:675-700 */

/* triangular solves, B <- T^-1 B in place, T unit triangular (diagonal and other triangle never read):
 * plain substitution, the base case of m4ri/triangular.c:406-455 (lower) and :467-514 (upper) applied to
 * the whole matrix -- the solution is unique, so this equals the reference's recursive / Russian schedules */
void gf2o_trsm_lower_left(const gf2o_mat *L, gf2o_mat *B);
void gf2o_trsm_upper_left(const gf2o_mat *U, gf2o_mat *B);
/* right-hand forms, B <- B T^-1 (m4ri/triangular.c:41-130, :301-393): column substitution */
void gf2o_trsm_upper_right(const gf2o_mat *U, gf2o_mat *B);
void gf2o_trsm_lower_right(const gf2o_mat *L, gf2o_mat *B);

/* PLE decomposition in place, the semantics of _mzd_ple_russian (m4ri/ple_russian.c:380-617): columns left
 * to right, pivot = the first row at or below the current rank position with a set bit in the column after
 * elimination by the earlier pivots, that row swapped up, rows below cleared from the NEXT column on (the
 * multipliers stay in the pivot column), then L compressed to the left (ple_russian.c:596-602).  P (nrows
 * entries) and Q (ncols entries) receive the row / column transpositions in the reference's LAPACK-style
 * convention.  Returns the rank.  The reference's blocks, tables and lazy updates (ple_russian.c:119-196)
 * change when row operations happen, never their outcome. */
int32_t gf2o_ple(gf2o_mat *A, int32_t *P, int32_t *Q);

/* _mzd_ple (m4ri/ple.c:62-171): the same decomposition through the reference's column-halving recursion, which
 * leaves its own values in Q behind the rank (see gf2_oracle.c).  cutoff: __M4RI_PLE_CUTOFF (ple.h:40), in words. */
#define GF2O_PLE_CUTOFF 524288
int32_t gf2o_ple_recursive(gf2o_mat *A, int32_t *P, int32_t *Q, int64_t cutoff);

/* PLUQ decomposition in place: the PLE, then the column transpositions of Q applied to the rows of U above their
 * pivots -- mzd_apply_p_right_trans_tri (m4ri/mzp.c:279-293): row r takes the swaps (i, Q[i]) for i = r+1 .. ncols-1 in
 * ascending order.  gf2o_pluq = _mzd_pluq_russian (ple_russian.c:625-629), gf2o_pluq_recursive = _mzd_pluq
 * (ple.c:50-60: on the recursive PLE, and only the first `rank` rows when 0 < rank < nrows). */
int32_t gf2o_pluq(gf2o_mat *A, int32_t *P, int32_t *Q);
int32_t gf2o_pluq_recursive(gf2o_mat *A, int32_t *P, int32_t *Q, int64_t cutoff);

/* Row echelon form (full = 0) / reduced row echelon form (full = 1) in place, returns the rank: what
 * mzd_echelonize_m4ri, mzd_echelonize_pluq and mzd_echelonize (m4ri/echelonform.c:29-139, brilliantrussian.c:603-841)
 * leave in A -- see gf2_oracle.c for why the three agree. */
int32_t gf2o_echelonize(gf2o_mat *A, int full);
/* mzd_apply_p_right (trans = 0) / mzd_apply_p_right_trans (trans = 1), m4ri/mzp.c:193-260 */
void gf2o_apply_p_right(gf2o_mat *A, const int32_t *P, int64_t length, int trans);

/* The drivers over PLUQ: row transpositions (m4ri/mzp.c:65-81), linear systems (m4ri/solve.c:30-152: A <- its PLUQ,
 * B <- a solution with the undefined rows zero; -1 = inconsistent), the kernel basis (solve.c:154-191) and the inverse
 * (m4ri/brilliantrussian.c:971-997). */
void gf2o_apply_p_left(gf2o_mat *A, const int32_t *P, int64_t length, int trans);
int gf2o_pluq_solve_left(const gf2o_mat *A, int32_t rank, const int32_t *P, const int32_t *Q, gf2o_mat *B, int check);
int gf2o_solve_left(gf2o_mat *A, gf2o_mat *B, int check);
int32_t gf2o_kernel_left_pluq(gf2o_mat *A, gf2o_mat *R);
void gf2o_inv(gf2o_mat *B, const gf2o_mat *A);

/* mzd_transpose (m4ri/mzd.c:1118-1139) and the in-place inverse of a unit upper triangular matrix, mzd_trtri_upper /
 * mzd_trtri_upper_russian (m4ri/triangular.c:518-547, triangular_russian.c:378-470). */
void gf2o_transpose(gf2o_mat *DST, const gf2o_mat *A);
void gf2o_trtri_upper(gf2o_mat *A);

/* table primitives of the elimination routines: m4ri/brilliantrussian.c:163-211 (mzd_make_table: the Gray-code
 * chain T[i] = T[i-1] ^ M[r + inc[i-1]], first word masked below column c, last by the column mask, L[ord[i]] = i,
 * steps whose row does not exist skipped) and :213-601 (mzd_process_rows, 2..6: nt tables, the k-bit strip cut as
 * the reference cuts it, rows XORed from word startcol/64 on). */
void gf2o_make_table(const gf2o_mat *M, int32_t r, int32_t c, int k, gf2o_mat *T, int32_t *L);
void gf2o_process_rows(gf2o_mat *M, int32_t startrow, int32_t stoprow, int32_t startcol, int k, int nt, const gf2o_mat *const *T,
                       const int32_t *const *L);

/* FNV-1a over the valid bits, row-major, excess masked: a size-independent fingerprint used for
 * the large fixtures (the reference's own mzd_hash is unusable: debug_dump.h:35 shifts by data). */
uint64_t gf2o_fingerprint(const gf2o_mat *A);

#ifdef __cplusplus
}
#endif
#endif
