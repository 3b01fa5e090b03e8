/* include/m4ri_amd.h -- C ABI of libm4ri_amd.so, the MI355X-native drop-in for M4RI's dense GF(2)
 * multiply path (mzd_mul -> Strassen-Winograd -> M4RM).  Plain pointers and sizes only.
 *
 * Part 1 is the drop-in boundary: the exact symbols (names, signatures, semantics, fatal-error
 * behaviour) a libm4ri user of this path binds, each citing the reference declaration it replaces.
 * Part 2 is the device-resident API those entry points are built from; hosts that keep matrices
 * in HBM across calls (bench.py, the multi-GPU driver, chains of products) call it directly.
 *
 * All file:line citations are relative to the reference tree (malb/m4ri @ 20251207).
 */
#ifndef M4RI_AMD_H
#define M4RI_AMD_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types: m4ri/misc.h:72,81,87 -------------------------------------------------------------- */
typedef int rci_t;      /* row/column index */
typedef int64_t wi_t;   /* word index       */
typedef uint64_t word;  /* 64 columns, LSB = lowest column */

/* mzd_t: layout-identical to m4ri/mzd.h:68-99 (sizeof == 64; nrows@0 ncols@4 width@8 rowstride@16
 * flags@24 high_bitmask@48 data@56).  Bit (r,c) = (data[r*rowstride + c/64] >> (c%64)) & 1.
 * A program that already includes <m4ri/m4ri.h> must NOT include this header's struct: define
 * M4RI_AMD_NO_MZD_T first and use M4RI's own declaration -- the two are the same bytes. */
#ifndef M4RI_AMD_NO_MZD_T
typedef struct mzd_t {
  rci_t nrows;
  rci_t ncols;
  wi_t width;
  wi_t rowstride;
  uint8_t flags;        /* 0x2 non-zero excess, 0x4 windowed (mzd.h:144,150) */
  uint8_t padding[23];
  word high_bitmask;
  word *data;
} mzd_t;
#endif

/* mzp_t: layout-identical to m4ri/mzp.h:37-49 (LAPACK-style transpositions: values[i] is the index swapped
 * with i, for i ascending). */
#ifndef M4RI_AMD_NO_MZD_T
typedef struct mzp_t {
  rci_t *values;
  rci_t length;
} mzp_t;
#endif

/* =================================================================================================
 * Part 1 -- drop-in entry points (host mzd_t in, host mzd_t out, blocking)
 *
 * Common contract: C == NULL => C is allocated with the host program's mzd_init
 * (resolved with dlsym; see m4ri_amd_mzd_init below when no libm4ri is loaded) and returned;
 * otherwise C must be A->nrows x B->ncols.  A, B, C may be windows; words at index >= width and
 * bits outside high_bitmask of a windowed C are never written; for a non-window C they end up 0.
 * `cutoff` and `k` are performance hints: every value yields the same bits.  A == B is legal.
 * Dimension mismatch or cutoff < 0 prints to stderr and abort()s, like m4ri_die (misc.c:36-42);
 * so does any HIP failure -- there is no CPU fallback.
 * ============================================================================================== */

/* C = A*B.  Replaces mzd_mul, m4ri/strassen.h:52 (strassen.c:345-365). */
mzd_t *mzd_mul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
/* C += A*B.  Replaces mzd_addmul, m4ri/strassen.h:68 (strassen.c:675-700). */
mzd_t *mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
/* Unchecked variants L4 calls directly.  m4ri/strassen.h:88, :109, :126. */
mzd_t *_mzd_mul_even(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
mzd_t *_mzd_addmul_even(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
mzd_t *_mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
/* Squaring entry points mzd_mul/_mzd_addmul dispatch to when A == B; exported by libm4ri although
 * not declared in strassen.h (strassen.c:210, :528). */
mzd_t *_mzd_sqr_even(mzd_t *C, mzd_t const *A, int cutoff);
mzd_t *_mzd_addsqr_even(mzd_t *C, mzd_t const *A, int cutoff);
/* M4RM leaf only (no Strassen).  m4ri/brilliantrussian.h:274, :291, :317. */
mzd_t *mzd_mul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k);
mzd_t *mzd_addmul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k);
mzd_t *_mzd_mul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k, int clear);
/* OpenMP block-parallel entry points, m4ri/mp.h:47, :62 (mp.c:277-324): same results.  With several
 * devices configured (part 4: every visible GPU by default) and min(m, l, n) at or above the multi-device
 * threshold, the sub-products of the top Strassen-Winograd level(s) are spread over the devices; otherwise
 * -- one GPU, small products, pinned operands -- they run the single-GPU schedule (where the GPU grid
 * replaces the omp sections). */
mzd_t *mzd_mul_mp(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);
mzd_t *mzd_addmul_mp(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff);

/* ---- triangular solves with a matrix right-hand side (the callers of the multiply
 * path one step up; TRSM base cases are where an LD_PRELOADed PLE / solve spends its CPU time) ------------
 * B <- L^-1 B / B <- U^-1 B in place, T unit triangular: its diagonal and its other triangle are never read
 * (m4ri/triangular.c:406-455, :467-514), B and T may be windows, the bits of B's last word outside its
 * columns are kept.  The solution is unique, so every schedule gives the reference's bits; `k` and `cutoff`
 * are hints.  m4ri/triangular.h:115, :127, :142, :153; m4ri/triangular_russian.h:43, :55. */
void mzd_trsm_lower_left(mzd_t const *L, mzd_t *B, const int cutoff);
void _mzd_trsm_lower_left(mzd_t const *L, mzd_t *B, const int cutoff);
void _mzd_trsm_lower_left_russian(mzd_t const *L, mzd_t *B, int k);
void mzd_trsm_upper_left(mzd_t const *U, mzd_t *B, const int cutoff);
void _mzd_trsm_upper_left(mzd_t const *U, mzd_t *B, const int cutoff);
void _mzd_trsm_upper_left_russian(mzd_t const *U, mzd_t *B, int k);
/* The right-hand forms, B <- B U^-1 / B <- B L^-1 (X T = B): m4ri/triangular.h:50, :64, :82, :100 (triangular.c:41-130,
 * :301-393).  Same rules: unique solution, diagonal and other triangle never read, windows allowed. */
void mzd_trsm_upper_right(mzd_t const *U, mzd_t *B, const int cutoff);
void _mzd_trsm_upper_right(mzd_t const *U, mzd_t *B, const int cutoff);
void mzd_trsm_lower_right(mzd_t const *L, mzd_t *B, const int cutoff);
void _mzd_trsm_lower_right(mzd_t const *L, mzd_t *B, const int cutoff);

/* ---- PLE decomposition -----------------------------------------------------------
 * A = P L E Q in place: returns the rank r; afterwards the first r columns of A hold L below the diagonal (unit
 * diagonal implied), E sits in the rows 0..r-1 from each row's pivot column on, P (A->nrows entries) and Q
 * (A->ncols entries) hold the row / column transpositions -- exactly the reference's output, which is fixed by
 * its pivoting rule (first row with a set bit, columns left to right), not by its schedule -- except the entries
 * of Q behind the rank, which mzd_ple / _mzd_ple fill by their recursion on column halves and _mzd_ple_russian
 * leaves as the identity; both are reproduced (M4RI_AMD_PLE_CUTOFF below).  `cutoff`, `k`: hints.  m4ri/ple.h:103, :137; m4ri/ple_russian.h:81 (ple.c:33-171, ple_russian.c:380-617). */
rci_t mzd_ple(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff);
rci_t _mzd_ple(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff);
rci_t _mzd_ple_russian(mzd_t *A, mzp_t *P, mzp_t *Q, int k);
/* A = P L U Q in place (m4ri/ple.h:70, :120; ple_russian.h:98; ple.c:41-60): the PLE above, then the columns of the
 * first r rows permuted so that U is upper triangular with its pivots on the diagonal
 * (mzd_apply_p_right_trans_tri, m4ri/mzp.h:202, mzp.c:279-293, exported as well). */
rci_t mzd_pluq(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff);
rci_t _mzd_pluq(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff);
rci_t _mzd_pluq_russian(mzd_t *A, mzp_t *P, mzp_t *Q, int k);
void mzd_apply_p_right_trans_tri(mzd_t *A, mzp_t const *Q);

/* ---- echelon forms (the drivers over the elimination kernels) ---------------------------
 * Row echelon form (full == 0) or reduced row echelon form (full != 0) of A in place; returns the rank.  The three
 * drivers of the reference (Four-Russians strips, PLE/PLUQ based, density heuristic) leave the same matrix -- their
 * common pivoting rule fixes it -- so all of them map to one device routine (echelon.hip); `k`, `heuristic` and
 * `threshold` are hints.  m4ri/echelonform.h:50, :63, :79; m4ri/brilliantrussian.h:215 (echelonform.c:29-139,
 * brilliantrussian.c:603-841). */
rci_t mzd_echelonize(mzd_t *A, int full);
rci_t mzd_echelonize_m4ri(mzd_t *A, int full, int k);
rci_t mzd_echelonize_pluq(mzd_t *A, int full);
rci_t mzd_echelonize_naive(mzd_t *A, int full); /* m4ri/mzd.h:740 (mzd.c:208-233): plain Gaussian elimination -- same pivoting rule, same result */
rci_t _mzd_echelonize_m4ri(mzd_t *A, const int full, int k, int heuristic, const double threshold);
/* A <- A * P / A * P^T: the column transpositions (i, P[i]) on every row, i descending / ascending.
 * m4ri/mzp.h:142, :153 (mzp.c:193-260). */
void mzd_apply_p_right(mzd_t *A, mzp_t const *P);
void mzd_apply_p_right_trans(mzd_t *A, mzp_t const *P);

/* ---- linear systems, kernels, inverses (the drivers over PLUQ; solve.hip) -------------------------------------
 * mzd_solve_left (m4ri/solve.h:50, :123; solve.c:30-152): A X = B.  A (m x n) is left holding its PLUQ decomposition, B
 * (max(m, n) rows) the solution in its first n rows with the undefined rows zero; returns 0, or -1 when
 * inconsistency_check finds no solution.  mzd_pluq_solve_left (solve.h:76, :103): the same from a decomposition.
 * mzd_kernel_left_pluq (solve.h:140; solve.c:154-191): A <- its PLUQ; returns a new n x (n - rank) matrix whose columns
 * span {x : A x = 0}, or NULL when the rank is n.  mzd_inv_m4ri (m4ri/brilliantrussian.h:256; .c:971-997): B <- A^-1
 * through the reduced echelon form of [A | I] (B == NULL: a new matrix).  mzd_apply_p_left{,_trans} (m4ri/mzp.h:120,
 * :131; mzp.c:65-81): the row transpositions (i, P[i]) ascending / descending. */
int mzd_solve_left(mzd_t *A, mzd_t *B, int const cutoff, int const inconsistency_check);
int _mzd_solve_left(mzd_t *A, mzd_t *B, int const cutoff, int const inconsistency_check);
int mzd_pluq_solve_left(mzd_t const *A, rci_t rank, mzp_t const *P, mzp_t const *Q, mzd_t *B, int const cutoff, int const inconsistency_check);
int _mzd_pluq_solve_left(mzd_t const *A, rci_t rank, mzp_t const *P, mzp_t const *Q, mzd_t *B, int const cutoff, int const inconsistency_check);
mzd_t *mzd_kernel_left_pluq(mzd_t *A, int const cutoff);
mzd_t *mzd_inv_m4ri(mzd_t *B, mzd_t const *A, int k);
void mzd_apply_p_left(mzd_t *A, mzp_t const *P);
void mzd_apply_p_left_trans(mzd_t *A, mzp_t const *P);

/* ---- transposes and triangular inverses (transpose.hip, trsm.hip) ------------------------------------------------
 * mzd_transpose (m4ri/mzd.h:611; mzd.c:1118-1139): DST <- A^T (DST == NULL: a new matrix; wrong dimensions die with the
 * reference's message).  DST may be A itself here (the reference forbids it).
 * mzd_trtri_upper (m4ri/triangular.h:163; triangular.c:518-547) and mzd_trtri_upper_russian (m4ri/triangular_russian.h:66;
 * .c:384-470): A <- A^-1 in place for a unit upper triangular A.  As in the reference the diagonal and the lower
 * triangle are not written; unlike the reference they are not read either: a zero on the diagonal (a singular input,
 * outside the reference's contract) makes the reference's result depend on its blocking, while this library returns
 * the inverse of the unit-diagonal matrix. */
mzd_t *mzd_transpose(mzd_t *DST, mzd_t const *A);
mzd_t *mzd_trtri_upper(mzd_t *A);
mzd_t *mzd_trtri_upper_russian(mzd_t *A, int k);

/* ---- the table primitives of M4RI's elimination routines -----------------------------
 * mzd_make_table (m4ri/brilliantrussian.h:56, .c:163-211): T[i], i = 1 .. 2^k - 1, = the Gray-code combinations of
 * rows r .. r+k-1 of M from word c/64 on (first word masked below column c, last word by M's column mask), and
 * L[ord[i]] = i.  mzd_process_rows{,2..6} (brilliantrussian.h:74-190, .c:213-601): for every row in
 * [startrow, endrow) the k-bit strip at column startcol is cut into N groups, group t looked up in L_t, and
 * T_t's row XORed onto the row from word startcol/64 on.  Same bits as the reference for tables made by
 * mzd_make_table (the k == 1 shortcut of mzd_process_rows, .c:220-296, reads T's row 1 without L). */
void mzd_make_table(mzd_t const *M, rci_t r, rci_t c, int k, mzd_t *T, rci_t *L);
void mzd_process_rows(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T, rci_t const *L);
void mzd_process_rows2(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1);
void mzd_process_rows3(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2);
void mzd_process_rows4(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2, mzd_t const *T3, rci_t const *L3);
void mzd_process_rows5(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2, mzd_t const *T3, rci_t const *L3, mzd_t const *T4,
                       rci_t const *L4);
void mzd_process_rows6(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2, mzd_t const *T3, rci_t const *L3, mzd_t const *T4,
                       rci_t const *L4, mzd_t const *T5, rci_t const *L5);

/* ---- matrix I/O formats (pure host code, m4ri/io.h:44-193, io.c:49-357) -----------------
 * Text rows "[1 1 :...|...]" (64-bit groups, ':' every four entries); a row-major string of '0'/'1'; the JCF sparse
 * text format ("m n 2", number of non-zeros, then signed 1-based column indices, a negative index starts the next
 * row); 1-bit grayscale PNG, one pixel per entry, black = 1 (written and parsed directly on zlib: libpng is what the
 * reference uses, the files are the same).  Matrices returned are allocated like C == NULL results. */
void mzd_fprint_row(FILE *stream, mzd_t const *M, const rci_t i);
void mzd_fprint(FILE *stream, mzd_t const *M);
void mzd_print(mzd_t const *M);
mzd_t *mzd_from_str(rci_t m, rci_t n, const char *str);
mzd_t *mzd_from_jcf(const char *fn, int verbose);
mzd_t *mzd_from_png(const char *fn, int verbose);
int mzd_to_png(const mzd_t *A, const char *fn, int compression_level, const char *comment, int verbose);

/* Allocation used for C == NULL when the process has no libm4ri (standalone use, our tests):
 * same layout rules as mzd_init/mzd_free (mzd.c:142-157,179-185). */
mzd_t *m4ri_amd_mzd_init(rci_t r, rci_t c);
void m4ri_amd_mzd_free(mzd_t *A);
/* Frees a C == NULL result with the allocator it came from (the host program's mzd_free when its
 * mzd_init produced it, m4ri_amd_mzd_free otherwise) -- for bindings that cannot call mzd_free. */
void m4ri_amd_result_free(mzd_t *A);

/* =================================================================================================
 * Part 2 -- device-resident API.  Matrices live in HBM in the same bit-packed row-major layout:
 * `data` is a device pointer to word (0,0), `stride` the distance between rows in words.
 * Invariant the caller keeps: bits at column >= ncols inside the last word of a row are zero
 * (m4ri_amd_fill_dev and every product below preserve it; m4ri_amd_mask_tail_dev establishes it).
 * All calls are asynchronous on `stream` (a hipStream_t; NULL = the null stream) unless noted and
 * return 0 on success or a hipError_t value.
 * ============================================================================================== */

/* Bind the calling thread's HIP device for later calls and create the engine (idempotent). */
int m4ri_amd_init(int device);
int m4ri_amd_device_count(void);

/* C (+)= A*B on the device: Strassen-Winograd levels over batched M4RM leaves.
 * C: m x n, A: m x l, B: l x n.  add != 0 accumulates.  cutoff: 0 = engine default, otherwise the
 * reference's meaning (recurse until 3*dim < 4*cutoff for some dim, strassen.c:39,51).
 * Asynchronous on `stream`.  The engine has ONE workspace per device, so a product issued on another
 * stream than the previous one first waits (on the device) for that one to finish; several host
 * threads may call concurrently (the host side is serialised). */
int m4ri_amd_mul_dev(word *C, int64_t c_stride, const word *A, int64_t a_stride, const word *B,
                     int64_t b_stride, int64_t m, int64_t l, int64_t n, int add, int cutoff,
                     void *stream);
/* One M4RM leaf launch, no Strassen (the device twin of _mzd_mul_m4rm).  ksplit: 0 = automatic. */
int m4ri_amd_m4rm_dev(word *C, int64_t c_stride, const word *A, int64_t a_stride, const word *B,
                      int64_t b_stride, int64_t m, int64_t l, int64_t n, int add, int ksplit,
                      void *stream);
/* `batch` products of one shape in ONE launch: C_b (+)= A_b * B_b, X_b = X + b * x_bs (word offsets).  Plain M4RM leaves
 * (no Strassen levels): for the many small equal products of a blocked algorithm. */
int m4ri_amd_m4rm_batch_dev(word *C, int64_t c_stride, int64_t c_bs, const word *A, int64_t a_stride, int64_t a_bs, const word *B,
                            int64_t b_stride, int64_t b_bs, int64_t m, int64_t l, int64_t n, int64_t batch, int add, void *stream);
/* `batch` independent products of one shape WITH the Strassen-Winograd levels of m4ri_amd_mul_dev, every pass and the leaf launch
 * shared by the whole batch (what the ranks of a sharded Strassen level use for the several sub-products each of them owns,
 * multi.hip; the sub-products of strassen.c:111-150 are independent).  Bit-identical to `batch` calls of m4ri_amd_mul_dev. */
int m4ri_amd_mul_batch_dev(word *C, int64_t c_stride, int64_t c_bs, const word *A, int64_t a_stride, int64_t a_bs, const word *B,
                           int64_t b_stride, int64_t b_bs, int64_t m, int64_t l, int64_t n, int64_t batch, int add, int cutoff, void *stream);
/* the time model's estimate for `batch` products m x l x n scheduled as one at `levels` levels (pure host arithmetic) */
double m4ri_amd_model_seconds_batch(int64_t m, int64_t l, int64_t n, int levels, int64_t batch);
/* C = A ^ B on rows x ncols bits: the device twin of _mzd_add (mzd.c:1471-1583).  In-place allowed,
 * operands may have different strides; the last word of every row is written under the column mask
 * and the other bits of C's last word are kept (mzd.c:1489). */
int m4ri_amd_xor_dev(word *C, int64_t c_stride, const word *A, int64_t a_stride, const word *B,
                     int64_t b_stride, int64_t rows, int64_t ncols, void *stream);
/* B (mb x nb, device) <- L^-1 B / U^-1 B in place, T (mb x mb, device) unit triangular -- the device twins of
 * _mzd_trsm_lower_left / _mzd_trsm_upper_left: halves recursion down to 64-row blocks, every update one
 * m4ri_amd_mul_dev on views.  Bits of B at column >= nb: zero in, zero out. */
int m4ri_amd_trsm_lower_left_dev(const word *L, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                 void *stream);
int m4ri_amd_trsm_upper_left_dev(const word *U, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                 void *stream);
/* B (mb x nb) <- B U^-1 / B L^-1, T (nb x nb, device) unit triangular: the right-hand twins. */
int m4ri_amd_trsm_upper_right_dev(const word *U, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                  void *stream);
int m4ri_amd_trsm_lower_right_dev(const word *L, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                  void *stream);
/* Device twins of mzd_process_rowsN / mzd_make_table (elim.hip): streaming, HBM-bound.  kbits[t]: bits of group
 * t (lowest first); L[t]: 2^kbits[t] table row numbers; idx_scratch: 6 * (stoprow - startrow) int32; jstar: see
 * elim.hip (all zeros when rows r .. r+k-1 exist).  Asynchronous on `stream`. */
int m4ri_amd_process_rows_dev(word *M, int64_t stride, int64_t width, int64_t startrow, int64_t stoprow, int64_t startcol, int ntables,
                              const int32_t *kbits, const word *const *T, const int64_t *t_stride, const int32_t *const *L,
                              int32_t *idx_scratch, void *stream);
int m4ri_amd_make_table_dev(const word *M, int64_t m_stride, int64_t m_rows, int64_t ncols, int64_t r, int64_t c, int k, const word *Tin,
                            word *Tout, int64_t t_stride, const int32_t *jstar, void *stream);
/* PLE of a device matrix in place (bits at column >= ncols zero in, zero out).  P (nrows entries) and Q (ncols
 * entries) are HOST arrays.  Blocking: the pivots of every 64-column block are read back.
 * recursion_cutoff: 0 gives _mzd_ple_russian's Q (the pivot columns, then the identity); M4RI_AMD_PLE_CUTOFF gives
 * _mzd_ple's, whose recursion on column halves (m4ri/ple.c:62-171) leaves other transpositions behind the rank --
 * which ones depends on the recursion's shape, i.e. on __M4RI_PLE_CUTOFF of the reference build (m4ri/ple.h:40:
 * 524288 words for any L3 cache of 4 MiB or more).  Matrix, P, rank and pivots are the same either way. */
#define M4RI_AMD_PLE_CUTOFF 524288
int m4ri_amd_ple_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, int32_t *P, int32_t *Q, int32_t *rank_out,
                     int64_t recursion_cutoff, void *stream);
/* PLUQ in place (m4ri/ple.c:50-60): the PLE, then the column step below on the first `rank` rows.  Blocking. */
int m4ri_amd_pluq_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, int32_t *P, int32_t *Q, int32_t *rank_out,
                      int64_t recursion_cutoff, void *stream);
/* Device twins of the drivers over PLUQ (solve.hip); P, Q: HOST arrays; *retval: 0 / -1 as mzd_solve_left.  Blocking. */
int m4ri_amd_apply_p_left_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, const int32_t *P, int64_t length, int trans, void *stream);
int m4ri_amd_pluq_solve_left_dev(const word *A, int64_t a_stride, int64_t m, int64_t n, int32_t rank, const int32_t *P, const int32_t *Q, word *B,
                                 int64_t b_stride, int64_t b_rows, int64_t b_cols, int cutoff, int inconsistency_check, int *retval, void *stream);
int m4ri_amd_solve_left_dev(word *A, int64_t a_stride, int64_t m, int64_t n, word *B, int64_t b_stride, int64_t b_rows, int64_t b_cols, int cutoff,
                            int inconsistency_check, int *retval, void *stream);
int m4ri_amd_kernel_left_pluq_dev(word *A, int64_t a_stride, int64_t m, int64_t n, word *R, int64_t r_stride, int cutoff, int32_t *rank_out,
                                  void *stream);
int m4ri_amd_inv_dev(word *Binv, int64_t b_stride, const word *A, int64_t a_stride, int64_t n, void *stream);
/* Device twins of mzd_transpose and mzd_trtri_upper.  m4ri_amd_transpose_dev: D (ncols x nrows) <- A^T, D must not
 * overlap A; one HBM-bound launch (A read once, D written once, whole 128-byte lines on both sides); asynchronous.
 * m4ri_amd_trtri_upper_dev: U (n x n) <- U^-1, only the bits strictly above the diagonal are read and written;
 * asynchronous on `stream`, scratch grow-only per device. */
int m4ri_amd_transpose_dev(word *D, int64_t d_stride, const word *A, int64_t a_stride, int64_t nrows, int64_t ncols, void *stream);
int m4ri_amd_trtri_upper_dev(word *U, int64_t stride, int64_t n, void *stream);
/* Device twins of mzd_echelonize* and mzd_apply_p_right{,_trans} (echelon.hip).  P: HOST array.  Blocking. */
int m4ri_amd_echelonize_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, int full, int32_t *rank_out, void *stream);
int m4ri_amd_apply_p_right_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, const int32_t *P, int64_t length, int trans, void *stream);
/* Row r <- its columns under the transpositions (i, Q[i]), i = r+1 .. ncols-1 ascending (mzd_apply_p_right_trans_tri,
 * m4ri/mzp.c:279-293).  Q: HOST array, ncols entries, Q[i] >= i.  Blocking. */
int m4ri_amd_apply_p_right_trans_tri_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, const int32_t *Q, void *stream);
/* Deterministic fill: word (r, j) = splitmix64 stream `seed`, output number r*width + j, last word
 * masked -- the order mzd_randomize_custom fills a matrix in (mzd.c:1282-1292). */
int m4ri_amd_fill_dev(word *M, int64_t stride, int64_t rows, int64_t ncols, uint64_t seed, void *stream);
int m4ri_amd_mask_tail_dev(word *M, int64_t stride, int64_t rows, int64_t ncols, void *stream);

/* Schedule statistics of the most recent m4ri_amd_mul_dev / m4ri_amd_m4rm_dev on this thread's
 * engine.  With profiling on, every leaf launch is bracketed by HIP events on its own stream and
 * leaf_ms is their sum (reading it synchronises the stream). */
typedef struct m4ri_amd_stats {
  int32_t levels;           /* Strassen-Winograd levels used                       */
  int32_t leaf_launches;    /* kernel launches of the M4RM leaf                     */
  int64_t leaf_products;    /* products those launches computed (batch members)     */
  int32_t leaf_m, leaf_l, leaf_n; /* shape of the batched leaves                     */
  int32_t leaf_gen;         /* generation of the leaf kernel used: 1 m4rm_leaf,
                               4 m4rm8q, 5 m4rm_small (small products, one launch)  */
  double leaf_ms;           /* sum of leaf launch durations (profiling on), else 0  */
  double leaf_bytes;        /* algorithmic bytes of the leaf launches:
                               8*(m*W(l) + l*W(n) + m*W(n)) per product             */
  double aux_bytes;         /* declared bytes of the fused down/up passes           */
  double workspace_bytes;   /* HBM held by the engine's workspace                   */
  double cum_leaf_ms;       /* cumulative profiling (mode 2): leaf time of ALL products since it was
                               switched on, and how many launches that was           */
  int64_t cum_leaf_launches;
} m4ri_amd_stats;
/* 0: off.  1: the stats describe the most recent product only.  2: additionally cum_leaf_ms /
 * cum_leaf_launches accumulate over all products until profiling is set again (no host
 * synchronisation happens until m4ri_amd_get_stats is called). */
void m4ri_amd_set_profiling(int on);
int m4ri_amd_get_stats(m4ri_amd_stats *out);

/* Strassen-Winograd levels the engine uses for an m x l x n product and this cutoff (pure host
   logic, callable without a GPU): cutoff > 0 follows the reference's rule (strassen.c:39,51 --
   halve while no dimension satisfies 3*dim < 4*cutoff, cutoff rounded down to a multiple of 64),
   cutoff == 0 the engine's own plan: the depth a small time model of the schedule gives the first
   (largest) block of rows -- see m4ri_amd_plan_row_blocks; both capped so that every level still
   halves whole words. */
int m4ri_amd_plan_levels(int64_t m, int64_t l, int64_t n, int cutoff);

/* The engine's own plan for an m x l x n product (cutoff == 0), pure host logic: the rows of A and C
   in blocks, largest first, each multiplied by B as a product of its own at its own depth (the leaf's
   tile is 4096 rows: 65664 = 65536 + 128 rows run as a four-level product and a thin one instead of
   one level over rows that do not tile).  Writes the first `cap` blocks to rows[] / levels[] (either
   may be NULL) and returns the number of blocks of the plan. */
int m4ri_amd_plan_row_blocks(int64_t m, int64_t l, int64_t n, int64_t *rows, int *levels, int cap);

/* Does a direct (no Strassen level) product of `batch` members of this shape take the one-launch small leaf (m4rm_small.hip), and with
   how many inner-dimension splits?  0: no (generation 4 / 1); k >= 1: yes, k splits (k > 1: the splits meet by atomic XOR).  `cus`: the
   compute units the launch is planned for (0 = 256, an MI355X).  Pure host arithmetic, callable without a GPU. */
int m4ri_amd_plan_small_leaf(int64_t m, int64_t l, int64_t n, int64_t batch, int cus);
/* What the engine's time model gives ONE m x l x n product at `levels` Strassen-Winograd levels, in seconds on the box the
   constants were measured on (the plans above are minima of sums of it; tools/depth_model_sweep.py prints it beside measurements). */
double m4ri_amd_model_seconds(int64_t m, int64_t l, int64_t n, int levels);

/* Bytes the breadth-first schedule's workspace may take (0 = automatic: what the device has left).
   Levels that do not fit run depth-first, one sub-product after the other, like the reference's
   recursion -- same bits.  Returns the previous value; negative arguments only query. */
int64_t m4ri_amd_set_workspace_budget(int64_t bytes);

/* How many of the deepest Strassen-Winograd levels one fused pass covers each way (1..4, default 4;
   a scheduling knob: results are bit-identical for every value).  Returns the previous value;
   out-of-range arguments only query. */
int m4ri_amd_set_max_fuse(int levels);

/* The M4RI-named products pipeline a large product from host memory over four row slabs of A and C (upload of
   slab k+1 and download of slab k-1 under product k; same bits).  min_bytes: size of A + B + C from which that
   happens (default 64 MiB; 0 = never).  Returns the previous value; negative arguments only query. */
int64_t m4ri_amd_set_host_pipeline(int64_t min_bytes);

/* Small products.  The reference switches algorithm by size inside the functions this library replaces (_mzd_mul_m4rm -> mzd_mul_naive
   below 54 columns / 16 rows, m4ri/brilliantrussian.c:1063-1068, m4ri/mzd.c:1141-1172); here the switch sits where a call's upload,
   launches and download (28 ... 80 us whatever the size) stop paying: a product with m * l * n at or below the threshold whose
   matrices live in host memory is computed by the library's own host Method of Four Russians (small_host.cpp) on the calling
   thread -- on an initialised device, never instead of one: without a GPU the entry points still abort.  Default 2^27; 0 sends
   every product to the GPU; returns the previous value, negative arguments only query.  The routine's own cost is bounded as well
   (threshold / 240 word operations of its algorithm, 39 us at the default: 512^3 -- 31 us on the host, 40 through the GPU -- is the
   largest cube it takes, and degenerate shapes such as 1 x 1 x 2^26 go to the GPU whatever m * l * n says);
   m4ri_amd_small_product_wanted is that rule (pure arithmetic, no GPU: 1 = the host routine would take this product).  The device
   lock is released while the routine runs: small products of many threads run side by side.  m4ri_amd_small_product_count: how
   many products took that path.  m4ri_amd_small_mul_host is the routine itself: C (+)= A * B on host mzd_t (windows allowed, the
   bits outside C's columns kept); returns 0, or -1 on mismatched dimensions. */
int64_t m4ri_amd_set_small_product_threshold(int64_t ops);
int64_t m4ri_amd_small_product_count(void);
int m4ri_amd_small_product_wanted(int64_t m, int64_t l, int64_t n);
int m4ri_amd_small_mul_host(mzd_t *C, const mzd_t *A, const mzd_t *B, int add);

/* Release the engine's workspace (device memory pool) and the host entry points' staging arena;
   pinned matrices keep their device copies. */
void m4ri_amd_release_workspace(void);

/* ---- part 3: residency (device-resident matrix handles) -----------------------------
 * Opt-in, for programs that chain products (TRSM / PLE Schur complements: triangular.c:100,348,439,
 * 503, ple.c:126 call mzd_addmul on windows of the same few matrices).  A pinned matrix keeps a device
 * copy in the host layout; the M4RI-named entry points of part 1 then read pinned operands -- and
 * windows (mzd_init_window) into them -- where they are, and leave a pinned result on the device: no
 * PCIe traffic at all when everything is pinned.  The host copy of a pinned matrix that received a
 * result is STALE until m4ri_amd_sync or m4ri_amd_unpin; the host must not modify a pinned matrix
 * without telling (m4ri_amd_host_modified).  Unpin before mzd_free.  All return 0 on success, -1 if
 * M is NULL / a window / empty (pin) or not pinned (the others). */
int m4ri_amd_pin(mzd_t *M);            /* M owns its block (not a window): upload it now               */
int m4ri_amd_sync(mzd_t *M);           /* bring the host copy up to date (M or a window into it)        */
int m4ri_amd_host_modified(mzd_t *M);  /* the host wrote into M: refresh the device copy               */
int m4ri_amd_unpin(mzd_t *M);          /* sync, then drop the device copy                               */
int m4ri_amd_is_pinned(const mzd_t *M);/* 0 no, 1 yes and host copy current, 2 yes and host copy stale  */

/* ---- part 4: one product over several GPUs (m4ri/mp.c:158-324 is the reference's own
 * block-parallel template) ------------------------------------------------------------------------------
 * The units handed out are the sub-products of the top Strassen level(s) (strassen.c:111-150): 7 (levels = 1), or
 * (levels = 2) the 47 of ONE application of the rank-47 scheme of the 4 x 4 x 4 block product (scheme444.h) -- 49 of
 * Strassen-Winograd twice where the scheme's passes cannot take the slabs (children narrower than 64 words).  Layout: with S = 2^levels row blocks per matrix,
 * rank r of `world` holds rows [cut(r), cut(r+1)) of EVERY block ("slab-cyclic"; cut(r) = rows*r/world),
 * stacked into a local parent of 1/world of the rows and the same quadrant structure -- so the level's
 * additions are ordinary local passes and only slabs of sub-product operands (one way) and of products
 * (the other) cross the links, every rank to every rank.  Dimensions are zero-padded (M, L, N). */
typedef struct m4ri_amd_shard_plan {
  int32_t world, levels, nprod, blocks; /* ranks; sharded levels (1|2); sub-products: 7, 49 or -- two levels as ONE application of the
                                           rank-R scheme of the 4 x 4 x 4 block product, where its passes take the slabs -- R = 47; 2^levels */
  int64_t m, l, n;                      /* the product C (m x n) = A (m x l) * B (l x n)                */
  int64_t M, L, N;                      /* padded: M % blocks == 0, L and N % (64*blocks) == 0          */
  int64_t bm, bl;                       /* rows of a sub-product's A operand (= of its result), of its B */
  int64_t cwl, cwn;                     /* words per row of an A child; of a B child / a product        */
} m4ri_amd_shard_plan;
/* One slab of one sub-product operand (side 0: A child, 1: B child) or result (side 2): it lives at
 * holder_off words into rank `holder`'s child / slab array and at owner_off words into rank `owner`'s
 * operand / product array; sides 0, 1 travel holder -> owner, side 2 owner -> holder. */
typedef struct m4ri_amd_shard_piece {
  int32_t holder, owner;
  int64_t holder_off, owner_off, words;
} m4ri_amd_shard_piece;
enum { M4RI_AMD_SHARD_BUF_LOCAL_A = 0, M4RI_AMD_SHARD_BUF_LOCAL_B, M4RI_AMD_SHARD_BUF_LOCAL_C, /* local parents (stride L/64, N/64, N/64) */
       M4RI_AMD_SHARD_BUF_CHILD_A, M4RI_AMD_SHARD_BUF_CHILD_B, M4RI_AMD_SHARD_BUF_SLABS_P,      /* nprod slabs each, back to back         */
       M4RI_AMD_SHARD_BUF_OPER_A, M4RI_AMD_SHARD_BUF_OPER_B, M4RI_AMD_SHARD_BUF_PROD };         /* whole operands/results of owned j's    */
/* Pure host arithmetic (no GPU needed).  levels: 1, 2 or 0 = automatic.  0 on success. */
int m4ri_amd_shard_plan_make(m4ri_amd_shard_plan *p, int world, int64_t m, int64_t l, int64_t n, int levels);
int64_t m4ri_amd_shard_cut(int64_t rows, int world, int r);
int m4ri_amd_shard_owner(const m4ri_amd_shard_plan *p, int j);             /* rank that multiplies sub-product j */
/* how many of its sub-products a rank multiplies as ONE batched product (m4ri_amd_mul_batch_dev): rounds q, q + 1, ... of a group go
 * together -- by the engine's time model, the smallest group within 5 % of the best; 1 = one sub-product at a time.  Host arithmetic. */
int m4ri_amd_shard_group(const m4ri_amd_shard_plan *p, int cutoff);
int64_t m4ri_amd_shard_slab_rows(const m4ri_amd_shard_plan *p, int rank, int which); /* 0: of A/C/products, 1: of B */
int64_t m4ri_amd_shard_buffer_words(const m4ri_amd_shard_plan *p, int rank, int which);
int m4ri_amd_shard_piece_of(const m4ri_amd_shard_plan *p, int side, int j, int r, m4ri_amd_shard_piece *out);
/* Per-rank device steps (asynchronous on `stream`; one process per GPU moves the pieces itself, e.g. with
 * RCCL send/recv): local parents -> slabs of all children; slabs of all products -> local parent of C.
 * The sub-products themselves are plain m4ri_amd_mul_dev calls of bm x bl x (64*cwn). */
int m4ri_amd_shard_down_dev(const m4ri_amd_shard_plan *p, int rank, const word *A_local, int64_t a_stride, const word *B_local,
                            int64_t b_stride, word *child_a, word *child_b, void *stream);
int m4ri_amd_shard_up_dev(const m4ri_amd_shard_plan *p, int rank, const word *slabs_p, word *C_local, int64_t c_stride, int add,
                          void *stream);
/* Rows [row0, row0 + rows) of the matrix m4ri_amd_fill_dev(seed) would fill, written to M (for filling
 * local parents of a distributed matrix without materialising the whole). */
int m4ri_amd_fill_rows_dev(word *M, int64_t stride, int64_t row0, int64_t rows, int64_t ncols, uint64_t seed, void *stream);
/* ---- one process, all devices: distributed device-resident matrices ----------------------------------
 * The reference's multi-core entry is a C function (mzd_mul_mp, m4ri/mp.c:158-297), so the multi-GPU schedules sit
 * behind this boundary.  A m4ri_amd_dmat is a matrix spread over the configured devices (m4ri_amd_set_devices) and
 * resident in HBM; products leave their result distributed, so they chain without PCIe traffic.  Layouts (every
 * dimension zero-padded to 256 bits in the local buffers, one row stride for all layouts of a matrix):
 *   ROWS        rank r holds rows [r k, (r+1) k), k = ceil(rows / W)            -- the row-slab schedule
 *   CYCLIC1/2   slab-cyclic over the 2 / 4 row blocks of 1 / 2 Strassen-Winograd levels (above)
 *   REPLICATED  every rank holds the whole matrix (a B that needs no gather)
 * Schedules (m4ri_amd_dmat_mul, `variant`): M4RI_AMD_VARIANT_SLABS -- C_r = A_r * B, B's row slabs gathered by peer
 * copies under the product with the rank's own slab (no reduction; SURVEY.md 8e "rectangular"); M4RI_AMD_VARIANT_STRASSEN
 * -- the sub-products of the top Strassen level(s), operand and result slabs pulled by hipMemcpyPeerAsync on copy streams
 * under the products (row chunks).  0 = by the operands' layout, else by shape and world size
 * (m4ri_amd_multi_default_variant: slabs up to 4 ranks and for every product a Strassen level cannot help, e.g.
 * 131072 x 8192 x 131072).  One host thread per rank issues the work; all calls but _mul are blocking, _mul is
 * asynchronous (m4ri_amd_multi_sync).  Return 0 or a hipError_t value unless noted. */
enum { M4RI_AMD_LAYOUT_ROWS = 0, M4RI_AMD_LAYOUT_CYCLIC1 = 1, M4RI_AMD_LAYOUT_CYCLIC2 = 2, M4RI_AMD_LAYOUT_REPLICATED = 3 };
enum { M4RI_AMD_VARIANT_AUTO = 0, M4RI_AMD_VARIANT_SLABS = 1, M4RI_AMD_VARIANT_STRASSEN = 2 };
typedef struct m4ri_amd_dmat m4ri_amd_dmat;
typedef struct m4ri_amd_dmat_info_t {
  int64_t rows, ncols, stride;  /* stride: words of a local row */
  int32_t layout, world, alive; /* alive == 0: the device list changed since the matrix was made -- only _free is valid */
} m4ri_amd_dmat_info_t;
m4ri_amd_dmat *m4ri_amd_dmat_create(int64_t rows, int64_t ncols, int layout);  /* zero matrix; NULL on failure */
void m4ri_amd_dmat_free(m4ri_amd_dmat *d);
int m4ri_amd_dmat_info(const m4ri_amd_dmat *d, m4ri_amd_dmat_info_t *out);
/* device pointer, rows (padding included) and HIP device of rank `rank`'s local buffer */
int m4ri_amd_dmat_local(const m4ri_amd_dmat *d, int rank, word **data, int64_t *local_rows, int *device);
int m4ri_amd_dmat_fill(m4ri_amd_dmat *d, uint64_t seed);                /* the matrix m4ri_amd_fill_dev(seed) fills, distributed */
int m4ri_amd_dmat_upload(m4ri_amd_dmat *d, const mzd_t *M);             /* every device its own rows over its own PCIe link */
int m4ri_amd_dmat_download(const m4ri_amd_dmat *d, mzd_t *M);           /* a windowed M keeps the bits outside its columns */
int m4ri_amd_dmat_convert(m4ri_amd_dmat *dst, const m4ri_amd_dmat *src); /* dst <- src, any two layouts (ROWS -> REPLICATED = all-gather) */
int m4ri_amd_dmat_mul(m4ri_amd_dmat *C, const m4ri_amd_dmat *A, const m4ri_amd_dmat *B, int add, int cutoff, int variant);
/* The same on lane `lane` (0 or 1): every lane has its own streams, events and arenas on every device, and products of different
 * lanes do not wait for each other -- two INDEPENDENT products (different C, neither C an operand of the other) issued
 * alternately on lanes 0 and 1 keep two in flight: the operand and result transport of one under the multiplications of the
 * other (on a device the multiplications themselves take turns: one engine workspace).  Everything that is not a product
 * (fill, upload, convert, download) runs on lane 0 behind the last operation of EVERY lane, and the first product of a lane after
 * such an operation waits for it.  m4ri_amd_dmat_mul == lane 0. */
int m4ri_amd_dmat_mul_lane(m4ri_amd_dmat *C, const m4ri_amd_dmat *A, const m4ri_amd_dmat *B, int add, int cutoff, int variant, int lane);
int m4ri_amd_multi_sync(void);
/* First contact with DIFFERENT devices (the first call that configures the ranks): peer access is enabled pair by pair, then every
 * ordered pair makes one 1 MiB copy behind one cross-device event wait and checks the data; a failure prints the pair and the HIP
 * error and fails the call before any product is scheduled.  Pairs without peer access copy through pinned host memory instead
 * (m4ri_amd_multi_stats.pairs_staged counts them).  Environment, for tests on one GPU: M4RI_AMD_SELFTEST_PAIRS=1 runs the self-test
 * between ranks of one device too; M4RI_AMD_NO_PEER="all" | "i-j,k-l" treats these rank pairs as having no peer access.
 * m4ri_amd_multi_pair_table is the pure rule (no GPU): staged[dst * world + src] from the ranks' devices, the ndev x ndev table
 * can_access[d * ndev + e] (hipDeviceCanAccessPeer) and that hook. */
void m4ri_amd_multi_pair_table(int world, const int *devices, int ndev, const int *can_access, const char *no_peer_spec, char *staged);
/* The links, measured over the ranks' own link streams: `bytes` copied dst <- src for every ordered pair of ranks one at a time
 * (pair_gbs[dst * W + src] in GB/s) and all pairs at once (*all_gbs: all bytes / wall time); staged[] as above; *same_device = 1 when
 * ranks share a device (those pairs report the on-device copy rate).  all_gbs, staged, same_device may be NULL. */
int m4ri_amd_multi_link_probe(int64_t bytes, double *pair_gbs, double *all_gbs, char *staged, int *same_device);
/* what the most recent m4ri_amd_dmat_mul / m4ri_amd_mul_multi / mzd_mul_mp did */
typedef struct m4ri_amd_multi_stats {
  int32_t world, variant, levels, sub_products; /* variant: M4RI_AMD_VARIANT_*; levels / sub_products: Strassen schedule   */
  int32_t chunks, overlap, converted;           /* row chunks per sub-product; slabs: gather under the first product;       */
  int32_t pairs_staged;                         /* operands converted to the schedule's layout first (0 on the fast path);  */
                                                /* ordered rank pairs that copy through the host (no peer access)           */
  int64_t m, l, n;
  double link_bytes;                            /* bytes that crossed between ranks                                          */
  int32_t group, reserved;                      /* Strassen schedule: sub-products of a rank multiplied as one batched product */
} m4ri_amd_multi_stats;
int m4ri_amd_multi_get_stats(m4ri_amd_multi_stats *out);
/* Marks of rank `rank`'s part of the most recent m4ri_amd_dmat_mul, ms after its compute stream entered the operation
 * (synchronises first).  Strassen: down pass done; per unit (round, row chunk): operands in, product done; result slabs
 * in; up pass done.  Slabs: gather done; first product done; all done.  Returns the number of marks, < 0 on error. */
int m4ri_amd_multi_timeline(int rank, double *ms, int cap);
/* Pure host arithmetic (no GPU needed): the schedule a product takes, the layout that schedule wants, and the geometry
 * of a layout -- rows of rank `rank`'s local buffer; its runs of valid global rows (global first row, rows, local first
 * row; returns their number, at most 4). */
int m4ri_amd_multi_default_variant(int world, int64_t m, int64_t l, int64_t n);
int m4ri_amd_multi_layout_for(int variant, int world, int64_t m, int64_t l, int64_t n);
int64_t m4ri_amd_layout_local_rows(int layout, int world, int rank, int64_t rows);
int m4ri_amd_layout_runs(int layout, int world, int rank, int64_t rows, int64_t *g0, int64_t *nrows, int64_t *l0, int cap);
/* Force a schedule for m4ri_amd_mul_multi / mzd_mul_mp and for m4ri_amd_dmat_mul on operands whose layouts fit none
 * (0 automatic, M4RI_AMD_VARIANT_SLABS, M4RI_AMD_VARIANT_STRASSEN); returns the previous value, others only query. */
int m4ri_amd_set_multi_variant(int variant);
/* One process, several devices, HOST matrices: C (+)= A*B = upload into distributed temporaries (kept between calls) +
 * m4ri_amd_dmat_mul + download.  levels: 0 = automatic schedule (above), 1 / 2 = the Strassen schedule with that many
 * sharded levels.  What mzd_mul_mp / mzd_addmul_mp run for large products.  Returns 0 or a hipError_t value. */
int m4ri_amd_mul_multi(mzd_t *C, const mzd_t *A, const mzd_t *B, int add, int cutoff, int levels);
/* The devices mzd_mul_mp / m4ri_amd_mul_multi / the distributed matrices are spread over.  Default: the comma-separated
 * list in the environment variable M4RI_AMD_DEVICES, else every visible device.  An id may repeat (several ranks
 * on one GPU: how the tests exercise the path on a one-GPU box).  n == 0 restores the default. */
int m4ri_amd_set_devices(int n, const int *ids);
int m4ri_amd_get_device_list(int *ids, int cap);
/* Smallest min(m, l, n) that mzd_mul_mp / mzd_addmul_mp spread over several devices (default 16384);
 * returns the previous value, negative arguments only query. */
int64_t m4ri_amd_set_multi_threshold(int64_t min_dim);

#ifdef __cplusplus
}
#endif
#endif /* M4RI_AMD_H */
