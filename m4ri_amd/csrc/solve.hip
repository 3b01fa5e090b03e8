// solve.hip -- the drivers over PLUQ and the triangular solves, on device-resident matrices: row permutations, linear
// systems, left kernels, inverses.  Everything heavy is a call into ple.hip / trsm.hip / echelon.hip / the multiply
// engine; what lives here is the glue the reference keeps in
//   mzd_apply_p_left / mzd_apply_p_left_trans          /root/reference m4ri/mzp.c:65-81
//   _mzd_pluq_solve_left / _mzd_solve_left              m4ri/solve.c:57-152
//   mzd_kernel_left_pluq                                m4ri/solve.c:154-191
//   mzd_inv_m4ri                                        m4ri/brilliantrussian.c:971-997
// done without leaving the device between the steps (the reference's drivers, run over this library through the PLT,
// cross PCIe once per step).
#include <hip/hip_runtime.h>
#include <vector>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

namespace {

#define HIPTRY(expr)                                  \
  do {                                                \
    hipError_t e_ = (hipError_t)(expr);               \
    if (e_ != hipSuccess) return (int)e_;             \
  } while (0)

constexpr int SV_THREADS = 256;

// dst row k <- src row idx[k] (whole rows of `width` words)
__global__ __launch_bounds__(SV_THREADS) void gather_rows_kernel(word *__restrict__ dst, int64_t d_stride, const word *__restrict__ src, int64_t s_stride,
                                                                 int64_t width, const int32_t *__restrict__ idx) {
  const word *s = src + (int64_t)idx[blockIdx.x] * s_stride;
  word *d       = dst + (int64_t)blockIdx.x * d_stride;
  for (int64_t w = threadIdx.x; w < width; w += SV_THREADS) d[w] = s[w];
}
// dst row idx[k] <- src row k
__global__ __launch_bounds__(SV_THREADS) void scatter_rows_kernel(word *__restrict__ dst, int64_t d_stride, const word *__restrict__ src, int64_t s_stride,
                                                                  int64_t width, const int32_t *__restrict__ idx) {
  const word *s = src + (int64_t)blockIdx.x * s_stride;
  word *d       = dst + (int64_t)idx[blockIdx.x] * d_stride;
  for (int64_t w = threadIdx.x; w < width; w += SV_THREADS) d[w] = s[w];
}

// *flag |= 1 when any word of the rows x width block is non-zero
__global__ __launch_bounds__(SV_THREADS) void any_nonzero_kernel(const word *__restrict__ A, int64_t stride, int64_t rows, int64_t width, int *flag) {
  const int64_t r = blockIdx.x;
  word any = 0;
  for (int64_t w = threadIdx.x; w < width; w += SV_THREADS) any |= A[r * stride + w];
  if (__syncthreads_or(any != 0) && threadIdx.x == 0) atomicOr(flag, 1);
}

// dst (rows x dcols bits, aligned) <- the columns [c0, c0 + dcols) of src: a funnel shift per word
__global__ __launch_bounds__(SV_THREADS) void extract_cols_kernel(word *__restrict__ dst, int64_t d_stride, const word *__restrict__ src, int64_t s_stride,
                                                                  int64_t s_width, int64_t c0, int64_t dcols) {
  const int64_t r = blockIdx.x, dw = (dcols + 63) >> 6;
  const int sh = (int)(c0 % 64);
  const int64_t w0 = c0 / 64;
  for (int64_t w = threadIdx.x; w < dw; w += SV_THREADS) {
    const word lo = src[r * s_stride + w0 + w];
    const word hi = (sh && w0 + w + 1 < s_width) ? src[r * s_stride + w0 + w + 1] : 0;
    word v        = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    if (w == dw - 1 && dcols % 64) v &= ((word)1 << (dcols % 64)) - 1;
    dst[r * d_stride + w] = v;
  }
}

// bit (r0 + i, c0 + i) <- 1 for i < count
__global__ void set_diagonal_kernel(word *A, int64_t stride, int64_t r0, int64_t c0, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) atomicOr(reinterpret_cast<unsigned long long *>(A + (r0 + i) * stride + (c0 + i) / 64), 1ull << ((c0 + i) % 64));
}

int zero_rows(word *A, int64_t stride, int64_t r0, int64_t r1, int64_t width, hipStream_t st) {
  if (r1 <= r0 || width <= 0) return 0;
  return (int)hipMemset2DAsync(A + r0 * stride, (size_t)stride * 8, 0, (size_t)width * 8, (size_t)(r1 - r0), st);
}

int is_zero(const word *A, int64_t stride, int64_t rows, int64_t width, hipStream_t st, bool *zero) {
  *zero = true;
  if (rows <= 0 || width <= 0) return 0;
  int *d = nullptr, h = 0;
  HIPTRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(int)));
  int rc = (int)hipMemsetAsync(d, 0, sizeof(int), st);
  if (!rc) {
    hipLaunchKernelGGL(any_nonzero_kernel, dim3((unsigned)rows), dim3(SV_THREADS), 0, st, A, stride, rows, width, d);
    rc = (int)hipGetLastError();
  }
  if (!rc) rc = (int)hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, st);
  if (!rc) rc = (int)hipStreamSynchronize(st);
  (void)hipFree(d);
  *zero = h == 0;
  return rc;
}

// a clean copy of the leading r x r block of A (its last word masked to r columns): the triangular solves split their
// triangle into blocks that go to the multiply engine as operands, which must not carry bits of the neighbouring columns
int square_copy(word **out, const word *A, int64_t stride, int64_t r, hipStream_t st) {
  *out = nullptr;
  if (r <= 0) return 0;
  const int64_t wr = words_of(r);
  HIPTRY(hipMalloc(reinterpret_cast<void **>(out), (size_t)r * wr * 8));
  HIPTRY(hipMemcpy2DAsync(*out, (size_t)wr * 8, A, (size_t)stride * 8, (size_t)wr * 8, (size_t)r, hipMemcpyDeviceToDevice, st));
  return m4ri_amd_mask_tail_dev(*out, wr, r, r, st);
}

}  // namespace

extern "C" {

// The row transpositions (i, P[i]), i < min(length, nrows), ascending (trans == 0: mzd_apply_p_left) or descending
// (mzd_apply_p_left_trans), mzp.c:65-81.  P: HOST array.  Blocking.
int m4ri_amd_apply_p_left_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, const int32_t *P, int64_t length, int trans, void *stream) {
  if (nrows < 0 || ncols < 0 || length < 0 || !P) return (int)hipErrorInvalidValue;
  if (ncols == 0 || nrows == 0) return 0;
  if (length > nrows) length = nrows;
  for (int64_t i = 0; i < length; ++i)
    if (P[i] < 0 || P[i] >= nrows) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  std::vector<int32_t> src((size_t)nrows);  // src[r]: the row that ends up at position r
  for (int64_t r = 0; r < nrows; ++r) src[(size_t)r] = (int32_t)r;
  for (int64_t t = 0; t < length; ++t) {
    const int64_t i = trans ? length - 1 - t : t;
    const int32_t x = src[(size_t)i];
    src[(size_t)i]    = src[(size_t)P[i]];
    src[(size_t)P[i]] = x;
  }
  std::vector<int32_t> moved, from;
  for (int64_t r = 0; r < nrows; ++r)
    if (src[(size_t)r] != r) { moved.push_back((int32_t)r); from.push_back(src[(size_t)r]); }
  if (moved.empty()) return (int)hipStreamSynchronize(st);  // "blocking": whatever the caller queued on A is complete on return
  const int64_t width = words_of(ncols), k = (int64_t)moved.size();
  word *tmp = nullptr;
  int32_t *d_idx = nullptr;
  auto run = [&]() -> int {
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)k * width * 8));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_idx), (size_t)k * 8));
    HIPTRY(hipMemcpyAsync(d_idx, from.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
    HIPTRY(hipMemcpyAsync(d_idx + k, moved.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)k), dim3(SV_THREADS), 0, st, tmp, width, A, stride, width, d_idx);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)k), dim3(SV_THREADS), 0, st, A, stride, tmp, width, width, d_idx + k);
    HIPTRY(hipGetLastError());
    HIPTRY(hipStreamSynchronize(st));
    return 0;
  };
  const int rc = run();
  if (tmp) (void)hipFree(tmp);
  if (d_idx) (void)hipFree(d_idx);
  return rc;
}

// B <- the solution steps of _mzd_pluq_solve_left (solve.c:57-121) for a device matrix A that holds its PLUQ
// decomposition (rank, P, Q: HOST): P^T, L^-1 on the first `rank` rows, optionally the consistency check on the rows
// behind them (*retval = -1 when they do not vanish), U^-1, zero the undefined rows, Q^T.  B: b_rows x b_cols with
// b_rows >= max(m, n).  Blocking.
int m4ri_amd_pluq_solve_left_dev(const word *A, int64_t a_stride, int64_t m, int64_t n, int32_t rank, const int32_t *P, const int32_t *Q, word *B,
                                 int64_t b_stride, int64_t b_rows, int64_t b_cols, int cutoff, int inconsistency_check, int *retval, void *stream) {
  if (!retval || !P || !Q || m < 0 || n < 0 || rank < 0 || rank > m || rank > n || b_rows < m || b_rows < n || b_cols < 0 || cutoff < 0)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  *retval = 0;
  const int64_t bw = words_of(b_cols), wr = words_of(rank);
  word *LU = nullptr;
  auto run = [&]() -> int {
    if (int rc = m4ri_amd_apply_p_left_dev(B, b_stride, b_rows, b_cols, P, m, 0, st)) return rc;                       // solve.c:72
    if (int rc = square_copy(&LU, A, a_stride, rank, st)) return rc;                                                   // :76
    if (int rc = m4ri_amd_trsm_lower_left_dev(LU, wr, B, b_stride, rank, b_cols, cutoff, st)) return rc;               // :77-79
    if (inconsistency_check) {                                                                                         // :81-98
      if (m < b_rows) HIPTRY(zero_rows(B, b_stride, m, b_rows, bw, st));
      if (m > rank && rank > 0 && b_cols > 0)  // the rows of A behind the rank hold L only: nothing beyond column `rank`
        HIPTRY(m4ri_amd_mul_dev(B + (int64_t)rank * b_stride, b_stride, A + (int64_t)rank * a_stride, a_stride, B, b_stride, m - rank, rank, b_cols, 1,
                                cutoff, st));
      bool zero = true;
      if (int rc = is_zero(B + (int64_t)rank * b_stride, b_stride, m - rank, bw, st, &zero)) return rc;
      if (!zero) *retval = -1;
    }
    if (int rc = m4ri_amd_trsm_upper_left_dev(LU, wr, B, b_stride, rank, b_cols, cutoff, st)) return rc;               // :100
    if (!inconsistency_check) HIPTRY(zero_rows(B, b_stride, rank, b_rows, bw, st));                                    // :104-114
    if (int rc = m4ri_amd_apply_p_left_dev(B, b_stride, b_rows, b_cols, Q, n, 1, st)) return rc;                       // :116
    return (int)hipStreamSynchronize(st);
  };
  const int rc = run();
  if (LU) (void)hipFree(LU);
  return rc;
}

// _mzd_solve_left (solve.c:123-152): A <- its PLUQ decomposition (the reference's flavour, recursion leftovers in Q
// included), B <- a solution X of A X = B with the undefined rows zero; *retval = -1 when the check finds none.
int m4ri_amd_solve_left_dev(word *A, int64_t a_stride, int64_t m, int64_t n, word *B, int64_t b_stride, int64_t b_rows, int64_t b_cols, int cutoff,
                            int inconsistency_check, int *retval, void *stream) {
  if (!retval || m < 0 || n < 0 || b_rows < m || b_rows < n) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  *retval = 0;
  if (inconsistency_check && b_rows > m) {  // :124-128 -- the window starts one row late in the reference, so does this
    bool zero = true;
    if (int rc = is_zero(B + (m + 1) * b_stride, b_stride, b_rows - (m + 1), words_of(b_cols), st, &zero)) return rc;
    if (!zero) { *retval = -1; return 0; }
  }
  std::vector<int32_t> P((size_t)(m > 0 ? m : 1)), Q((size_t)(n > 0 ? n : 1));
  int32_t rank = 0;
  if (int rc = m4ri_amd_pluq_dev(A, a_stride, m, n, P.data(), Q.data(), &rank, M4RI_AMD_PLE_CUTOFF, st)) return rc;
  return m4ri_amd_pluq_solve_left_dev(A, a_stride, m, n, rank, P.data(), Q.data(), B, b_stride, b_rows, b_cols, cutoff, inconsistency_check, retval, st);
}

// mzd_kernel_left_pluq (solve.c:154-191): A <- its PLUQ decomposition; when rank < n, R (n x (n - rank), device,
// zeroed by the caller, stride r_stride) <- a basis of the right kernel {x : A x = 0} as the reference lays it out.
// *rank_out = the rank; R is untouched when the rank is n.
int m4ri_amd_kernel_left_pluq_dev(word *A, int64_t a_stride, int64_t m, int64_t n, word *R, int64_t r_stride, int cutoff, int32_t *rank_out,
                                  void *stream) {
  if (!rank_out || m < 0 || n < 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  std::vector<int32_t> P((size_t)(m > 0 ? m : 1)), Q((size_t)(n > 0 ? n : 1));
  int32_t r = 0;
  if (int rc = m4ri_amd_pluq_dev(A, a_stride, m, n, P.data(), Q.data(), &r, M4RI_AMD_PLE_CUTOFF, st)) return rc;
  *rank_out = r;
  if (r == n) return (int)hipStreamSynchronize(st);
  const int64_t kc = n - r;
  if (r > 0) {
    hipLaunchKernelGGL(extract_cols_kernel, dim3((unsigned)r), dim3(SV_THREADS), 0, st, R, r_stride, A, a_stride, words_of(n), (int64_t)r, kc);   // :170-175
    HIPTRY(hipGetLastError());
    word *U = nullptr;
    int rc  = square_copy(&U, A, a_stride, r, st);
    if (!rc) rc = m4ri_amd_trsm_upper_left_dev(U, words_of(r), R, r_stride, r, kc, cutoff, st);                                                   // :177
    if (!rc) rc = (int)hipStreamSynchronize(st);
    if (U) (void)hipFree(U);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(set_diagonal_kernel, dim3((unsigned)((kc + 255) / 256)), dim3(256), 0, st, R, r_stride, (int64_t)r, (int64_t)0, kc);        // :179
  HIPTRY(hipGetLastError());
  if (int rc = m4ri_amd_apply_p_left_dev(R, r_stride, n, kc, Q.data(), n, 1, st)) return rc;                                                  // :180
  return (int)hipStreamSynchronize(st);  // blocking as documented, also when Q moves no row (apply_p returns early then)
}

// mzd_inv_m4ri (brilliantrussian.c:971-997): Binv (n x n, device) <- the right block of the reduced row echelon form of
// [A | 0 | I] (2 * 64 * ceil(n / 64) columns): A^-1 when A is invertible, whatever the elimination leaves otherwise.
// An invertible A has one inverse however it is reached, so the common case takes the cheaper road -- PLUQ of A itself
// (n columns instead of 2n), then A X = I through the decomposition (solve.c:41-97 with B = I) -- and only a singular A
// goes through the reference's augmented elimination, whose leftovers are part of its result.
int m4ri_amd_inv_dev(word *Binv, int64_t b_stride, const word *A, int64_t a_stride, int64_t n, void *stream) {
  if (n < 0) return (int)hipErrorInvalidValue;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t wn = words_of(n), cw = 2 * wn, ws = (wn + 1) & ~(int64_t)1;
  word *C = nullptr;
  HIPTRY(hipMalloc(reinterpret_cast<void **>(&C), (size_t)n * (size_t)(cw > ws ? cw : ws) * 8));
  auto by_decomposition = [&](bool &done) -> int {
    done = false;
    HIPTRY(hipMemcpy2DAsync(C, (size_t)ws * 8, A, (size_t)a_stride * 8, (size_t)wn * 8, (size_t)n, hipMemcpyDeviceToDevice, st));
    if (ws != wn) HIPTRY(hipMemset2DAsync(C + wn, (size_t)ws * 8, 0, 8, (size_t)n, st));
    HIPTRY(m4ri_amd_mask_tail_dev(C, ws, n, n, st));
    std::vector<int32_t> P((size_t)n), Q((size_t)n);
    int32_t rank = 0;
    if (int rc = m4ri_amd_pluq_dev(C, ws, n, n, P.data(), Q.data(), &rank, 0, st)) return rc;
    if (rank != n) return 0;
    HIPTRY(hipMemset2DAsync(Binv, (size_t)b_stride * 8, 0, (size_t)wn * 8, (size_t)n, st));
    hipLaunchKernelGGL(set_diagonal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Binv, b_stride, (int64_t)0, (int64_t)0, n);
    HIPTRY(hipGetLastError());
    int ret = 0;
    if (int rc = m4ri_amd_pluq_solve_left_dev(C, ws, n, n, rank, P.data(), Q.data(), Binv, b_stride, n, n, 0, 0, &ret, st)) return rc;
    done = true;
    return (int)hipStreamSynchronize(st);
  };
  auto by_elimination = [&]() -> int {
    HIPTRY(hipMemsetAsync(C, 0, (size_t)n * cw * 8, st));
    HIPTRY(hipMemcpy2DAsync(C, (size_t)cw * 8, A, (size_t)a_stride * 8, (size_t)wn * 8, (size_t)n, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(set_diagonal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, C, cw, (int64_t)0, wn * 64, n);
    HIPTRY(hipGetLastError());
    int32_t rank = 0;
    if (int rc = m4ri_amd_echelonize_dev(C, cw, n, cw * 64, 1, &rank, st)) return rc;
    HIPTRY(hipMemcpy2DAsync(Binv, (size_t)b_stride * 8, C + wn, (size_t)cw * 8, (size_t)wn * 8, (size_t)n, hipMemcpyDeviceToDevice, st));
    return (int)hipStreamSynchronize(st);
  };
  bool done = false;
  int rc = by_decomposition(done);
  if (rc == 0 && !done) rc = by_elimination();
  (void)hipFree(C);
  return rc;
}

}  // extern "C"
