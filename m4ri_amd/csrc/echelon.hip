// echelon.hip -- row echelon forms on the device, and the column permutations they need.
//
// Reference interfaces replaced:
//   mzd_echelonize / mzd_echelonize_m4ri / mzd_echelonize_pluq   /root/reference m4ri/echelonform.c:29-139
//   _mzd_echelonize_m4ri                                          m4ri/brilliantrussian.c:603-841
//   mzd_apply_p_right / mzd_apply_p_right_trans                   m4ri/mzp.c:193-260
// The three drivers of the reference differ in schedule only (strips of 6k columns with Gray-code tables; PLE / PLUQ
// plus a triangular solve; a density heuristic that switches between the two): with the common pivoting rule -- columns
// left to right, first row at or below the rank with the bit set, swapped up -- every row ends as itself plus the one
// combination of pivot rows that clears its pivot columns, so the result is fixed (tests/test_echelon_oracle.py pins
// that claim to all three).  Here both forms go through the decomposition kernels of ple.hip:
//   full = 0: PLE, then the rows keep their E part (echelonform.c:116-125: bits 0 .. i of row i cleared, the pivot
//             bit written at column Q[i]) and the rows behind the rank are zeroed (:128-132);
//   full = 1: PLUQ, B <- U^-1 B on the columns behind the rank (:66-103), U <- identity (:105), the columns of the first
//             `rank` rows back under Q (mzd_apply_p_right, :108-112), zero rows behind the rank.
// The reference copies the word that straddles column `rank` out and back because its solver wants word-aligned
// operands (:74-101); here U is copied aside whole (HBM is plentiful) and the solve runs on every word from rank/64 on:
// on the columns of U that share that word it produces U^-1 U = identity, which is what :105 writes anyway.
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

constexpr int EC_THREADS = 256;

// new row[c] = old row[map[c]] for the words [w0, w1) of `rows` rows, the same map for every row: a workgroup per row,
// the row's words in LDS (or, beyond 64 KiB, in a copy made by the caller), a wave per output word, lane = bit.
template <bool LDSROW>
__global__ __launch_bounds__(EC_THREADS) void colperm_gather_kernel(word *__restrict__ A, int64_t stride, int64_t width, int64_t w0, int64_t w1,
                                                                    int64_t ncols, const uint32_t *__restrict__ map, const word *__restrict__ rowcopy) {
  extern __shared__ word lrow[];
  word *row = A + (int64_t)blockIdx.x * stride;
  const word *src;
  if (LDSROW) {
    for (int64_t w = threadIdx.x; w < width; w += EC_THREADS) lrow[w] = row[w];
    __syncthreads();
    src = lrow;
  } else {
    src = rowcopy + (int64_t)blockIdx.x * width;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t w = w0 + wave; w < w1; w += EC_THREADS / 64) {
    const int64_t c = w * 64 + lane;
    int bit         = 0;
    if (c < ncols) {
      const uint32_t sc = map[c];
      bit               = (int)((src[sc >> 6] >> (sc & 63)) & 1);
    }
    const word v = __ballot(bit);
    if (lane == 0) row[w] = v;
  }
}

// full = 0, after the PLE (echelonform.c:116-132): row i < rank loses its bits 0 .. i (the multipliers) and gets its
// pivot bit at column Q[i]; the rows from `rank` on are zeroed.
__global__ __launch_bounds__(EC_THREADS) void echelon_from_ple_kernel(word *__restrict__ A, int64_t stride, int64_t width, int64_t nrows, int rank,
                                                                      const int32_t *__restrict__ Q) {
  const int64_t i = blockIdx.x;
  word *row       = A + i * stride;
  if (i >= rank) {
    for (int64_t w = threadIdx.x; w < width; w += EC_THREADS) row[w] = 0;
    return;
  }
  const int64_t q = Q[i];
  for (int64_t w = threadIdx.x; w <= q / 64 || w <= i / 64; w += EC_THREADS) {
    if (w >= width) break;
    word v = row[w];
    if (w < i / 64) v = 0;
    else if (w == i / 64) v &= (i % 64 == 63) ? (word)0 : (~(word)0 << (i % 64 + 1));
    if (w == q / 64) v |= (word)1 << (q % 64);
    row[w] = v;
  }
}

// full = 1: the first `rank` columns of the first `rank` rows <- identity (echelonform.c:105); rows behind the rank <- 0.
__global__ __launch_bounds__(EC_THREADS) void echelon_identity_kernel(word *__restrict__ A, int64_t stride, int64_t width, int64_t nrows, int rank) {
  const int64_t i = blockIdx.x;
  word *row       = A + i * stride;
  if (i >= rank) {
    for (int64_t w = threadIdx.x; w < width; w += EC_THREADS) row[w] = 0;
    return;
  }
  const int64_t wr = rank / 64;
  for (int64_t w = threadIdx.x; w <= wr && w < width; w += EC_THREADS) {
    const word id = (w == i / 64) ? ((word)1 << (i % 64)) : 0;
    if (w < wr) row[w] = id;
    else if (rank % 64) {
      const word low = ((word)1 << (rank % 64)) - 1;
      row[w]         = (row[w] & ~low) | id;
    }
  }
}

// the map of mzd_apply_p_right{,_trans} (mzp.c:204-216): the transpositions replayed on the identity arrangement
void build_map(std::vector<uint32_t> &map, const int32_t *P, int64_t length, int64_t ncols, bool trans, int64_t *lo, int64_t *hi) {
  map.resize((size_t)ncols);
  for (int64_t c = 0; c < ncols; ++c) map[(size_t)c] = (uint32_t)c;
  for (int64_t t = 0; t < length; ++t) {
    const int64_t i = trans ? t : length - 1 - t;
    const uint32_t x = map[(size_t)i];
    map[(size_t)i]    = map[(size_t)P[i]];
    map[(size_t)P[i]] = x;
  }
  *lo = ncols; *hi = -1;
  for (int64_t c = 0; c < ncols; ++c)
    if (map[(size_t)c] != (uint32_t)c) { if (c < *lo) *lo = c; *hi = c; }
}

int apply_map(word *A, int64_t stride, int64_t rows, int64_t ncols, const std::vector<uint32_t> &map, int64_t lo, int64_t hi, hipStream_t st) {
  if (rows <= 0 || hi < lo) return 0;
  const int64_t width = words_of(ncols);
  uint32_t *d_map = nullptr;
  word *d_copy    = nullptr;
  const bool ldsrow = width * 8 <= 64 * 1024;
  auto run = [&]() -> int {
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_map), (size_t)ncols * 4));
    HIPTRY(hipMemcpyAsync(d_map, map.data(), (size_t)ncols * 4, hipMemcpyHostToDevice, st));
    if (ldsrow) {
      hipLaunchKernelGGL((colperm_gather_kernel<true>), dim3((unsigned)rows), dim3(EC_THREADS), (size_t)width * 8, st, A, stride, width, lo / 64, hi / 64 + 1,
                         ncols, d_map, nullptr);
    } else {
      const int64_t chunk = 4096;  // rows per copy
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_copy), (size_t)(rows < chunk ? rows : chunk) * width * 8));
      for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t n = rows - r0 < chunk ? rows - r0 : chunk;
        HIPTRY(hipMemcpy2DAsync(d_copy, (size_t)width * 8, A + r0 * stride, (size_t)stride * 8, (size_t)width * 8, (size_t)n, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL((colperm_gather_kernel<false>), dim3((unsigned)n), dim3(EC_THREADS), 0, st, A + r0 * stride, stride, width, lo / 64, hi / 64 + 1,
                           ncols, d_map, d_copy);
      }
    }
    HIPTRY(hipGetLastError());
    HIPTRY(hipStreamSynchronize(st));
    return 0;
  };
  const int rc = run();
  if (d_map) (void)hipFree(d_map);
  if (d_copy) (void)hipFree(d_copy);
  return rc;
}

}  // namespace

extern "C" {

// A (nrows x ncols on the device) <- A * P (trans == 0) or A * P^T (trans != 0): the column transpositions (i, P[i]),
// i < length, descending resp. ascending, on every row (mzd_apply_p_right / mzd_apply_p_right_trans, mzp.c:193-260).
// P: HOST array.  Blocking.
int m4ri_amd_apply_p_right_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, const int32_t *P, int64_t length, int trans, void *stream) {
  if (nrows < 0 || ncols < 0 || length < 0 || !P) return (int)hipErrorInvalidValue;
  if (length > ncols) length = ncols;
  for (int64_t i = 0; i < length; ++i)
    if (P[i] < 0 || P[i] >= ncols) return (int)hipErrorInvalidValue;
  if (nrows == 0 || ncols == 0) return 0;
  std::vector<uint32_t> map;
  int64_t lo, hi;
  build_map(map, P, length, ncols, trans != 0, &lo, &hi);
  return apply_map(A, stride, nrows, ncols, map, lo, hi, (hipStream_t)stream);
}

// (Reduced) row echelon form of the device matrix in place; *rank_out = the rank.  Blocking.
int m4ri_amd_echelonize_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, int full, int32_t *rank_out, void *stream) {
  if (nrows < 0 || ncols < 0 || !rank_out) return (int)hipErrorInvalidValue;
  *rank_out = 0;
  if (nrows == 0 || ncols == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t width = words_of(ncols);
  std::vector<int32_t> P((size_t)nrows), Q((size_t)ncols);
  int32_t rank = 0;
  if (!full) {
    if (int rc = m4ri_amd_ple_dev(A, stride, nrows, ncols, P.data(), Q.data(), &rank, 0, st)) return rc;
    int32_t *d_Q = nullptr;
    if (rank > 0) {
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_Q), (size_t)rank * 4));
      hipError_t e = hipMemcpyAsync(d_Q, Q.data(), (size_t)rank * 4, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) { (void)hipFree(d_Q); return (int)e; }
    }
    hipLaunchKernelGGL(echelon_from_ple_kernel, dim3((unsigned)nrows), dim3(EC_THREADS), 0, st, A, stride, width, nrows, (int)rank, d_Q);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_Q) (void)hipFree(d_Q);
    *rank_out = rank;
    return (int)e;
  }
  if (int rc = m4ri_amd_pluq_dev(A, stride, nrows, ncols, P.data(), Q.data(), &rank, 0, st)) return rc;
  *rank_out = rank;
  if (rank > 0 && rank != ncols) {  // echelonform.c:66-103
    const int64_t wr = words_of(rank);
    word *U = nullptr;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&U), (size_t)rank * wr * 8));
    int rc = (int)hipMemcpy2DAsync(U, (size_t)wr * 8, A, (size_t)stride * 8, (size_t)wr * 8, (size_t)rank, hipMemcpyDeviceToDevice, st);
    if (!rc) rc = m4ri_amd_mask_tail_dev(U, wr, rank, rank, st);  // the copy's last word also took the first columns of B along
    const int64_t w0 = rank / 64;
    if (!rc) rc = m4ri_amd_trsm_upper_left_dev(U, wr, A + w0, stride, rank, ncols - w0 * 64, 0, st);
    if (!rc) rc = (int)hipStreamSynchronize(st);
    (void)hipFree(U);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(echelon_identity_kernel, dim3((unsigned)nrows), dim3(EC_THREADS), 0, st, A, stride, width, nrows, (int)rank);
  HIPTRY(hipGetLastError());
  if (rank > 0) {  // :108-112
    std::vector<uint32_t> map;
    int64_t lo, hi;
    build_map(map, Q.data(), ncols, ncols, false, &lo, &hi);
    if (int rc = apply_map(A, stride, rank, ncols, map, lo, hi, st)) return rc;
  }
  return (int)hipStreamSynchronize(st);
}

}  // extern "C"
