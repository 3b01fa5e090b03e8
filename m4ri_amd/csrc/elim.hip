// elim.hip -- the table primitives of M4RI's elimination routines on the device:
//   mzd_make_table          /root/reference m4ri/brilliantrussian.c:163-211   2^k Gray-code combinations of k rows
//   mzd_process_rows{,2..6} m4ri/brilliantrussian.c:213-601                   M[r, block:] ^= T0[L0[bits0]] ^ ... ^ T5[L5[bits5]]
// They share the multiply leaf's primitive (tables of row combinations, one lookup per k bits) but not its
// shape: the "inner dimension" is a single k <= 48 bit strip and the matrix is read-modify-written once, so
// these are streaming kernels bounded by HBM, not by LDS (the tables -- caller-supplied, up to 6 x 2^8 rows --
// are read through L2).  Two launches per call: the strip of every row is decoded into table row numbers first
// (the strip itself lies inside the words the update rewrites), then every (row, 16-byte chunk) is updated.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

namespace {

constexpr int EL_THREADS = 256;

struct ProcArgs {
  const word *T[6];
  int64_t t_stride[6];
  const int32_t *L[6];
  int32_t kbits[6];
  int32_t ntables;
};

// bits [col, col + n) of a row, column col at bit 0 (mzd.h:892-901 mzd_read_bits); n <= 64
__device__ __forceinline__ word read_bits(const word *row, int64_t col, int n) {
  const int spot    = (int)(col & 63);
  const int64_t blk = col >> 6;
  const int spill   = spot + n - 64;
  word t = spill <= 0 ? (row[blk] << -spill) : ((row[blk + 1] << (64 - spill)) | (row[blk] >> spill));
  return t >> (64 - n);
}

__global__ __launch_bounds__(EL_THREADS) void decode_strip_kernel(const word *__restrict__ M, int64_t stride, int64_t startrow, int64_t rows,
                                                                 int64_t startcol, int k, ProcArgs a, int32_t *__restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * EL_THREADS + threadIdx.x;
  if (i >= rows) return;
  word bits = read_bits(M + (startrow + i) * stride, startcol, k);
  for (int t = 0; t < a.ntables; ++t) {
    const int kb  = a.kbits[t];
    const word bm = kb >= 64 ? ~(word)0 : (((word)1 << kb) - 1);
    idx[i * 6 + t] = a.L[t][bits & bm];
    bits = kb >= 64 ? 0 : (bits >> kb);
  }
}

// With several tables the lookups outweigh the matrix itself (six 8-bit tables over 65536 columns: 12 MB of table rows,
// six reads per chunk), and a workgroup that walks whole rows pulls all of it through its XCD's 4 MB L2.  Here the
// workgroups of XCD x (blockIdx % 8: the dispatcher deals them round-robin) keep to the x-th eighth of the columns, so
// each L2 holds its own eighth of the tables: six tables 2.5 -> 2.9 TB/s over 65536 columns, 2.55 -> 3.3 over 32768; two
// tables 4.7 -> 4.8 / 4.9 -> 5.2.  (Walking the eighth in narrower strips does not add to that, and the tables' slices in
// LDS -- which caps the tile at 64 bytes of a row for six 8-bit tables -- are slower than L2: both measured.)
template <typename V>
__global__ __launch_bounds__(EL_THREADS) void apply_tables_slab_kernel(V *__restrict__ M, int64_t stride, int64_t startrow, int64_t rows, int64_t block,
                                                                      int64_t wide, ProcArgs a, const int32_t *__restrict__ idx) {
  const int xcd      = blockIdx.x & 7;
  const int64_t nb   = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int64_t s0   = xcd * wide / 8, sw = (xcd + 1) * wide / 8 - s0;
  const int64_t total = rows * sw;
  for (int64_t e = j * EL_THREADS + threadIdx.x; e < total; e += nb * EL_THREADS) {
    const int64_t i = e / sw, w = block + s0 + (e - i * sw);
    V x = M[(startrow + i) * stride + w];
    for (int t = 0; t < a.ntables; ++t)
      x ^= reinterpret_cast<const V *>(a.T[t])[(int64_t)idx[i * 6 + t] * a.t_stride[t] + w];
    M[(startrow + i) * stride + w] = x;
  }
}

template <typename V>
__global__ __launch_bounds__(EL_THREADS) void apply_tables_kernel(V *__restrict__ M, int64_t stride, int64_t startrow, int64_t rows, int64_t block,
                                                                 int64_t wide, ProcArgs a, const int32_t *__restrict__ idx) {
  const int64_t total = rows * wide;
  const int64_t gstr  = (int64_t)gridDim.x * EL_THREADS;
  for (int64_t e = (int64_t)blockIdx.x * EL_THREADS + threadIdx.x; e < total; e += gstr) {
    const int64_t i = e / wide, w = block + (e - i * wide);
    V x = M[(startrow + i) * stride + w];
    for (int t = 0; t < a.ntables; ++t)
      x ^= reinterpret_cast<const V *>(a.T[t])[(int64_t)idx[i * 6 + t] * a.t_stride[t] + w];
    M[(startrow + i) * stride + w] = x;
  }
}

// T[i] for i = 1 .. 2^k - 1, words home .. width - 1.  gray[i] = i ^ (i >> 1): bit b <-> row r + b.
// jstar[i]: the last step j <= i whose row does not exist (its T row keeps the caller's content, brilliantrussian.c:181);
// 0 if none (then the chain starts from the caller's T[0]).
__global__ __launch_bounds__(EL_THREADS) void make_table_kernel(const word *__restrict__ M, int64_t m_stride, int64_t m_rows, int64_t r, int k,
                                                               const word *__restrict__ Tin, word *__restrict__ Tout, int64_t t_stride,
                                                               int64_t home, int64_t width, word mask_begin, word mask_end,
                                                               const int32_t *__restrict__ jstar) {
  const int64_t wide  = width - home;
  const int64_t total = (((int64_t)1 << k) - 1) * wide;
  const int64_t gstr  = (int64_t)gridDim.x * EL_THREADS;
  for (int64_t e = (int64_t)blockIdx.x * EL_THREADS + threadIdx.x; e < total; e += gstr) {
    const int64_t i = 1 + e / wide, w = home + (e % wide);
    const int64_t js = jstar[i];
    if (js == i) { Tout[i * t_stride + w] = Tin[i * t_stride + w]; continue; }
    word x = Tin[js * t_stride + w];
    unsigned g = (unsigned)(i ^ (i >> 1)) ^ (unsigned)(js ^ (js >> 1));
    while (g) {
      const int b = __builtin_ctz(g);
      g &= g - 1;
      if (r + b < m_rows) x ^= M[(r + b) * m_stride + w];
    }
    if (w == home) x &= mask_begin;       // brilliantrussian.c:185
    if (w == width - 1) x &= mask_end;    // :206 (a one-word table row takes both, :167-168)
    Tout[i * t_stride + w] = x;
  }
}

unsigned grid_for(int64_t total) {
  int64_t g = (total + EL_THREADS - 1) / EL_THREADS;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

// M[r, startcol/64 .. width) ^= T0[L0[bits0]] ^ ... for rows [startrow, stoprow); bits_t = the t-th group of
// kbits[t] bits of the k-bit strip starting at column startcol (lowest group first), exactly how
// mzd_process_rowsN splits k (brilliantrussian.c:357-361, :394-398, ...).  T_t, L_t are device pointers
// (L_t: 2^kbits[t] table row numbers); idx_scratch: 6 * (stoprow - startrow) int32.  Asynchronous on `stream`.
int m4ri_amd_process_rows_dev(word *M, int64_t stride, int64_t width, int64_t startrow, int64_t stoprow, int64_t startcol, int ntables,
                              const int32_t *kbits, const word *const *T, const int64_t *t_stride, const int32_t *const *L,
                              int32_t *idx_scratch, void *stream) {
  if (ntables < 1 || ntables > 6 || startrow < 0 || stoprow < startrow || startcol < 0) return (int)hipErrorInvalidValue;
  const int64_t rows = stoprow - startrow, block = startcol / 64, wide = width - block;
  if (rows == 0 || wide <= 0) return 0;
  ProcArgs a{};
  int k = 0;
  bool vec = (reinterpret_cast<uintptr_t>(M) % 16 == 0) && stride % 2 == 0 && block % 2 == 0 && wide % 2 == 0;
  for (int t = 0; t < ntables; ++t) {
    a.T[t] = T[t]; a.t_stride[t] = t_stride[t]; a.L[t] = L[t]; a.kbits[t] = kbits[t];
    k += kbits[t];
    vec = vec && (reinterpret_cast<uintptr_t>(T[t]) % 16 == 0) && t_stride[t] % 2 == 0;
  }
  a.ntables = ntables;
  if (k < 1 || k > 64) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(decode_strip_kernel, dim3((unsigned)((rows + EL_THREADS - 1) / EL_THREADS)), dim3(EL_THREADS), 0, st, M, stride, startrow, rows,
                     startcol, k, a, idx_scratch);
  if (vec) {
    typedef unsigned long long __attribute__((ext_vector_type(2))) word2;
    ProcArgs h = a;
    for (int t = 0; t < ntables; ++t) h.t_stride[t] /= 2;
    static const int slab_env = getenv("M4RI_AMD_ELIM_SLABS") ? atoi(getenv("M4RI_AMD_ELIM_SLABS")) : -1;
    const bool slabs = slab_env >= 0 ? slab_env != 0 : (ntables >= 2 && wide / 2 >= 64);
    if (slabs)
      hipLaunchKernelGGL((apply_tables_slab_kernel<word2>), dim3((grid_for(rows * (wide / 2)) + 7) / 8 * 8), dim3(EL_THREADS), 0, st,
                         reinterpret_cast<word2 *>(M), stride / 2, startrow, rows, block / 2, wide / 2, h, idx_scratch);
    else
      hipLaunchKernelGGL((apply_tables_kernel<word2>), dim3(grid_for(rows * (wide / 2))), dim3(EL_THREADS), 0, st, reinterpret_cast<word2 *>(M),
                         stride / 2, startrow, rows, block / 2, wide / 2, h, idx_scratch);
  } else {
    hipLaunchKernelGGL((apply_tables_kernel<word>), dim3(grid_for(rows * wide)), dim3(EL_THREADS), 0, st, M, stride, startrow, rows, block, wide, a,
                       idx_scratch);
  }
  return (int)hipGetLastError();
}

// The device twin of mzd_make_table: T[i] (i = 1 .. 2^k - 1, words c/64 .. width - 1) from rows r .. r+k-1 of M.
// Tin: the table's previous content (rows whose source row does not exist keep it); jstar: 2^k int32 on the
// device, see make_table_kernel.  Tin may equal Tout.
int m4ri_amd_make_table_dev(const word *M, int64_t m_stride, int64_t m_rows, int64_t ncols, int64_t r, int64_t c, int k, const word *Tin,
                            word *Tout, int64_t t_stride, const int32_t *jstar, void *stream) {
  if (k < 1 || k > 16 || ncols <= 0) return (int)hipErrorInvalidValue;
  const int64_t width = words_of(ncols), home = c / 64;
  if (home >= width) return 0;
  const word mask_end   = (ncols % 64) ? ((~(word)0) >> (64 - ncols % 64)) : ~(word)0;
  const word pure_begin = ~(word)0 << (c % 64);
  const word mask_begin = (width - home != 1) ? pure_begin : (pure_begin & mask_end);
  hipLaunchKernelGGL(make_table_kernel, dim3(grid_for((((int64_t)1 << k) - 1) * (width - home))), dim3(EL_THREADS), 0, (hipStream_t)stream, M, m_stride,
                     m_rows, r, k, Tin, Tout, t_stride, home, width, mask_begin, mask_end, jstar);
  return (int)hipGetLastError();
}

}  // extern "C"
