// small_host.cpp -- products too small for the GPU to win: a host Method-of-Four-Russians of this library's own.
//
// The reference switches algorithm by size inside the very function this library replaces (_mzd_mul_m4rm falls to
// mzd_mul_naive / mzd_addmul_naive below 54 columns or 16 rows, /root/reference m4ri/brilliantrussian.c:1063-1068,
// m4ri/mzd.c:1141-1172).  The GPU's own switch sits higher: every call through the host entry points pays 28 ... 80 us for the
// upload, the launches and the download whatever its size, which a product of a few hundred rows does not repay
// (profiles/r04_crossover_cpu_gpu.log).  Below m4ri_amd_set_small_product_threshold (m * l * n bit operations) the entry points of
// mzd_api.hip therefore compute the product here, on the calling thread -- so that a program under LD_PRELOAD never gets slower by
// interposing.  This is an algorithm choice made on an initialised device, never a fallback: a HIP failure still aborts, and the
// library still refuses to work without its GPU (the entry points bind the device before they look at the size).
//
// Own code, the textbook algorithm: the inner dimension in groups of 4 or 8 rows of B (by the number of rows of A), one table of
// their XOR combinations per group (built by doubling), every row of A looks its nibble / byte up and adds the entry to an
// accumulator row; fewer than 16 rows add B's rows bit by bit; windows (row stride
// larger than the width, dirty bits beyond the last column) are handled by masking what is read and merging what is written
// under the column mask (mzd.h:117-123).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/m4ri_amd.h"

namespace {

constexpr uint8_t FLAG_WINDOW = 0x4;  // mzd.h:150

// row i of M, word k, with the bits beyond the last column cleared
inline word rd(const mzd_t *M, rci_t i, wi_t k) {
  const word v = M->data[(int64_t)i * M->rowstride + k];
  return k == M->width - 1 ? (v & M->high_bitmask) : v;
}

}  // namespace

extern "C" int m4ri_amd_small_mul_host(mzd_t *C, const mzd_t *A, const mzd_t *B, int add) {
  if (!C || !A || !B || A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols) return -1;
  const rci_t m = A->nrows, l = A->ncols, n = B->ncols;
  if (m == 0 || n == 0) return 0;
  const wi_t wn = C->width, wl = A->width;
  std::vector<word> acc((size_t)m * (size_t)wn, 0);
  if (add)
    for (rci_t i = 0; i < m; ++i)
      for (wi_t k = 0; k < wn; ++k) acc[(size_t)i * wn + k] = rd(C, i, k);
  if (l > 0) {
    if (m < 16) {
      // a handful of rows: tables would cost more than they save -- every set bit of A adds one row of B
      for (rci_t i = 0; i < m; ++i) {
        word *a = &acc[(size_t)i * wn];
        for (wi_t q = 0; q < wl; ++q)
          for (word bitsleft = rd(A, i, q); bitsleft; bitsleft &= bitsleft - 1) {
            const rci_t j = (rci_t)(q * 64 + __builtin_ctzll(bitsleft));
            for (wi_t k = 0; k < wn; ++k) a[k] ^= rd(B, j, k);
          }
      }
    } else {
      // groups of K rows of B, one table of their 2^K XOR combinations per group; per inner bit that costs (2^K + m) / K row
      // operations: K = 4 below 224 rows, K = 8 above (both divide 64: a group never straddles a word of A)
      const int K = m < 224 ? 4 : 8;
      std::vector<word> table(((size_t)1 << K) * (size_t)wn);
      for (rci_t g0 = 0; g0 < l; g0 += K) {
        const int bits = (l - g0) < K ? (int)(l - g0) : K;
        // table[x] = XOR of the rows g0 + b of B with bit b of x set, by doubling: the second half of every step is the first
        // half plus one more row
        std::memset(table.data(), 0, (size_t)wn * 8);
        for (int b = 0; b < bits; ++b) {
          const size_t half = (size_t)1 << b;
          for (size_t x = 0; x < half; ++x) {
            const word *src = &table[x * (size_t)wn];
            word *dst       = &table[(x + half) * (size_t)wn];
            for (wi_t k = 0; k < wn; ++k) dst[k] = src[k] ^ rd(B, g0 + b, k);
          }
        }
        const wi_t aw      = g0 / 64;
        const int shift    = g0 % 64;
        const word lowbits = ((word)1 << bits) - 1;
        for (rci_t i = 0; i < m; ++i) {
          const size_t x = (size_t)((rd(A, i, aw) >> shift) & lowbits);
          if (!x) continue;
          const word *t = &table[x * (size_t)wn];
          word *a       = &acc[(size_t)i * wn];
          for (wi_t k = 0; k < wn; ++k) a[k] ^= t[k];
        }
      }
    }
  }
  // the result, under the column mask: bits of C's last word beyond its columns keep their value in a window and end up zero otherwise
  const bool window = (C->flags & FLAG_WINDOW) != 0;
  for (rci_t i = 0; i < m; ++i) {
    word *c = C->data + (int64_t)i * C->rowstride;
    for (wi_t k = 0; k + 1 < wn; ++k) c[k] = acc[(size_t)i * wn + k];
    const word v = acc[(size_t)i * wn + wn - 1] & C->high_bitmask;
    c[wn - 1]    = window ? ((c[wn - 1] & ~C->high_bitmask) | v) : v;
  }
  return 0;
}
