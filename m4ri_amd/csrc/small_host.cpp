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
// Own code.  Generation 2 (round 6; the first one was the textbook loop: one table per 4 or 8 inner bits, one pass over C per table,
// 1.2 ... 2 x SLOWER than the reference between 96^3 and 400^3 -- tests/small_host_timing.c):
//   * the columns of C in blocks of at most 8 words whose width is a compile-time constant: an accumulator row lives in registers,
//     the word loops vectorise;
//   * a whole 64-bit word of A per pass (full words with the chunk layout as compile-time constants: offsets, masks and table positions
//     fold into the instructions -- from small arrays they cost three more loads per lookup): its bits are cut into c chunks of K or K + 1 bits (K from the number of rows:
//     the minimum of (2^K + m) / K row operations per inner bit), ONE table of XOR combinations of the rows of B per chunk, built by
//     doubling, and every row of A adds its c table entries to its accumulator row in ONE read-modify-write (what the reference gets
//     from its eight tables per pass, brilliantrussian.c:1111-1154);
//   * a handful of rows (at most 12, or 24 when a row of C has more than one word): every set bit of A adds one row of B, no tables;
//   * windows (row stride larger than the width, dirty bits beyond the last column) are handled by masking what is read -- only
//     the last word of a row needs it -- and merging what is written under the column mask (mzd.h:117-123).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/m4ri_amd.h"

namespace {

constexpr uint8_t FLAG_WINDOW = 0x4;  // mzd.h:150
constexpr int MIN_K = 3, MAX_K = 8;
constexpr int MAX_CHUNKS = 64 / MIN_K + 1;
constexpr int BLOCK_W = 8;            // words of C per column block
// at most this many rows: no tables (measured against the tables on 9 ... 64 rows: rows of one word 21.5 against 23.0 us at 12 rows and
// 29.5 against 26.4 at 16; wider rows 10.5 against 14.4 us at 16 rows, 15.8 against 16.8 at 24, 21.2 against 19.7 at 32)
inline int bitwise_rows(wi_t wn) { return wn >= 2 ? 24 : 12; }
constexpr size_t KEEP_WORDS = (size_t)1 << 20;

struct Scratch {
  std::vector<word> acc, table;
};
thread_local Scratch t_scratch;

// bits per table for m rows: the K that minimises (2^K + m) / K, the row operations per inner bit (table building + lookups)
int pick_k(int64_t m) {
  int best = MIN_K;
  double cost = 1e300;
  for (int K = MIN_K; K <= MAX_K; ++K) {
    const double c = ((double)(1 << K) + (double)m) / (double)K;
    if (c <= cost) { cost = c; best = K; }  // (ties go to the larger K: fewer tables to set up)
  }
  return best;
}

// The lookups of one FULL word of A (64 inner bits) with the chunk layout as compile-time constants -- offsets, masks and table positions
// fold into the instructions, one load per looked-up word (from small arrays they cost three more loads per lookup: on blocks of one
// word the load ports were the bound, 13.4 -> 8 us at 16 x 4096 x 16).  Same layout as the builder below: c = ceil(64 / K) chunks, the
// first 64 % c of them one bit longer.
template <int W, int K, int U = 0, int O = 0, int T = 0>
inline void lookups_full(word (&v)[W], word a, const word *table) {
  constexpr int c = (64 + K - 1) / K, base = 64 / c, extra = 64 % c;
  if constexpr (U < c) {
    constexpr int s = base + (U < extra ? 1 : 0);
    const word *e   = table + (size_t)T * W + (size_t)((a >> O) & (((word)1 << s) - 1)) * W;
    for (int k = 0; k < W; ++k) v[k] ^= e[k];
    lookups_full<W, K, U + 1, O + s, T + (1 << s)>(v, a, table);
  }
}
// ... and the tables of a full word with the same constants: chunk U = rows O .. O + s of the word's 64 rows of B, by doubling
template <int W, int K, int U = 0, int O = 0, int T = 0>
inline void build_full(word *table, const word *b0, int64_t b_stride, word bmask) {
  constexpr int c = (64 + K - 1) / K, base = 64 / c, extra = 64 % c;
  if constexpr (U < c) {
    constexpr int s = base + (U < extra ? 1 : 0);
    word *t = table + (size_t)T * W;
    for (int k = 0; k < W; ++k) t[k] = 0;
    for (int b = 0; b < s; ++b) {
      const word *brow = b0 + (int64_t)(O + b) * b_stride;
      word r[W];
      for (int k = 0; k < W; ++k) r[k] = brow[k];
      r[W - 1] &= bmask;
      const int half = 1 << b;
      word *dst = t + (size_t)half * W;
      for (int x = 0; x < half; ++x)
        for (int k = 0; k < W; ++k) dst[x * W + k] = t[x * W + k] ^ r[k];
    }
    build_full<W, K, U + 1, O + s, T + (1 << s)>(table, b0, b_stride, bmask);
  }
}
template <int W, int K>
void gather_full(word *acc, int64_t acc_stride, const word *ap, int64_t a_stride, rci_t m, const word *table) {
  for (rci_t i = 0; i < m; ++i) {
    const word a = ap[(int64_t)i * a_stride];
    if (!a) continue;
    word *dst = acc + (int64_t)i * acc_stride;
    word v[W];
    for (int k = 0; k < W; ++k) v[k] = dst[k];
    lookups_full<W, K>(v, a, table);
    for (int k = 0; k < W; ++k) dst[k] = v[k];
  }
}

// One column block of W words (first word k0 of the rows of B and of the accumulator; `bmask` = mask of the block's last word of B):
// acc[i][0 .. W) ^= sum over the inner dimension of A[i][j] * B[j][k0 .. k0 + W)
template <int W>
void block_tables(word *acc, int64_t acc_stride, const mzd_t *A, const mzd_t *B, wi_t k0, word bmask, int K, word *table) {
  const rci_t m = A->nrows, l = A->ncols;
  const wi_t wl = A->width;
  for (wi_t q = 0; q < wl; ++q) {
    const int bits = (l - (rci_t)q * 64) < 64 ? (int)(l - (rci_t)q * 64) : 64;
    if (bits == 64) {  // a full word of A (no bits beyond the last column: no mask): everything with compile-time chunk layouts
      const word *b0 = B->data + (int64_t)q * 64 * B->rowstride + k0, *ap = A->data + q;
#define FULL_(KK) build_full<W, KK>(table, b0, B->rowstride, bmask); gather_full<W, KK>(acc, acc_stride, ap, A->rowstride, m, table)
      switch (K) {
        case 3: FULL_(3); break;
        case 4: FULL_(4); break;
        case 5: FULL_(5); break;
        case 6: FULL_(6); break;
        case 7: FULL_(7); break;
        default: FULL_(8); break;
      }
#undef FULL_
      continue;
    }
    const int c    = (bits + K - 1) / K;  // chunks of this word: `extra` of them one bit longer than the others
    const int base = bits / c, extra = bits % c;
    int off[MAX_CHUNKS];
    word msk[MAX_CHUNKS];
    const word *tab[MAX_CHUNKS];
    word *t = table;
    for (int u = 0, o = 0; u < c; ++u) {
      const int s = base + (u < extra ? 1 : 0);
      off[u] = o;
      msk[u] = ((word)1 << s) - 1;
      tab[u] = t;
      // t[x] = XOR of the rows 64 q + o + b of B with bit b of x set, by doubling
      for (int k = 0; k < W; ++k) t[k] = 0;
      for (int b = 0; b < s; ++b) {
        const word *brow = B->data + (int64_t)((rci_t)q * 64 + o + b) * B->rowstride + k0;
        word r[W];
        for (int k = 0; k < W; ++k) r[k] = brow[k];
        r[W - 1] &= bmask;
        const size_t half = (size_t)1 << b;
        word *dst = t + half * W;
        for (size_t x = 0; x < half; ++x)
          for (int k = 0; k < W; ++k) dst[x * W + k] = t[x * W + k] ^ r[k];
      }
      t += ((size_t)1 << s) * W;
      o += s;
    }
    const word *ap = A->data + q;
    // the last, partial word of A: per-chunk offsets, masks and table pointers from small arrays (the lookups of a row stay independent
    // of each other; shifting the word along chunk by chunk instead makes them a dependent chain: 10 ... 25 % slower on narrow blocks)
    const word amask = A->high_bitmask;
    for (rci_t i = 0; i < m; ++i) {
      const word a = ap[(int64_t)i * A->rowstride] & amask;
      if (!a) continue;
      word *dst = acc + (int64_t)i * acc_stride;
      word v[W];
      for (int k = 0; k < W; ++k) v[k] = dst[k];
      for (int u = 0; u < c; ++u) {
        const word *e = tab[u] + (size_t)((a >> off[u]) & msk[u]) * W;
        for (int k = 0; k < W; ++k) v[k] ^= e[k];
      }
      for (int k = 0; k < W; ++k) dst[k] = v[k];
    }
  }
}

// the same for a handful of rows: every set bit of A adds one row of B
template <int W>
void block_bitwise(word *acc, int64_t acc_stride, const mzd_t *A, const mzd_t *B, wi_t k0, word bmask) {
  const rci_t m = A->nrows;
  const wi_t wl = A->width;
  for (rci_t i = 0; i < m; ++i) {
    word v[W];
    word *dst = acc + (int64_t)i * acc_stride;
    for (int k = 0; k < W; ++k) v[k] = 0;
    const word *ap = A->data + (int64_t)i * A->rowstride;
    for (wi_t q = 0; q < wl; ++q) {
      word a = ap[q] & (q == wl - 1 ? A->high_bitmask : ~(word)0);
      const word *bq = B->data + (int64_t)q * 64 * B->rowstride + k0;
      for (; a; a &= a - 1) {
        const word *brow = bq + (int64_t)__builtin_ctzll(a) * B->rowstride;
        for (int k = 0; k < W; ++k) v[k] ^= brow[k];
      }
    }
    v[W - 1] &= bmask;  // (XOR and the mask commute: masking the sum once is masking every row)
    for (int k = 0; k < W; ++k) dst[k] ^= v[k];
  }
}

template <int W>
void block(word *acc, int64_t acc_stride, const mzd_t *A, const mzd_t *B, wi_t k0, word bmask, int K, word *table) {
  if (K == 0) block_bitwise<W>(acc, acc_stride, A, B, k0, bmask);
  else block_tables<W>(acc, acc_stride, A, B, k0, bmask, K, table);
}

// row operations (table entries built + lookups) of the words of A that hold `bits` inner bits, for m rows
double word_row_ops(int bits, int K, int64_t m) {
  const int c = (bits + K - 1) / K, base = bits / c, extra = bits % c;
  return (double)extra * (double)(2 << base) + (double)(c - extra) * (double)(1 << base) + (double)m * (double)c;
}

}  // namespace

// What the routine below costs, in word operations: its row operations (table entries + lookups, or one per set bit of A for a
// handful of rows) times the words of a row of C plus 1.1 per column block (a row operation on a block of W words measures 0.07 (W + 1.1) ns
// on the GPU boxes' host cores: 512^3 30.0 us, 256^3 5.2 us, 2048 x 2048 x 16 95 us -- profiles/r06_small_products_host_routine.log),
// plus the accumulator's way in and out.  The size switch of the entry points (mzd_api.hip: small_product_wanted) bounds it.
extern "C" double gf2_small_host_cost(int64_t m, int64_t l, int64_t n) {
  if (m <= 0 || n <= 0) return 0.0;
  const double wn = (double)((n + 63) / 64), nblocks = (double)(((n + 63) / 64 + BLOCK_W - 1) / BLOCK_W);
  double row_ops;
  if (m <= bitwise_rows((wi_t)((n + 63) / 64))) {
    row_ops = 0.5 * (double)m * (double)l;
  } else {
    const int K = pick_k(m);
    row_ops     = (double)(l / 64) * word_row_ops(64, K, m) + (l % 64 ? word_row_ops((int)(l % 64), K, m) : 0.0);
  }
  return row_ops * (wn + 1.1 * nblocks) + 2.0 * (double)m * wn;
}

extern "C" int m4ri_amd_small_mul_host(mzd_t *C, const mzd_t *A, const mzd_t *B, int add) {
  if (!C || !A || !B || A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols) return -1;
  const rci_t m = A->nrows, l = A->ncols, n = B->ncols;
  if (m == 0 || n == 0) return 0;
  const wi_t wn = C->width;
  Scratch &S = t_scratch;
  if (S.acc.size() < (size_t)m * (size_t)wn) S.acc.resize((size_t)m * (size_t)wn);
  word *acc = S.acc.data();
  if (add) {
    for (rci_t i = 0; i < m; ++i) {
      const word *c = C->data + (int64_t)i * C->rowstride;
      word *a       = acc + (size_t)i * wn;
      for (wi_t k = 0; k < wn; ++k) a[k] = c[k];
      a[wn - 1] &= C->high_bitmask;
    }
  } else {
    std::memset(acc, 0, (size_t)m * (size_t)wn * 8);
  }
  if (l > 0) {
    const int K = m <= bitwise_rows(wn) ? 0 : pick_k(m);  // 0: no tables
    // the tables of one word of A: at most 64 / K + 1 chunks of at most K + 1 bits
    const size_t tw = K ? (size_t)(64 / K + 1) * ((size_t)2 << K) * BLOCK_W : 0;
    if (S.table.size() < tw) S.table.resize(tw);
    word *table = S.table.data();
    // column blocks of (almost) equal width: 9 words are 5 + 4, not 8 + 1 -- a one-word block costs as many lookups as a full one
    const wi_t nblocks = (wn + BLOCK_W - 1) / BLOCK_W, wbase = wn / nblocks, wextra = wn % nblocks;
    wi_t k0 = 0;
    for (wi_t blk = 0; blk < nblocks; ++blk) {
      const int w      = (int)(wbase + (blk < wextra ? 1 : 0));
      const word bmask = (k0 + w == wn) ? B->high_bitmask : ~(word)0;
      word *a0         = acc + k0;
      switch (w) {
        case 1: block<1>(a0, wn, A, B, k0, bmask, K, table); break;
        case 2: block<2>(a0, wn, A, B, k0, bmask, K, table); break;
        case 3: block<3>(a0, wn, A, B, k0, bmask, K, table); break;
        case 4: block<4>(a0, wn, A, B, k0, bmask, K, table); break;
        case 5: block<5>(a0, wn, A, B, k0, bmask, K, table); break;
        case 6: block<6>(a0, wn, A, B, k0, bmask, K, table); break;
        case 7: block<7>(a0, wn, A, B, k0, bmask, K, table); break;
        default: block<8>(a0, wn, A, B, k0, bmask, K, table); break;
      }
      k0 += w;
    }
  }
  // the result, under the column mask: bits of C's last word beyond its columns keep their value in a window and end up zero otherwise
  const bool window = (C->flags & FLAG_WINDOW) != 0;
  for (rci_t i = 0; i < m; ++i) {
    word *c       = C->data + (int64_t)i * C->rowstride;
    const word *a = acc + (size_t)i * wn;
    for (wi_t k = 0; k + 1 < wn; ++k) c[k] = a[k];
    const word v = a[wn - 1] & C->high_bitmask;
    c[wn - 1]    = window ? ((c[wn - 1] & ~C->high_bitmask) | v) : v;
  }
  if (S.acc.size() > KEEP_WORDS) std::vector<word>().swap(S.acc);  // a thread keeps at most 8 MiB of accumulator between calls
  return 0;
}
