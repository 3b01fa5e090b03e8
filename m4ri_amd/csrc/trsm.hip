// trsm.hip -- triangular solves with a matrix right-hand side on the device: B <- L^-1 B (L unit lower
// triangular) and B <- U^-1 B (U unit upper triangular), the callers of the multiply path one step up
// (SURVEY.md 8f rank 3).
//
// Reference interfaces replaced (same results -- the solution of a triangular system is unique, so any
// correct schedule is bit-identical):
//   _mzd_trsm_lower_left / _mzd_trsm_upper_left               /root/reference m4ri/triangular.c:406-455, :467-514
//   _mzd_trsm_lower_left_russian / _mzd_trsm_upper_left_russian  m4ri/triangular_russian.c:206-330, :50-170
// The reference recurses on halves with mzd_addmul for the update (triangular.c:437-441, :501-505), solves
// blocks of <= 2048 rows by a Four-Russians sweep (tables of the rows just solved, one pass over the rows
// below per 8k pivot rows) and blocks of <= 64 rows by direct substitution.  Here the recursion on halves runs on
// the engine's own product (addmul on device-resident views) down to blocks of 512 rows; those are solved by
// multiplying with the block's inverse, all diagonal blocks inverted up front in one launch (see below).  Systems
// of <= 64 rows (the PLE's pivot rows) take one small kernel: a thread owns one 64-bit word column of B, keeps its
// 64 words in registers and substitutes; the rows of the triangle are wave-uniform (scalar loads).
// The diagonal is never read (taken as 1) and neither is the other triangle, exactly like the reference.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdlib.h>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

namespace {

#define HIPTRY(expr)                                  \
  do {                                                \
    hipError_t e_ = (hipError_t)(expr);               \
    if (e_ != hipSuccess) return (int)e_;             \
  } while (0)

constexpr int TRSM_THREADS = 64;

// B (mb x wn words, mb <= 64) <- T^-1 B in place; T's block sits in bits 0..mb-1 of the first word of its rows.
template <bool UPPER>
__global__ __launch_bounds__(TRSM_THREADS) void trsm_base_kernel(const word *__restrict__ T, int64_t t_stride, word *__restrict__ B,
                                                                  int64_t b_stride, int mb, int64_t wn, word last_mask) {
  const int64_t j = (int64_t)blockIdx.x * TRSM_THREADS + threadIdx.x;
  if (j >= wn) return;
  word x[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) x[i] = i < mb ? B[(int64_t)i * b_stride + j] : 0;
  if (!UPPER) {
#pragma unroll
    for (int i = 1; i < 64; ++i) {
      if (i < mb) {
        const word row = T[(int64_t)i * t_stride];  // wave-uniform: a scalar load, so the tests below are scalar branches
        word acc       = 0;                          // (half of the XORs are skipped outright instead of being masked)
#pragma unroll
        for (int k = 0; k < i; ++k)
          if ((row >> k) & 1) acc ^= x[k];
        x[i] ^= acc;
      }
    }
  } else {
#pragma unroll
    for (int i = 62; i >= 0; --i) {
      if (i < mb - 1) {
        const word row = T[(int64_t)i * t_stride];
        word acc       = 0;
#pragma unroll
        for (int k = i + 1; k < 64; ++k)
          if (k < mb && ((row >> k) & 1)) acc ^= x[k];
        x[i] ^= acc;
      }
    }
  }
  // the last word of a row is merged under the column mask (triangular.c:421, :482): bits beyond B's columns stay
  const bool last = (j == wn - 1) && last_mask != ~(word)0;
#pragma unroll
  for (int i = 0; i < 64; ++i)
    if (i < mb) {
      word *p = B + (int64_t)i * b_stride + j;
      *p      = last ? ((*p & ~last_mask) | (x[i] & last_mask)) : x[i];
    }
}

// ---- blocks of 512 rows through their inverses -------------------------------------------------------------------
// A product call costs ~65 us whatever its size (pack, leaf prologue, split reduction), so a recursion that bottoms out
// at 64 rows spends its time launching: 1023 products and 1024 base kernels for a 65536-row triangle.  The diagonal
// blocks of 512 rows are inverted up front instead -- all of them in ONE launch, a workgroup per block (the kernel
// below) -- and a block's solve becomes X = T_bb^-1 * B_b, one product; the recursion stops at 512 rows (127 + 128
// products).
#ifndef TRSM_TB
#define TRSM_TB 512
#endif
constexpr int TB = TRSM_TB;  // rows of a diagonal block (256 / 512 / 1024: 44.0 / 38.8 / 37.7 ms at 65536^2, 6.2 / 5.2 / 5.8 ms at 16384^2)

// The inverse of a block is built in LDS, in place, bottom-up (the first version let every thread substitute its own
// column through all 512 rows: 512 dependent steps, 0.32 ms for a block, 0.19 ms with the rows in LDS -- nothing when all
// blocks of a triangle are inverted in one launch, but the PLE inverts one block per panel, on its critical path):
//   1. wave b inverts the 64 x 64 diagonal block b: lane j substitutes column j in one register, and the ballot of the
//      64 lanes' new bits IS row i of the inverse, written over row i of the block (no later step reads it);
//   2. s = 64, 128, 256: every pair of finished s-blocks closes its off-diagonal block,  X10 = X11 * (L10 * X00)  for a
//      lower triangle,  X01 = X00 * (U01 * X11)  for an upper one -- two bit-matrix products per level, a thread per output
//      word, the middle product in a second LDS array.
template <bool UPPER>
__global__ __launch_bounds__(TB) void trsm_invert_blocks_kernel(const word *__restrict__ T, int64_t t_stride, int64_t mb, word *__restrict__ Tinv) {
  static_assert(TB <= 512 && TB % 128 == 0, "the in-LDS inversion holds a block of at most 512 rows");
  constexpr int TW = TB / 64;
  __shared__ word X[TB][TW];           // the block with its unit diagonal and the other triangle cleared; its inverse at the end
  __shared__ word Mid[TB / 2][TW / 2];  // the middle product of a level: (TB / 2s) pairs x s rows x s / 64 words
  const int64_t r0 = (int64_t)blockIdx.x * TB;
  const int sz     = (int)((mb - r0) < TB ? (mb - r0) : TB);
  const word *blk  = T + r0 * t_stride + r0 / 64;  // the block's rows, from its own first column on
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x; idx < TB * TW; idx += TB) {
    const int i = idx / TW, w = idx % TW;
    word v = 0;
    if (i < sz && w * 64 < sz) {
      v = blk[(int64_t)i * t_stride + w];
      // keep the proper triangle's columns inside the block: (i, c) with c < i (lower) or i < c < sz (upper)
      word keep;
      if (UPPER) {
        keep = (w * 64 > i) ? ~(word)0 : (w * 64 + 63 <= i) ? 0 : (((~(word)0) << (i - w * 64)) << 1);
        if (sz - w * 64 < 64) keep &= (~(word)0) >> (64 - (sz - w * 64));
      } else {
        keep = (w * 64 + 63 < i) ? ~(word)0 : (w * 64 >= i) ? 0 : (((word)1 << (i - w * 64)) - 1);
      }
      v &= keep;
    }
    if (w == i / 64) v |= (word)1 << (i % 64);  // rows beyond sz: identity, cut off at the end
    X[i][w] = v;
  }
  __syncthreads();
  {  // 1. the diagonal 64 x 64 blocks
    word xcol = 0;  // column `lane` of the inverse so far: bit i = entry (i, lane)
    for (int t = 0; t < 64; ++t) {
      const int i      = UPPER ? 63 - t : t;
      const word roww  = X[64 * wave + i][wave];
      const int bit    = (__popcll(roww & xcol) & 1) ^ (i == lane ? 1 : 0);  // xcol has no bit i yet: the diagonal does not count
      xcol |= (word)bit << i;
      const word rowi = __ballot(bit);
      if (lane == 0) X[64 * wave + i][wave] = rowi;
    }
  }
  __syncthreads();
  // 2. pairs of finished s-blocks
#pragma unroll 1
  for (int sb = 64; sb < TB; sb *= 2) {
    const int sw = sb / 64, total = (TB / (2 * sb)) * sb * sw;
    for (int o = threadIdx.x; o < total; o += TB) {  // Mid = L10 * X00  |  U01 * X11
      const int pair = o / (sb * sw), rem = o - pair * sb * sw, i = rem / sw, w = rem - i * sw;
      const int base = pair * 2 * sb;
      const int arow = UPPER ? base + i : base + sb + i, acol = UPPER ? (base + sb) / 64 : base / 64;  // the off-diagonal block's row
      const int brow = UPPER ? base + sb : base, bcol = UPPER ? (base + sb) / 64 : base / 64;            // the finished block next to it
      word acc = 0;
      for (int kw = 0; kw < sw; ++kw) {
        const word aw = X[arow][acol + kw];
#pragma unroll 8
        for (int bt = 0; bt < 64; ++bt) acc ^= X[brow + kw * 64 + bt][bcol + w] & ((word)0 - ((aw >> bt) & 1));
      }
      Mid[pair * sb + i][w] = acc;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < total; o += TB) {  // X10 = X11 * Mid  |  X01 = X00 * Mid
      const int pair = o / (sb * sw), rem = o - pair * sb * sw, i = rem / sw, w = rem - i * sw;
      const int base = pair * 2 * sb;
      const int arow = UPPER ? base + i : base + sb + i, acol = UPPER ? base / 64 : (base + sb) / 64;  // the finished block on this row
      word acc = 0;
      for (int kw = 0; kw < sw; ++kw) {
        const word aw = X[arow][acol + kw];
#pragma unroll 8
        for (int bt = 0; bt < 64; ++bt) acc ^= Mid[pair * sb + kw * 64 + bt][w] & ((word)0 - ((aw >> bt) & 1));
      }
      X[arow][(UPPER ? (base + sb) / 64 : base / 64) + w] = acc;  // over the off-diagonal block, which Mid has replaced
    }
    __syncthreads();
  }
  const int i = threadIdx.x;
  word *dst   = Tinv + ((int64_t)blockIdx.x * TB + i) * TW;
#pragma unroll
  for (int w = 0; w < TW; ++w) dst[w] = (i < sz) ? X[i][w] : 0;
}

struct TrsmScratch {
  word *inv = nullptr, *tmp = nullptr, *big = nullptr, *mid = nullptr;  // big: inverses of 4096-row blocks; mid: the middle products while building them
  size_t inv_words = 0, tmp_words = 0, big_words = 0, mid_words = 0;
  hipEvent_t last = nullptr;  // end of the previous solve that used the scratch (it may have run on another stream)
};
TrsmScratch g_trsm_scratch[16];
std::mutex g_trsm_mu;

struct TrsmRun {
  bool upper;
  const word *T;  // the whole triangle
  int64_t ts;
  word *B;
  int64_t bs, nb;
  int cutoff;
  hipStream_t st;
  const word *inv;  // the inverted diagonal blocks of `be` rows, block k at inv + k * be * (be / 64)
  word *tmp;        // be x words_of(nb) words
  int64_t be = TB;  // rows of an inverted block: TB, or TRSM_BIG when the bigger inverses were built
};

// rows [r0, r0 + mb) of the system, r0 a multiple of TB
int solve_blocks(const TrsmRun &R, int64_t r0, int64_t mb) {
  const int64_t wn = words_of(R.nb);
  word *Bb = R.B + r0 * R.bs;
  if (mb <= R.be) {  // X = T_bb^-1 * B_b
    HIPTRY(hipMemcpy2DAsync(R.tmp, (size_t)wn * 8, Bb, (size_t)R.bs * 8, (size_t)wn * 8, (size_t)mb, hipMemcpyDeviceToDevice, R.st));
    return m4ri_amd_mul_dev(Bb, R.bs, R.inv + (r0 / R.be) * R.be * (R.be / 64), R.be / 64, R.tmp, wn, mb, mb, R.nb, 0, R.cutoff, R.st);
  }
  const int64_t mb1 = (((mb - 1) / R.be + 1) >> 1) * R.be;  // halves on a block boundary
  const word *Tr = R.T + r0 * R.ts + r0 / 64;           // this sub-triangle
  if (!R.upper) {
    if (int rc = solve_blocks(R, r0, mb1)) return rc;
    HIPTRY(m4ri_amd_mul_dev(Bb + mb1 * R.bs, R.bs, Tr + mb1 * R.ts, R.ts, Bb, R.bs, mb - mb1, mb1, R.nb, 1, R.cutoff, R.st));
    return solve_blocks(R, r0 + mb1, mb - mb1);
  }
  if (int rc = solve_blocks(R, r0 + mb1, mb - mb1)) return rc;
  HIPTRY(m4ri_amd_mul_dev(Bb, R.bs, Tr + mb1 / 64, R.ts, Bb + mb1 * R.bs, R.bs, mb1, mb - mb1, R.nb, 1, R.cutoff, R.st));
  return solve_blocks(R, r0, mb1);
}

// ---- inverses of 4096-row blocks ------------------------------------------------------------------------------------
// With 512-row inverses a 65536-row solve still issues 128 block solves and 120 updates of 512 .. 4096 rows -- 248 of its
// 255 products, 14 of its 34 ms, launch-bound.  For big systems the inverses of the 4096-row diagonal blocks are built
// from the 512-row ones by three doubling levels (the scheme of trtri_upper below), every level a handful of BATCHED
// launches over all blocks; the recursion then stops at 4096 rows: 16 block solves and 15 updates.
#ifndef TRSM_BIG
#define TRSM_BIG 4096
#endif
constexpr int64_t BIG = TRSM_BIG, BIGW = TRSM_BIG / 64;

// I4: ng blocks of BIG rows x BIGW words; row r of the triangle -> its block's proper triangle, cleared elsewhere
template <bool UPPER>
__global__ __launch_bounds__(64) void trsm_big_clean_kernel(const word *__restrict__ T, int64_t ts, int64_t mb, word *__restrict__ I4) {
  const int64_t r = blockIdx.x, g = r / BIG;
  const int w     = threadIdx.x;  // word inside the block: BIGW == 64 threads
  const int64_t c0 = g * BIG + 64 * w;
  word v = 0;
  if (r < mb && c0 < mb) {
    v = T[r * ts + g * BIGW + w];
    word keep;
    if (UPPER) {
      keep = (c0 > r) ? ~(word)0 : (c0 + 63 <= r) ? 0 : (((~(word)0) << (r - c0)) << 1);
      if (mb - c0 < 64) keep &= (~(word)0) >> (64 - (mb - c0));
    } else {
      keep = (c0 + 63 < r) ? ~(word)0 : (c0 >= r) ? 0 : (((word)1 << (r - c0)) - 1);
    }
    v &= keep;
  }
  I4[r * BIGW + w] = v;
}
// the inverted TB-row blocks onto the diagonal of I4
__global__ __launch_bounds__(TB) void trsm_big_scatter_kernel(const word *__restrict__ inv, int64_t nblk, word *__restrict__ I4) {
  const int64_t b = blockIdx.x, i = threadIdx.x, r = b * TB + i;
  const int64_t wl = ((b * TB) % BIG) / 64;  // the block's first word inside its BIG block
#pragma unroll
  for (int w = 0; w < TB / 64; ++w) I4[r * BIGW + wl + w] = (b < nblk) ? inv[(b * TB + i) * (TB / 64) + w] : 0;
}

int build_big_inverses(bool upper, const word *T, int64_t ts, int64_t mb, TrsmScratch &s, hipStream_t st) {
  static_assert(BIG % TB == 0 && BIGW == 64, "4096-row blocks, 64 words wide");
  const int64_t ng = (mb + BIG - 1) / BIG, nblk = (mb + TB - 1) / TB;
  if (upper) hipLaunchKernelGGL((trsm_big_clean_kernel<true>), dim3((unsigned)(ng * BIG)), dim3(64), 0, st, T, ts, mb, s.big);
  else       hipLaunchKernelGGL((trsm_big_clean_kernel<false>), dim3((unsigned)(ng * BIG)), dim3(64), 0, st, T, ts, mb, s.big);
  hipLaunchKernelGGL(trsm_big_scatter_kernel, dim3((unsigned)(ng * (BIG / TB))), dim3(TB), 0, st, s.inv, nblk, s.big);
  HIPTRY(hipGetLastError());
  const int64_t gs = BIG * BIGW;  // words between consecutive blocks
  for (int64_t sb = TB; sb < BIG; sb *= 2) {
    const int64_t sw = sb / 64;
    for (int64_t base = 0; base < BIG; base += 2 * sb) {
      word *X00 = s.big + base * BIGW + base / 64, *X11 = s.big + (base + sb) * BIGW + (base + sb) / 64;
      word *off = upper ? s.big + base * BIGW + (base + sb) / 64 : s.big + (base + sb) * BIGW + base / 64;  // U01 | L10
      if (upper) {  // X01 = X00 * U01 * X11
        HIPTRY(m4ri_amd_m4rm_batch_dev(s.mid, sw, sb * sw, X00, BIGW, gs, off, BIGW, gs, sb, sb, sb, ng, 0, st));
        HIPTRY(m4ri_amd_m4rm_batch_dev(off, BIGW, gs, s.mid, sw, sb * sw, X11, BIGW, gs, sb, sb, sb, ng, 0, st));
      } else {      // X10 = X11 * L10 * X00
        HIPTRY(m4ri_amd_m4rm_batch_dev(s.mid, sw, sb * sw, off, BIGW, gs, X00, BIGW, gs, sb, sb, sb, ng, 0, st));
        HIPTRY(m4ri_amd_m4rm_batch_dev(off, BIGW, gs, X11, BIGW, gs, s.mid, sw, sb * sw, sb, sb, sb, ng, 0, st));
      }
    }
  }
  return 0;
}

int solve(bool upper, const word *T, int64_t ts, word *B, int64_t bs, int64_t mb, int64_t nb, int cutoff, hipStream_t st) {
  if (mb <= 1 || nb <= 0) return 0;
  if (mb <= 64) {
    const int64_t wn = words_of(nb);
    const word mask  = (nb % 64) ? ((~(word)0) >> (64 - nb % 64)) : ~(word)0;
    const unsigned g = (unsigned)((wn + TRSM_THREADS - 1) / TRSM_THREADS);
    if (upper) hipLaunchKernelGGL((trsm_base_kernel<true>), dim3(g), dim3(TRSM_THREADS), 0, st, T, ts, B, bs, (int)mb, wn, mask);
    else       hipLaunchKernelGGL((trsm_base_kernel<false>), dim3(g), dim3(TRSM_THREADS), 0, st, T, ts, B, bs, (int)mb, wn, mask);
    return (int)hipGetLastError();
  }
  // The operands handed to the multiply engine must not carry bits beyond their own columns: the sub-diagonal blocks
  // T[r.., c..c+l) end on a 512-column boundary or on the triangle's last column mb -- the caller keeps T's bits beyond
  // column mb out of the way (a clean mb x mb matrix, or a masked copy: see echelon.hip / solve.hip).
  std::lock_guard<std::mutex> lk(g_trsm_mu);
  int dev = 0;
  HIPTRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  TrsmScratch &s = g_trsm_scratch[dev];
  const int64_t nblk = (mb + TB - 1) / TB, wn = words_of(nb);
  const size_t need_inv = (size_t)nblk * TB * (TB / 64), need_tmp = (size_t)TB * (size_t)wn;
  // grow-only scratch; a buffer still in use by an earlier call on another stream must not be freed under it
  if (need_inv > s.inv_words) {
    if (s.inv) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.inv)); }
    s.inv = nullptr; s.inv_words = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.inv), need_inv * 8));
    s.inv_words = need_inv;
  }
  if (need_tmp > s.tmp_words) {
    if (s.tmp) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.tmp)); }
    s.tmp = nullptr; s.tmp_words = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.tmp), need_tmp * 8));
    s.tmp_words = need_tmp;
  }
  if (!s.last) HIPTRY(hipEventCreateWithFlags(&s.last, hipEventDisableTiming));
  else HIPTRY(hipStreamWaitEvent(st, s.last, 0));
  if (upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<true>), dim3((unsigned)nblk), dim3(TB), 0, st, T, ts, mb, s.inv);
  else       hipLaunchKernelGGL((trsm_invert_blocks_kernel<false>), dim3((unsigned)nblk), dim3(TB), 0, st, T, ts, mb, s.inv);
  HIPTRY(hipGetLastError());
  TrsmRun R{upper, T, ts, B, bs, nb, cutoff, st, s.inv, s.tmp};
  static const int big_env = getenv("M4RI_AMD_TRSM_BIG") ? atoi(getenv("M4RI_AMD_TRSM_BIG")) : -1;
  // from 4097 rows on, whatever the number of columns: these solves are bound by their launches (8192 x 512: 1.12 -> 0.75 ms,
  // 16384 x 16384: 3.4 -> 1.4 ms, 32768^2: 9.4 -> 4.8, 65536^2: 33.3 -> 23.2; M4RI_AMD_TRSM_BIG=0 / 1 forces the choice)
  if (big_env >= 0 ? (big_env != 0 && mb > BIG) : (mb > BIG)) {
    const int64_t ng = (mb + BIG - 1) / BIG;
    const size_t need_big = (size_t)ng * BIG * BIGW, need_mid = (size_t)ng * (BIG / 2) * (BIGW / 2), need_tmp4 = (size_t)BIG * (size_t)wn;
    if (need_big > s.big_words) {
      if (s.big) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.big)); }
      s.big = nullptr; s.big_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.big), need_big * 8));
      s.big_words = need_big;
    }
    if (need_mid > s.mid_words) {
      if (s.mid) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.mid)); }
      s.mid = nullptr; s.mid_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.mid), need_mid * 8));
      s.mid_words = need_mid;
    }
    if (need_tmp4 > s.tmp_words) {
      if (s.tmp) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.tmp)); }
      s.tmp = nullptr; s.tmp_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.tmp), need_tmp4 * 8));
      s.tmp_words = need_tmp4;
    }
    if (int rc = build_big_inverses(upper, T, ts, mb, s, st)) return rc;
    R.inv = s.big; R.tmp = s.tmp; R.be = BIG;
  }
  const int rc = solve_blocks(R, 0, mb);
  HIPTRY(hipEventRecord(s.last, st));
  return rc;
}

// ---- right-hand solves: B <- B * T^-1 (X * T = B) -----------------------------------------------------------------
// (triangular.c:41-130 upper, :301-393 lower).  The unknowns of a row are its own bits, so the base case works
// inside one 64-bit word per row: x_j = b_j ^ parity(x & col_j(T)) for the columns in dependency order (ascending for
// an upper triangle, descending for a lower one) -- a thread per row, the 64 column masks of the triangle built once
// per workgroup in LDS.  Above 64 columns: halves on a word boundary, the update one engine product.
template <bool UPPER>
__global__ __launch_bounds__(256) void trsm_right_base_kernel(const word *__restrict__ T, int64_t t_stride, word *__restrict__ B,
                                                              int64_t b_stride, int64_t mb, int nb) {
  __shared__ word col[64];  // col[j]: bit i = T[i][j] for the rows i that x_j depends on
  if (threadIdx.x < 64) {
    const int j = threadIdx.x;
    word c = 0;
    if (j < nb)
      for (int i = UPPER ? 0 : j + 1; i < (UPPER ? j : nb); ++i) c |= ((T[(int64_t)i * t_stride] >> j) & 1) << i;
    col[j] = c;
  }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= mb) return;
  word x = B[r * b_stride];
  if (UPPER) {
    for (int j = 1; j < nb; ++j) x ^= (word)(__popcll(x & col[j]) & 1) << j;
  } else {
    for (int j = nb - 2; j >= 0; --j) x ^= (word)(__popcll(x & col[j]) & 1) << j;
  }
  B[r * b_stride] = x;
}

// columns [c0, c0 + nb) of the system X T = B, c0 a multiple of TB: the same scheme as solve_blocks -- X_b = B_b * T_bb^-1
int solve_right_blocks(const TrsmRun &R, int64_t mrows, int64_t c0, int64_t nb) {
  word *Bb = R.B + c0 / 64;
  if (nb <= R.be) {
    const int64_t wb = words_of(nb);
    HIPTRY(hipMemcpy2DAsync(R.tmp, (size_t)wb * 8, Bb, (size_t)R.bs * 8, (size_t)wb * 8, (size_t)mrows, hipMemcpyDeviceToDevice, R.st));
    if (nb % 64) HIPTRY(m4ri_amd_mask_tail_dev(R.tmp, wb, mrows, nb, R.st));  // the copy's last word may carry the columns behind the system
    // the product writes whole words of its nb columns: bits of B beyond column c0 + nb in that word are rewritten as zero,
    // which is what they are (B's own tail) whenever nb is not a multiple of 64 -- that only happens in the last block
    return m4ri_amd_mul_dev(Bb, R.bs, R.tmp, wb, R.inv + (c0 / R.be) * R.be * (R.be / 64), R.be / 64, mrows, nb, nb, 0, R.cutoff, R.st);
  }
  const int64_t nb1 = (((nb - 1) / R.be + 1) >> 1) * R.be;
  const word *Tr = R.T + c0 * R.ts + c0 / 64;
  if (R.upper) {
    if (int rc = solve_right_blocks(R, mrows, c0, nb1)) return rc;
    HIPTRY(m4ri_amd_mul_dev(Bb + nb1 / 64, R.bs, Bb, R.bs, Tr + nb1 / 64, R.ts, mrows, nb1, nb - nb1, 1, R.cutoff, R.st));
    return solve_right_blocks(R, mrows, c0 + nb1, nb - nb1);
  }
  if (int rc = solve_right_blocks(R, mrows, c0 + nb1, nb - nb1)) return rc;
  HIPTRY(m4ri_amd_mul_dev(Bb, R.bs, Bb + nb1 / 64, R.bs, Tr + nb1 * R.ts, R.ts, mrows, nb - nb1, nb1, 1, R.cutoff, R.st));
  return solve_right_blocks(R, mrows, c0, nb1);
}

int solve_right(bool upper, const word *T, int64_t ts, word *B, int64_t bs, int64_t mb, int64_t nb, int cutoff, hipStream_t st) {
  if (mb <= 0 || nb <= 1) return 0;
  if (nb <= 64) {
    const unsigned g = (unsigned)((mb + 255) / 256);
    if (upper) hipLaunchKernelGGL((trsm_right_base_kernel<true>), dim3(g), dim3(256), 0, st, T, ts, B, bs, mb, (int)nb);
    else       hipLaunchKernelGGL((trsm_right_base_kernel<false>), dim3(g), dim3(256), 0, st, T, ts, B, bs, mb, (int)nb);
    return (int)hipGetLastError();
  }
  std::lock_guard<std::mutex> lk(g_trsm_mu);
  int dev = 0;
  HIPTRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  TrsmScratch &s = g_trsm_scratch[dev];
  const int64_t nblk = (nb + TB - 1) / TB;
  const size_t need_inv = (size_t)nblk * TB * (TB / 64), need_tmp = (size_t)mb * (TB / 64);
  if (need_inv > s.inv_words) {
    if (s.inv) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.inv)); }
    s.inv = nullptr; s.inv_words = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.inv), need_inv * 8));
    s.inv_words = need_inv;
  }
  if (need_tmp > s.tmp_words) {
    if (s.tmp) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.tmp)); }
    s.tmp = nullptr; s.tmp_words = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.tmp), need_tmp * 8));
    s.tmp_words = need_tmp;
  }
  if (!s.last) HIPTRY(hipEventCreateWithFlags(&s.last, hipEventDisableTiming));
  else HIPTRY(hipStreamWaitEvent(st, s.last, 0));
  if (upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<true>), dim3((unsigned)nblk), dim3(TB), 0, st, T, ts, nb, s.inv);
  else       hipLaunchKernelGGL((trsm_invert_blocks_kernel<false>), dim3((unsigned)nblk), dim3(TB), 0, st, T, ts, nb, s.inv);
  HIPTRY(hipGetLastError());
  TrsmRun R{upper, T, ts, B, bs, nb, cutoff, st, s.inv, s.tmp};
  static const int big_env = getenv("M4RI_AMD_TRSM_BIG") ? atoi(getenv("M4RI_AMD_TRSM_BIG")) : -1;
  if (big_env >= 0 ? (big_env != 0 && nb > BIG) : (nb > BIG)) {  // the 4096-row block inverses, as in solve()
    const int64_t ng = (nb + BIG - 1) / BIG;
    const size_t need_big = (size_t)ng * BIG * BIGW, need_mid = (size_t)ng * (BIG / 2) * (BIGW / 2), need_tmp4 = (size_t)mb * BIGW;
    if (need_big > s.big_words) {
      if (s.big) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.big)); }
      s.big = nullptr; s.big_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.big), need_big * 8));
      s.big_words = need_big;
    }
    if (need_mid > s.mid_words) {
      if (s.mid) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.mid)); }
      s.mid = nullptr; s.mid_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.mid), need_mid * 8));
      s.mid_words = need_mid;
    }
    if (need_tmp4 > s.tmp_words) {
      if (s.tmp) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.tmp)); }
      s.tmp = nullptr; s.tmp_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.tmp), need_tmp4 * 8));
      s.tmp_words = need_tmp4;
    }
    if (int rc = build_big_inverses(upper, T, ts, nb, s, st)) return rc;
    R.inv = s.big; R.tmp = s.tmp; R.be = BIG;
  }
  const int rc = solve_right_blocks(R, mb, 0, nb);
  HIPTRY(hipEventRecord(s.last, st));
  return rc;
}

}  // namespace

extern "C" {

// B (mb x nb) <- B * U^-1 / B * L^-1, T (nb x nb) unit triangular (diagonal and other triangle never read)
int m4ri_amd_trsm_upper_right_dev(const word *U, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                  void *stream) {
  if (mb < 0 || nb < 0 || cutoff < 0) return (int)hipErrorInvalidValue;
  return solve_right(true, U, t_stride, B, b_stride, mb, nb, cutoff, (hipStream_t)stream);
}
int m4ri_amd_trsm_lower_right_dev(const word *L, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                  void *stream) {
  if (mb < 0 || nb < 0 || cutoff < 0) return (int)hipErrorInvalidValue;
  return solve_right(false, L, t_stride, B, b_stride, mb, nb, cutoff, (hipStream_t)stream);
}

// B (mb x nb bits, stride b_stride) <- L^-1 B, L (mb x mb, stride t_stride) unit lower triangular: only the
// bits strictly below its diagonal are read.  Device pointers; bits of B at column >= nb must be zero on
// entry and are zero on return.  Asynchronous on `stream`.
int m4ri_amd_trsm_lower_left_dev(const word *L, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                 void *stream) {
  if (mb < 0 || nb < 0 || cutoff < 0) return (int)hipErrorInvalidValue;
  return solve(false, L, t_stride, B, b_stride, mb, nb, cutoff, (hipStream_t)stream);
}

// B <- U^-1 B, U unit upper triangular: only the bits strictly above its diagonal are read.
int m4ri_amd_trsm_upper_left_dev(const word *U, int64_t t_stride, word *B, int64_t b_stride, int64_t mb, int64_t nb, int cutoff,
                                 void *stream) {
  if (mb < 0 || nb < 0 || cutoff < 0) return (int)hipErrorInvalidValue;
  return solve(true, U, t_stride, B, b_stride, mb, nb, cutoff, (hipStream_t)stream);
}

}  // extern "C"

// ---- inverse of a unit upper triangular matrix, in place ------------------------------------------------------------
// mzd_trtri_upper (triangular.c:518-547: halves, two TRSMs on the off-diagonal block, recursion) and its base case
// mzd_trtri_upper_russian (triangular_russian.c:384-470: column by column inside k-bit blocks, the rows above through
// tables).  The inverse is unique, so the schedule is free; here it runs bottom-up on a clean copy W of the triangle
// (strict upper part + unit diagonal, nothing else -- the caller's diagonal and lower triangle are neither read nor
// written, as in the reference):
//   1. all diagonal blocks of 512 rows inverted in one launch (the TRSM kernel above) and put back into W;
//   2. for s = 512, 1024, ...: every pair of finished s-blocks [X00 ; X11] closes its off-diagonal block with two engine
//      products,  W01 <- X00 * W01 * X11  ((U^-1)01 = U00^-1 U01 U11^-1 over GF(2));  254 products at n = 65536 instead of
//      the ~1800 small ones a top-down recursion over TRSMs would launch, and the levels up to 4096 -- 240 of the 254 --
//      are two batched launches each;
//   3. the strict upper triangle of W goes back into the caller's matrix.
namespace {

__device__ __forceinline__ word strict_upper_mask(int64_t i, int64_t w) {  // bits of word w at columns > i
  const int64_t c0 = w * 64;
  if (c0 > i) return ~(word)0;
  if (c0 + 63 <= i) return 0;
  return ((~(word)0) << (i - c0)) << 1;
}

__global__ __launch_bounds__(256) void trtri_clean_kernel(word *__restrict__ W, int64_t ws, const word *__restrict__ U, int64_t us,
                                                          int64_t n, int64_t wn) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * wn) return;
  const int64_t i = idx / wn, w = idx - i * wn;
  word v = U[i * us + w] & strict_upper_mask(i, w);
  if ((i >> 6) == w) v |= (word)1 << (i & 63);
  if (w == wn - 1 && (n & 63)) v &= (~(word)0) >> (64 - (n & 63));
  W[i * ws + w] = v;
}

__global__ __launch_bounds__(TB) void trtri_scatter_kernel(word *__restrict__ W, int64_t ws, const word *__restrict__ inv, int64_t n,
                                                           int64_t wn) {
  const int64_t r0 = (int64_t)blockIdx.x * TB, i = r0 + threadIdx.x;
  if (i >= n) return;
  const word *src = inv + ((int64_t)blockIdx.x * TB + threadIdx.x) * (TB / 64);
#pragma unroll
  for (int w = 0; w < TB / 64; ++w)
    if (r0 / 64 + w < wn) W[i * ws + r0 / 64 + w] = src[w];
}

__global__ __launch_bounds__(256) void trtri_merge_kernel(word *__restrict__ U, int64_t us, const word *__restrict__ W, int64_t ws,
                                                          int64_t n, int64_t wn) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * wn) return;
  const int64_t i = idx / wn, w = idx - i * wn;
  word m = strict_upper_mask(i, w);
  if (w == wn - 1 && (n & 63)) m &= (~(word)0) >> (64 - (n & 63));
  if (m == 0) return;
  word *p = U + i * us + w;
  *p = (*p & ~m) | (W[i * ws + w] & m);
}

struct TrtriScratch { word *buf = nullptr; size_t words = 0; };
#ifndef TRTRI_BATCH_MAX
#define TRTRI_BATCH_MAX 4096  // levels up to this block size run as batched launches
#endif
TrtriScratch g_trtri_scratch[16];

int trtri_upper(word *U, int64_t us, int64_t n, hipStream_t st) {
  if (n <= 1) return 0;
  std::lock_guard<std::mutex> lk(g_trsm_mu);
  int dev = 0;
  HIPTRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  const int64_t wn = words_of(n), ws = (wn + 1) & ~(int64_t)1, nblk = (n + TB - 1) / TB;
  int64_t smax = TB;
  while (smax * 2 < n) smax *= 2;  // the largest level: pairs of smax-blocks
  const int64_t wt = (words_of(smax) + 1) & ~(int64_t)1;
  const size_t w_words = (size_t)n * (size_t)ws, t_words = (n > TB) ? (size_t)smax * (size_t)wt : 0,
               inv_words = (size_t)nblk * TB * (TB / 64);
  TrtriScratch &s = g_trtri_scratch[dev];
  if (w_words + t_words + inv_words > s.words) {
    if (s.buf) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.buf)); }
    s.buf = nullptr; s.words = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.buf), (w_words + t_words + inv_words) * 8));
    s.words = w_words + t_words + inv_words;
  }
  word *W = s.buf, *T = W + w_words, *inv = T + t_words;
  const unsigned g = (unsigned)((n * wn + 255) / 256);
  hipLaunchKernelGGL(trtri_clean_kernel, dim3(g), dim3(256), 0, st, W, ws, U, us, n, wn);
  hipLaunchKernelGGL((trsm_invert_blocks_kernel<true>), dim3((unsigned)nblk), dim3(TB), 0, st, W, ws, n, inv);
  hipLaunchKernelGGL(trtri_scatter_kernel, dim3((unsigned)nblk), dim3(TB), 0, st, W, ws, inv, n, wn);
  HIPTRY(hipGetLastError());
  for (int64_t sz = TB; sz < n; sz *= 2) {
    int64_t r0 = 0;
    // the complete pairs of a small level are one batched launch each way (a pair starts every 2 sz rows and 2 sz columns,
    // so the batch stride is constant): 240 of the 254 products at n = 65536 are in the levels up to 4096, launch-bound
    // one by one (~35 us each)
    const int64_t full = n / (2 * sz);
    if (sz <= TRTRI_BATCH_MAX && full > 1) {
      const int64_t wsz = sz / 64, pair = 2 * sz * ws + 2 * wsz;
      HIPTRY(m4ri_amd_m4rm_batch_dev(T, wsz, sz * wsz, W, ws, pair, W + wsz, ws, pair, sz, sz, sz, full, 0, st));
      HIPTRY(m4ri_amd_m4rm_batch_dev(W + wsz, ws, pair, T, wsz, sz * wsz, W + sz * ws + wsz, ws, pair, sz, sz, sz, full, 0, st));
      r0 = full * 2 * sz;
    }
    for (; r0 + sz < n; r0 += 2 * sz) {
      const int64_t mid = r0 + sz, s2 = (n - mid) < sz ? (n - mid) : sz;
      word *W00 = W + r0 * ws + r0 / 64, *W01 = W + r0 * ws + mid / 64, *W11 = W + mid * ws + mid / 64;
      HIPTRY(m4ri_amd_mul_dev(T, wt, W00, ws, W01, ws, sz, sz, s2, 0, 0, st));
      HIPTRY(m4ri_amd_mul_dev(W01, ws, T, wt, W11, ws, sz, s2, s2, 0, 0, st));
    }
  }
  hipLaunchKernelGGL(trtri_merge_kernel, dim3(g), dim3(256), 0, st, U, us, W, ws, n, wn);
  return (int)hipGetLastError();
}

}  // namespace

// U (n x n bits) <- U^-1 for a unit upper triangular U: only the bits strictly above the diagonal are read and written.
// Device pointer; asynchronous on `stream` (the scratch is grow-only per device, calls are serialised by a mutex).
extern "C" int m4ri_amd_trtri_upper_dev(word *U, int64_t stride, int64_t n, void *stream) {
  if (n < 0) return (int)hipErrorInvalidValue;
  return trtri_upper(U, stride, n, (hipStream_t)stream);
}
