// ple.hip -- PLE decomposition on the device: A = P L E Q in place, the elimination step above the multiply
// path.
//
// Reference interfaces replaced (same A, P, Q and rank, bit for bit):
//   _mzd_ple_russian   /root/reference m4ri/ple_russian.c:380-617   the block-iterative "Russian" base case
//   _mzd_ple / mzd_ple m4ri/ple.c:62-171, m4ri/ple.h                the recursive driver over it
// What they compute is fixed by the pivoting rule, not by the schedule: columns left to right, pivot = the
// first row at or below the current rank position whose bit in the column is set after elimination by the
// earlier pivots, swap it up, clear the rows below from the NEXT column on (the multiplier stays in the pivot
// column), finally move L's columns to the left (ple_russian.c:596-602).  The CPU checker states that rule
// column by column and tests/test_ple_oracle.py pins it against both reference routines.
//
// Schedule here -- right-looking, one 64-column word block at a time, everything on the device:
//   1. the block's word of every remaining row is copied into a dense vector; ONE workgroup finds the block's
//      pivots there column by column (candidates examined 1024 at a time, each reduced on the fly by the pivots
//      found so far: the only sequential part, and for generic input independent of the row count);
//   2. the row swaps are applied to the other words; a parallel pass replays the pivots on every row's slice
//      word (-> multipliers in place); the pivot rows (<= 64) are reduced among themselves on the words to the
//      right by a 64-row triangular solve (the reference's A10 step, ple_russian.c:306-325);
//   3. rows below, words to the right:  C ^= M * U  with M = the rows' multipliers (<= 64 bits each) and U the
//      block's pivot rows -- a rank-<=64 update by a streaming kernel with sixteen 4-bit tables per column tile in
//      LDS (the reference's table steps _mzd_ple_a11_N / _mzd_process_rows_ple_N, ple_russian_template.h, are this
//      product by seven 8-bit tables); it reads and writes the trailing matrix once per 64 columns: HBM-bound;
//   4. after the last block one gather pass compresses L (closed form of ple_russian.c:596-602: row r takes
//      new[j] = old[Q[j]] for j <= min(r, rank - 1) and zeros at the vacated pivot columns).
// The host only reads back the block's pivots (a few hundred bytes per 64 columns).
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

namespace {

#define HIPTRY(expr)                                  \
  do {                                                \
    hipError_t e_ = (hipError_t)(expr);               \
    if (e_ != hipSuccess) return (int)e_;             \
  } while (0)

#ifndef PLE_SLICE_THREADS
#define PLE_SLICE_THREADS 1024
#endif
constexpr int SLICE_THREADS = PLE_SLICE_THREADS;
constexpr int ROW_THREADS   = 256;

struct PleBlock {
  int32_t rank;
  int32_t pivcol[64];   // column of pivot t inside the block
  int32_t swaprow[64];  // absolute row that was swapped into position r0 + t
  word vhigh[64];       // pivot t's slice word from the column after its pivot column on
  word Lc[64];          // pivot row t's multipliers (bits j < t): the block's unit lower triangle L
  word Linv[64];        // row t of L^-1 (block_triangle below)
  int32_t nsrc;         // > 0: the block's row swaps as ONE permutation of the first nsrc rows (the one-wave search knows it) ...
  int32_t src[128];     // ... position t takes the row that stood at position src[t]
  word deferred;        // columns the one-wave search passed over because none of ITS 128 rows had a pivot there while rows lie
                        // beyond: to be confirmed against those rows (ple_verify_kernel) before anything is committed
  word stage[128];      // with deferred != 0: the rearranged slice words of the first 128 rows, not yet written to the matrix
  int32_t missed;       // ple_verify_kernel: a row beyond the window does have a bit in a deferred column
};

// ---- 0. the block's word of every remaining row -> dense vector -------------------------------------------------
__global__ __launch_bounds__(ROW_THREADS) void ple_extract_kernel(const word *__restrict__ A, int64_t stride, int64_t nrows, int64_t r0, int64_t wb,
                                                                 word *__restrict__ V) {
  const int64_t i = (int64_t)blockIdx.x * ROW_THREADS + threadIdx.x;
  if (i < nrows - r0) V[i] = A[(r0 + i) * stride + wb];
}

// The block's 64 x 64 unit triangle L and its inverse, by one wave at the end of a pivot search: lane t brings pivot row
// t's final slice word and pivot t's column.  Lc[t] = the row's multipliers (its bits at the columns of the pivots before
// it); L^-1 by substitution, a lane per column, rows assembled by ballots.  The rows below are updated with (M L^-1) U*
// instead of M U (ple_finish_kernel), so the pivot rows' own solve U = L^-1 U* leaves the critical path.
__device__ __forceinline__ word wave_read64(word x, int lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
  return ((word)hi << 32) | lo;
}
__device__ __forceinline__ void block_triangle(int tid, word my_word, int my_col, int rank, PleBlock *__restrict__ out) {
  word lc = 0;
  for (int j = 0; j < rank; ++j) {
    const int cj = __builtin_amdgcn_readlane(my_col, j);
    if (j < tid && tid < rank) lc |= ((my_word >> cj) & 1) << j;
  }
  word x = 0;  // column tid of L^-1
  for (int t = 0; t < rank; ++t) {
    const word lt = wave_read64(lc, t);
    x |= (word)((__popcll(lt & x) & 1) ^ (t == tid ? 1 : 0)) << t;
  }
  word mine = 0;
  for (int t = 0; t < rank; ++t) {
    const word row = __ballot((x >> t) & 1);
    if (tid == t) mine = row;
  }
  out->Lc[tid]   = lc;
  out->Linv[tid] = tid < rank ? mine : 0;
}

// ---- 1. the block's pivots (ONE workgroup) ---------------------------------------------------------------------------
// Column by column: the candidates are the rows from the current rank position down; the first 1024 of them live
// in the lanes of the workgroup, already reduced by the pivots found so far (see "the window" below); the first lane
// whose bit in the column is set holds the pivot (ple_russian.c:141-159 does the same lazily, row by row).
// Rows are only read here: the full elimination of the slice is a parallel pass afterwards (ple_finish_kernel),
// replaying the same steps from the original words -- a row's value when pivot l is applied does not depend on
// when that happens.  For generic input the pivot sits in the first chunk, so the cost is independent of nrows.
__global__ __launch_bounds__(SLICE_THREADS) void ple_pivots_kernel(int64_t n, int64_t r0, int ncb, word *__restrict__ V, PleBlock *__restrict__ out) {
  __shared__ int s_min;
  __shared__ word s_vp;
  __shared__ word s_high[64];  // pivot l from the column after its pivot column on
  __shared__ int s_col[64];
  __shared__ word s_head[SLICE_THREADS + 64];  // the ORIGINAL words of the first 1088 rows, kept in step with the swaps
  __shared__ word s_shift[SLICE_THREADS];      // the window moving one lane to the left
  __shared__ int s_cnt[SLICE_THREADS];
  const int tid   = threadIdx.x;
  const int nhead = (int)(n < SLICE_THREADS + 64 ? n : SLICE_THREADS + 64);
  for (int i = tid; i < nhead; i += SLICE_THREADS) s_head[i] = V[i];
  if (tid == 0) s_min = INT_MAX;
  __syncthreads();
  // The window: lane t watches row rank + t and holds its word reduced by the first `cnt` pivots.  A new pivot costs
  // the up-to-date lanes ONE step; then the window moves one lane to the left (the row the pivot displaced takes
  // the pivot's place) and a fresh row enters at the top lane, stale (cnt = 0) -- it catches up only if a column
  // ever has no hit among the up-to-date lanes, which for generic input never happens within a block.
  word v  = tid < nhead ? s_head[tid] : 0;
  int cnt = 0, rank = 0;
  for (int c = 0; c < ncb && rank < n; ++c) {
    int p = INT_MAX;
    for (int pass = 0; pass < 2 && p == INT_MAX; ++pass) {
      const bool valid = (int64_t)rank + tid < n;
      if (pass == 1) {
        const bool stale = valid && cnt < rank;
        if (!__syncthreads_or(stale)) break;  // nobody to catch up: the window has no pivot for this column
        if (stale) {
          for (int l = cnt; l < rank; ++l) v ^= ((v >> s_col[l]) & 1) ? s_high[l] : 0;
          cnt = rank;
        }
      }
      const bool hit = valid && cnt == rank && ((v >> c) & 1);
      const unsigned long long b = __ballot(hit);
      if (b && (tid & 63) == 0) atomicMin(&s_min, rank + (tid & ~63) + (int)__builtin_ctzll(b));
      __syncthreads();
      p = s_min;
      if (p == rank + tid) s_vp = v;  // the lane that holds row p publishes its reduced word
      __syncthreads();
    }
    // rows beyond the window: from their original words, all pivots replayed (ple_russian.c:141-159 walks on lazily too)
    for (int64_t base = (int64_t)rank + SLICE_THREADS; p == INT_MAX && base < n; base += SLICE_THREADS) {
      const int64_t i = base + tid;
      word x = i < nhead ? s_head[i] : (i < n ? V[i] : 0);
#pragma unroll 8
      for (int l = 0; l < rank; ++l) x ^= ((x >> s_col[l]) & 1) ? s_high[l] : 0;
      const unsigned long long b = __ballot((x >> c) & 1);
      if (b && (tid & 63) == 0) atomicMin(&s_min, (int)(base + (tid & ~63) + __builtin_ctzll(b)));
      __syncthreads();
      p = s_min;
      if (p == (int)i) s_vp = x;
      __syncthreads();
    }
    if (p == INT_MAX) continue;  // no pivot in this column
    if (tid == 0) {
      const word vp = s_vp;
      const word vr = rank < nhead ? s_head[rank] : V[rank];
      V[p]    = vr;       // the displaced row keeps its original word: it is reduced with everybody else afterwards
      V[rank] = vp;       // the pivot row's word is final (reduced by the earlier pivots, multipliers in place)
      if (p < nhead) s_head[p] = vr;
      if (rank < nhead) s_head[rank] = vp;
      s_high[rank] = c < 63 ? (vp & (~(word)0 << (c + 1))) : 0;
      s_col[rank]  = c;
      s_min        = INT_MAX;
      out->pivcol[rank]  = c;
      out->swaprow[rank] = (int32_t)(r0 + p);
      out->vhigh[rank]   = s_high[rank];
    }
    __syncthreads();
    // the window takes the pivot and moves on
    const int pl = p - rank;  // lane that held the pivot row (>= SLICE_THREADS: it came from beyond the window)
    if (cnt == rank) {
      if (tid != pl) v ^= ((v >> c) & 1) ? s_high[rank] : 0;
      cnt = rank + 1;
    }
    s_shift[tid] = v;
    s_cnt[tid]   = cnt;
    __syncthreads();
    if (tid == 0 && pl > 0 && pl < SLICE_THREADS) { s_shift[pl] = v; s_cnt[pl] = cnt; }  // the displaced row sits where the pivot was
    __syncthreads();
    if (tid + 1 < SLICE_THREADS) {
      v   = s_shift[tid + 1];
      cnt = s_cnt[tid + 1];
    } else {  // the row entering the window: original word, no pivot applied yet
      const int64_t i = (int64_t)rank + 1 + tid;
      v   = i < nhead ? s_head[i] : (i < n ? V[i] : 0);
      cnt = 0;
    }
    ++rank;
  }
  __syncthreads();
  if (tid < 64) block_triangle(tid, tid < rank ? s_head[tid] : 0, tid < rank ? s_col[tid] : 0, rank, out);
  if (tid == 0) { out->rank = rank; out->nsrc = 0; out->deferred = 0; }
}

// ---- 1'. the same search in ONE WAVE, for blocks whose pivots all sit within the first 128 rows ---------------------
// The general kernel above spends ~1 us per column on workgroup barriers and LDS hand-offs (60 us per block whatever the
// matrix).  Here lane t keeps the rows at positions t and 64 + t in registers -- the word reduced by the pivots so far, and
// the original word that goes back to V -- every candidate is reduced at every step (two XORs per lane), the pivot is the
// lowest set bit of a ballot over the low rows or else the high ones, its word comes by v_readlane, the row it displaces
// is handed over by v_readlane too, and pivot l's column and tail live in lane l: no LDS, no barrier, no lane shift.
// Same rule, same outputs.  With r pivots found there are 128 - r >= 64 candidates left, so a column that has a pivot
// further down misses here with probability 2^-64 for generic input; sparse and structured inputs do miss: then the kernel
// gives up before writing anything (rank = -1) and the host runs the general kernel on the block.
// V: the block's word of the rows from r0 on, element i at V[i * vs] -- the matrix's own word column (vs = the row stride):
// this path needs no dense copy of the slice.
// hout: the host's pinned mirror of the record -- what the host wants from it (rank, pivot columns, swapped rows) is written
// straight there, so no copy has to be queued behind the kernel.
__global__ __launch_bounds__(64) void ple_pivots_wave_kernel(int64_t n, int64_t r0, int ncb, word *__restrict__ V, int64_t vs, PleBlock *__restrict__ out,
                                                             PleBlock *__restrict__ hout) {
  const int tid = threadIdx.x;
  const bool has_lo = tid < n, has_hi = (int64_t)tid + 64 < n;
  const word org_lo = has_lo ? V[tid * vs] : 0, org_hi = has_hi ? V[(tid + 64) * vs] : 0;  // the slice words as they stand
  word v_lo = org_lo, v_hi = org_hi;  // the rows in this lane's two slots, reduced by the pivots found so far
  int i_lo = tid, i_hi = tid + 64;    // where the row in each slot stood at the start
  word pv = 0, ph = 0;                // lane l: pivot row l's final word, and the part of it behind its pivot column
  int pc = 0, psw = 0;                // lane l: pivot l's column, and the position the pivot row was found at
  int rank = 0;
  word deferred = 0;
  const int cols = ncb < 64 ? ncb : 64;
  for (int c = 0; c < cols && rank < n; ++c) {
    const unsigned long long b_lo = __ballot(tid >= rank && ((v_lo >> c) & 1));  // slots without a row hold 0
    if (__builtin_expect(b_lo != 0, 1)) {
      const int pl     = (int)__builtin_ctzll(b_lo);
      const word vp    = wave_read64(v_lo, pl);    // the pivot row's word: final
      const word red_r = wave_read64(v_lo, rank);  // the row at the rank position: the pivot row displaces it
      const int idx_r  = __builtin_amdgcn_readlane(i_lo, rank), idx_p = __builtin_amdgcn_readlane(i_lo, pl);
      const word high  = c < 63 ? (vp & (~(word)0 << (c + 1))) : 0;
      if (tid == pl) { v_lo = red_r; i_lo = idx_r; }
      if (tid == rank) { pv = vp; ph = high; pc = c; psw = pl; i_lo = idx_p; }
      if (tid > rank && ((v_lo >> c) & 1)) v_lo ^= high;
      if ((v_hi >> c) & 1) v_hi ^= high;
      ++rank;
      continue;
    }
    const unsigned long long b_hi = __ballot((v_hi >> c) & 1);
    if (!b_hi) {
      // no pivot among the rows held here.  If rows lie beyond, the column is passed over on the assumption that they have none
      // either (a column without a pivot is the usual reason), and the assumption is checked before anything is committed.
      // Later pivots do not disturb the check: they only change bits behind their own, later, columns.
      if (n > 128) deferred |= (word)1 << c;
      continue;
    }
    {  // the pivot sits in a high slot (rare)
      const int pl     = (int)__builtin_ctzll(b_hi);
      const word vp    = wave_read64(v_hi, pl);
      const word red_r = wave_read64(v_lo, rank);
      const int idx_r  = __builtin_amdgcn_readlane(i_lo, rank), idx_p = __builtin_amdgcn_readlane(i_hi, pl);
      const word high  = c < 63 ? (vp & (~(word)0 << (c + 1))) : 0;
      if (tid == pl) { v_hi = red_r; i_hi = idx_r; }
      if (tid == rank) { pv = vp; ph = high; pc = c; psw = pl + 64; i_lo = idx_p; }
      if (tid > rank && ((v_lo >> c) & 1)) v_lo ^= high;
      if ((v_hi >> c) & 1) v_hi ^= high;
      ++rank;
    }
  }
  // the slice words go back in their new order: pivot rows final, every other row with its ORIGINAL word (the parallel
  // pass afterwards replays the pivots on those) -- gathered from where the rows stood at the start
  {
    const word a = __shfl(org_lo, i_lo & 63), b2 = __shfl(org_hi, i_lo & 63);
    const word lo_out = tid < rank ? pv : (i_lo < 64 ? a : b2);
    const word c2 = __shfl(org_lo, i_hi & 63), d = __shfl(org_hi, i_hi & 63);
    const word hi_out = i_hi < 64 ? c2 : d;
    if (deferred) {  // not yet: the host has the deferred columns verified first (ple_blocks)
      out->stage[tid]      = lo_out;
      out->stage[tid + 64] = hi_out;
    } else {
      if (has_lo) V[tid * vs] = lo_out;
      if (has_hi) V[(tid + 64) * vs] = hi_out;
    }
  }
  block_triangle(tid, pv, pc, rank, out);
  out->src[tid]      = i_lo;
  out->src[tid + 64] = i_hi;
  if (tid == 0) out->nsrc = (int32_t)(n < 128 ? n : 128);
  if (tid < rank) {
    out->pivcol[tid]  = pc;
    out->swaprow[tid] = (int32_t)(r0 + psw);
    out->vhigh[tid]   = ph;
    hout->pivcol[tid]  = pc;
    hout->swaprow[tid] = (int32_t)(r0 + psw);
  }
  if (tid == 0) { out->rank = rank; hout->rank = rank; out->deferred = deferred; hout->deferred = deferred; }
}

// Are the columns the one-wave search passed over really without a pivot?  Every row beyond its window replays the block's
// pivots on its slice word and looks at the deferred columns.
__global__ __launch_bounds__(ROW_THREADS) void ple_verify_kernel(const word *__restrict__ V, int64_t vs, int64_t n, PleBlock *__restrict__ blk) {
  __shared__ word s_high[64];
  __shared__ int s_col[64];
  const int rank = blk->rank;
  if (threadIdx.x < 64) {
    s_high[threadIdx.x] = blk->vhigh[threadIdx.x];
    s_col[threadIdx.x]  = blk->pivcol[threadIdx.x];
  }
  __syncthreads();
  const int64_t i = 128 + (int64_t)blockIdx.x * ROW_THREADS + threadIdx.x;
  word v = i < n ? V[i * vs] : 0;
  for (int l = 0; l < rank; ++l) v ^= ((v >> s_col[l]) & 1) ? s_high[l] : 0;
  if (__syncthreads_or((v & blk->deferred) != 0) && threadIdx.x == 0) blk->missed = 1;
}

// the verified block's first 128 slice words go where the search would have put them
__global__ __launch_bounds__(128) void ple_commit_stage_kernel(word *__restrict__ V, int64_t vs, PleBlock *__restrict__ blk) {
  if ((int)threadIdx.x < blk->nsrc) V[(int64_t)threadIdx.x * vs] = blk->stage[threadIdx.x];
  if (threadIdx.x == 0) blk->deferred = 0;  // from here on the block is an ordinary one for the kernels that follow
}

// ---- 2a. the block's row swaps on every other word; one thread per word column ---------------------------------
__global__ __launch_bounds__(ROW_THREADS) void ple_swap_rows_kernel(word *__restrict__ A, int64_t stride, int64_t width, int64_t wb, int64_t r0,
                                                                   const PleBlock *__restrict__ blk) {
  const int64_t w = (int64_t)blockIdx.x * ROW_THREADS + threadIdx.x;
  if (w >= width || w == wb) return;
  const int rank = blk->rank;
  for (int t = 0; t < rank; ++t) {
    const int64_t a = r0 + t, b = blk->swaprow[t];
    if (a == b) continue;
    const word x = A[a * stride + w], y = A[b * stride + w];
    A[a * stride + w] = y;
    A[b * stride + w] = x;
  }
}

// The same swaps as one permutation of the first <= 128 rows (what the one-wave search hands over): a workgroup per tile of
// PERM_TW word columns, the moved rows through LDS -- all loads, then all stores, instead of up to 64 dependent exchanges.
constexpr int PERM_TW = 32;
__global__ __launch_bounds__(ROW_THREADS) void ple_permute_rows_kernel(word *__restrict__ A, int64_t stride, int64_t width, int64_t wb, int64_t r0,
                                                                      const PleBlock *__restrict__ blk) {
  __shared__ word tile[128][PERM_TW];
  __shared__ int s_src[128];
  if (blk->rank <= 0 || blk->deferred) return;  // nothing moved / not confirmed yet (ple_verify_kernel; the host launches this again)
  const int n = blk->nsrc;
  if (threadIdx.x < 128) s_src[threadIdx.x] = threadIdx.x < n ? blk->src[threadIdx.x] : (int)threadIdx.x;
  __syncthreads();
  const int64_t w0 = (int64_t)blockIdx.x * PERM_TW;
  for (int e = threadIdx.x; e < n * PERM_TW; e += ROW_THREADS) {
    const int t = e / PERM_TW, w = e % PERM_TW;
    if (s_src[t] != t && w0 + w < width && w0 + w != wb) tile[t][w] = A[(r0 + s_src[t]) * stride + w0 + w];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n * PERM_TW; e += ROW_THREADS) {
    const int t = e / PERM_TW, w = e % PERM_TW;
    if (s_src[t] != t && w0 + w < width && w0 + w != wb) A[(r0 + t) * stride + w0 + w] = tile[t][w];
  }
}

// ---- 2b. the slice eliminated row by row (parallel), written back, multipliers gathered --------------------------------
// rows below the pivots: replay the block's pivots on the row's word (the multiplier of pivot l is the bit at
// its column when its turn comes, and stays there); pivot rows already hold their final word.
// The rows below would be updated with M * U, U = L^-1 U* the pivot rows after their own solve; written as
// (M L^-1) * U* the update can read the pivot rows as they are, and their solve leaves the critical path (it runs on a
// side stream, ple_blocks).  So Mc[i - rank] = the row's multipliers times L^-1, with L^-1 from the pivot search
// (block_triangle); the block's triangle L itself is copied to Lc for that solve.
__global__ __launch_bounds__(ROW_THREADS) void ple_finish_kernel(word *A, int64_t stride, int64_t nrows, int64_t r0, int64_t wb,
                                                                const word *V, int64_t vs, const PleBlock *__restrict__ blk,
                                                                word *__restrict__ Mc, word *__restrict__ Lc) {
  __shared__ word s_Linv[64], s_high[64];
  __shared__ int s_col[64];  // the block's record through LDS once: the loops below would otherwise wait for one scalar load per step
  const int64_t i = (int64_t)blockIdx.x * ROW_THREADS + threadIdx.x;  // V index
  const int rank  = blk->rank;
  if (threadIdx.x < 64) {
    s_Linv[threadIdx.x] = blk->Linv[threadIdx.x];
    s_high[threadIdx.x] = blk->vhigh[threadIdx.x];
    s_col[threadIdx.x]  = blk->pivcol[threadIdx.x];
    if (blockIdx.x == 0) Lc[threadIdx.x] = blk->Lc[threadIdx.x];  // where the side stream's solve of the pivot rows reads it
  }
  __syncthreads();
  if (rank <= 0 || blk->deferred || i >= nrows - r0) return;  // no pivot (the slice words are unchanged) or a block not confirmed yet
  word v = V[i * vs];  // the dense slice (vs = 1) or the matrix's word column itself (vs = stride: the same word this thread writes)
  if (i < rank) { A[(r0 + i) * stride + wb] = v; return; }  // pivot rows already hold their final word
  // replay the pivots; the multiplier of pivot l is the bit at its column when its turn comes (it stays there), and the
  // row's multipliers times L^-1 are the XOR of the rows of L^-1 they select
  word mt = 0;
  for (int l = 0; l < rank; ++l) {
    const bool on = (v >> s_col[l]) & 1;
    v ^= on ? s_high[l] : 0;
    mt ^= on ? s_Linv[l] : 0;
  }
  A[(r0 + i) * stride + wb] = v;
  Mc[i - rank] = mt;
}

// ---- 3. rows below, words to the right: C ^= M * U, inner dimension <= 64 -----------------------------------------------
// The trailing matrix is read and written once per 64 pivot columns: a streaming kernel, bounded by HBM (the
// C-stationary multiply leaf is built for inner dimensions of thousands and reaches a tenth of the bandwidth
// here).  A workgroup owns a tile of TW words x RU_ROWS rows.  It first builds sixteen 4-bit tables of its
// column tile from the (<= 64) pivot rows -- 16 x 16 entries x 8 TW bytes of LDS -- then streams its rows: TW/2
// lanes per row, 16 bytes each, sixteen ds_read_b128 lookups per chunk (the reference's _mzd_process_rows_ple_N
// does the same with seven 8-bit tables per 56 columns, ple_russian_template.h).
#ifndef RU_UNR
#define RU_UNR 8  // rows in flight per lane: 2 / 4 / 8 / 12 / 16 -> 115.6 / 107.0 / 99.1 / 110.0 / 147.8 ms over the 1023 launches of a 65536^2 PLE
#endif
#ifndef RU_NT
#define RU_NT 0  // bit 0: nontemporal loads of the streamed rows, bit 1: nontemporal stores (measured: tools/prof_rank_update_variants.sh)
#endif
constexpr int RU_ROWS_DEFAULT = 2048, RU_DEFAULT_TW = 32;
typedef unsigned long long __attribute__((ext_vector_type(2))) word2;

// C, U point at word column `wfirst` (even: 16-byte accesses) of the rows; words < skip_below (the block's own word when the
// tile origin had to be rounded down to stay aligned) take no update.  TW: tile width in words (TW/2 lanes per row).
template <bool VEC, int TW, int THREADS>
__global__ __launch_bounds__(THREADS) void ple_rank_update_kernel(word *__restrict__ C, int64_t c_stride, const word *__restrict__ M,
                                                                  const word *__restrict__ U, int64_t u_stride, int64_t rows, int64_t wn,
                                                                  int rank, int skip_below, int RU_ROWS, const PleBlock *__restrict__ blk) {
  __shared__ __attribute__((aligned(16))) word tab[16][16][TW];  // [table][entry][word]
  if (blk) {  // launched before the host knew the block's rank: C and rows arrive for rank 0, the record has the real one
    rank = blk->rank;
    if (rank <= 0 || blk->deferred) return;
    C += (int64_t)rank * c_stride;
    rows -= rank;
    if (rows <= 0) return;
  }
  const int tid      = threadIdx.x;
  const int64_t w0   = (int64_t)blockIdx.x * TW;
  const int64_t r_lo = (int64_t)blockIdx.y * RU_ROWS;
  const int64_t r_hi = (r_lo + RU_ROWS) < rows ? (r_lo + RU_ROWS) : rows;
  const int tw       = (int)((wn - w0) < TW ? (wn - w0) : TW);
  // the tile's pivot rows through LDS first (coalesced, every word once), then the tables from there:
  // entry e of table t = XOR of pivot rows 4t + b for the bits b of e (rows >= rank count as zero)
  __shared__ word urow[64][TW];
  for (int i = tid; i < 64 * TW; i += THREADS) {
    const int w = i % TW, r = i / TW;
    urow[r][w] = (r < rank && w < tw && (w0 + w) >= skip_below) ? U[(int64_t)r * u_stride + w0 + w] : 0;
  }
  __syncthreads();
  for (int i = tid; i < 256 * TW; i += THREADS) {
    const int w = i % TW, te = i / TW, t = te >> 4, e = te & 15;
    word x = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) x ^= ((e >> b) & 1) ? urow[4 * t + b][w] : 0;
    tab[t][e][w] = x;
  }
  __syncthreads();
  constexpr int LPR = TW / 2;          // lanes per row: 16 bytes each
  const int lane  = tid % LPR;         // 16-byte chunk of the tile row
  const int rsub  = tid / LPR;
  const bool two  = (2 * lane + 1) < tw, one = (2 * lane) < tw;
  const int ntab  = (rank + 3) >> 2;
  if (!one) return;
  // four rows per trip: their loads are issued together (one row in flight per lane leaves HBM mostly idle)
  constexpr int RSTEP = THREADS / LPR, UNR = RU_UNR;
  for (int64_t rb = r_lo + rsub; rb < r_hi; rb += (int64_t)RSTEP * UNR) {
    word m[UNR], x0[UNR], x1[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t r = rb + (int64_t)u * RSTEP;
      m[u] = 0; x0[u] = 0; x1[u] = 0;
      if (r < r_hi) {
        m[u]          = M[r];
        const word *cp = C + r * c_stride + w0 + 2 * lane;
#if RU_NT & 1
        if (VEC && two) { const word2 c = __builtin_nontemporal_load(reinterpret_cast<const word2 *>(cp)); x0[u] = c.x; x1[u] = c.y; }
#else
        if (VEC && two) { const word2 c = *reinterpret_cast<const word2 *>(cp); x0[u] = c.x; x1[u] = c.y; }
#endif
        else { x0[u] = cp[0]; x1[u] = two ? cp[1] : 0; }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll 4
      for (int t = 0; t < ntab; ++t) {
        const word2 v = *reinterpret_cast<const word2 *>(&tab[t][(m[u] >> (4 * t)) & 15][2 * lane]);
        x0[u] ^= v.x;
        x1[u] ^= v.y;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t r = rb + (int64_t)u * RSTEP;
      if (r < r_hi) {
        word *cp = C + r * c_stride + w0 + 2 * lane;
#if RU_NT & 2
        if (VEC && two) { word2 c; c.x = x0[u]; c.y = x1[u]; __builtin_nontemporal_store(c, reinterpret_cast<word2 *>(cp)); }
#else
        if (VEC && two) { word2 c; c.x = x0[u]; c.y = x1[u]; *reinterpret_cast<word2 *>(cp) = c; }
#endif
        else { cp[0] = x0[u]; if (two) cp[1] = x1[u]; }
      }
    }
  }
}

template <bool VEC>
hipError_t launch_rank_update(hipStream_t st, int variant, word *C, int64_t cs, const word *M, const word *U, int64_t us, int64_t rows, int64_t wn,
                              int rank, int skip, const PleBlock *blk = nullptr) {
  // rows per workgroup: about 2048 workgroups per launch (8 per CU), between 256 and 2048 rows in whole 128-row trips --
  // the trailing matrix shrinks from 1 GiB to nothing over a decomposition, and a fixed 2048 rows left the late
  // launches with a handful of workgroups (65536^2: 135 -> 110 ms over all 1023 launches); M4RI_AMD_RU_ROWS overrides
  static const int forced = getenv("M4RI_AMD_RU_ROWS") ? atoi(getenv("M4RI_AMD_RU_ROWS")) : 0;
  const int tw_ = variant == 64 ? 64 : variant == 32 ? 32 : 16;
  int64_t want = rows * ((wn + tw_ - 1) / tw_) / 2048;
  want = (want + 127) / 128 * 128;
  const int RU_ROWS = forced > 0 ? forced : (int)(want < 256 ? 256 : want > RU_ROWS_DEFAULT ? RU_ROWS_DEFAULT : want);
  const unsigned gy = (unsigned)((rows + RU_ROWS - 1) / RU_ROWS);
#define RU_LAUNCH(TW, TH)                                                                                                              \
  hipLaunchKernelGGL((ple_rank_update_kernel<VEC, TW, TH>), dim3((unsigned)((wn + TW - 1) / TW), gy), dim3(TH), 0, st, C, cs, M, U, us, rows, wn, \
                     rank, skip, RU_ROWS, blk)
  if (variant == 64) RU_LAUNCH(64, 1024);
  else if (variant == 32) RU_LAUNCH(32, 512);
  else RU_LAUNCH(16, 256);
#undef RU_LAUNCH
  return hipGetLastError();
}

// ---- 4. compressing L: one workgroup per row ---------------------------------------------------------------------
// new[j] = old[Q[j]] for j <= t = min(r, rank - 1); pivot columns Q[j] > t (j <= t) become 0; everything else stays.
__global__ __launch_bounds__(ROW_THREADS) void ple_compress_kernel(word *__restrict__ A, int64_t stride, int64_t nrows, int64_t width,
                                                                  const int32_t *__restrict__ Q, const word *__restrict__ pivmask, int rank) {
  extern __shared__ word s_new[];  // gathered words 0 .. t/64
  const int64_t r = blockIdx.x;
  if (r >= nrows || rank == 0) return;
  word *row       = A + r * stride;
  const int t     = (int)(r < rank - 1 ? r : rank - 1);
  const int nw    = t / 64 + 1;
  const int64_t qt = Q[t];  // pivots j <= t sit in columns <= Q[t]
  for (int w = threadIdx.x; w < nw; w += ROW_THREADS) {
    word x = 0;
    const int jend = (w * 64 + 63) < t ? 64 : (t - w * 64 + 1);
    for (int b = 0; b < jend; ++b) {
      const int32_t q = Q[w * 64 + b];
      x |= ((row[q >> 6] >> (q & 63)) & 1) << b;
    }
    s_new[w] = x;
  }
  __syncthreads();
  // vacated pivot columns: those in (t, Q[t]]
  for (int64_t w = threadIdx.x; w <= (qt >> 6) && w < width; w += ROW_THREADS) {
    word keep = ~(word)0;
    word pm   = pivmask[w];
    if (w == (qt >> 6) && (qt & 63) != 63) pm &= (((word)1 << ((qt & 63) + 1)) - 1);  // only pivots <= t, i.e. columns <= Q[t]
    if (w * 64 + 63 <= t) pm = 0;                                                      // gathered words are rewritten below
    else if (w * 64 <= t) pm &= ~(word)0 << (t - w * 64 + 1);
    keep = ~pm;
    word v = row[w] & keep;
    if (w < nw) {
      const int jend = (w * 64 + 63) < t ? 64 : (int)(t - w * 64 + 1);
      const word gm  = jend == 64 ? ~(word)0 : (((word)1 << jend) - 1);
      v = (v & ~gm) | (s_new[w] & gm);
    }
    row[w] = v;
  }
}


// ---- the column step of PLUQ: mzd_apply_p_right_trans_tri (m4ri/mzp.c:279-293) ----------------------------------------
// Row r takes the column transpositions (i, Q[i]) for i = r+1 .. ncols-1 in ascending order.  As a gather: new
// row[c] = old row[s_r[c]], and the source maps of consecutive rows differ by one transposition of VALUES,
// s_{r-1} = (r  Q[r]) o s_r  (the swap of row r-1's first step acts before all the others).  The host walks r downwards
// keeping s and its inverse, which turns every step into two writes (position, value); the device rebuilds the maps of
// a group of rows from the group's base map plus a prefix of that write list, then gathers: a wave per output word,
// lane = bit, source bits from the row's copy in LDS, the word assembled by a ballot.
constexpr int QT_THREADS = 256;

__global__ __launch_bounds__(QT_THREADS) void qtri_build_kernel(const uint32_t *__restrict__ base, uint32_t *__restrict__ S,
                                                                 uint32_t *__restrict__ base_next, int64_t ncols, const int2 *__restrict__ writes,
                                                                 const int32_t *__restrict__ cnt, int g) {
  const int i   = blockIdx.x;
  uint32_t *dst = i < g ? S + (int64_t)i * ncols : base_next;
  for (int64_t c = threadIdx.x; c < ncols; c += QT_THREADS) dst[c] = base[c];
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = cnt[i];
    for (int t = 0; t < n; ++t) dst[writes[t].x] = (uint32_t)writes[t].y;  // in order: later steps overwrite earlier ones
  }
}

template <bool LDSROW>
__global__ __launch_bounds__(QT_THREADS) void qtri_gather_kernel(word *__restrict__ A, int64_t stride, int64_t width, int64_t wend, int64_t ncols,
                                                                  int64_t row_hi, const uint32_t *__restrict__ S, const word *__restrict__ rowcopy) {  // rowcopy: rows row_hi-g+1 .. row_hi
  extern __shared__ word lrow[];
  const int i       = blockIdx.x;
  const int64_t rho = row_hi - i;
  word *row         = A + rho * stride;
  const word *src;
  if (LDSROW) {
    for (int64_t w = threadIdx.x; w < width; w += QT_THREADS) lrow[w] = row[w];
    __syncthreads();
    src = lrow;
  } else {
    src = rowcopy + (int64_t)(gridDim.x - 1 - i) * width;
  }
  const uint32_t *s = S + (int64_t)i * ncols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t w = rho / 64 + wave; w < wend; w += QT_THREADS / 64) {
    const int64_t c = w * 64 + lane;
    int bit         = 0;
    if (c < ncols) {
      const uint32_t sc = s[c];
      bit               = (int)((src[sc >> 6] >> (sc & 63)) & 1);
    }
    const word v = __ballot(bit);
    if (lane == 0) row[w] = v;
  }
}

// ---- 6. the panel step ---------------------------------------------------------------------------------------------
// The blocks of 64 columns update only the words of their own panel (PLE_PANEL_WORDS words = 2048 columns); what lies to
// the right of the panel is brought up to date once per panel:  U12 = L11^-1 * T1  on the panel's pivot rows (the TRSM of
// trsm.hip against the multipliers among the pivot rows themselves), then  T2 ^= L21 * U12  on the rows below -- one
// engine product with the multipliers where the decomposition stores them, in the panel's columns of A.  Both work in
// "panel column" coordinates: row j of Lsq / Ubuf belongs to the panel's column j, and is zero when that column has no
// pivot (rows below have zeros there as well).  The streaming rank-64 updates over the whole trailing matrix -- 366 GB
// of HBM traffic in a 65536^2 decomposition, 100 of its 154 ms -- become 128 products.
__global__ __launch_bounds__(256) void ple_panel_gather_kernel(const word *__restrict__ A, int64_t stride, int64_t pw0, int pwn, int64_t pend,
                                                               int64_t wtrail, const int32_t *__restrict__ rowmap, word *__restrict__ Lsq,
                                                               word *__restrict__ Ubuf, int64_t ustride) {
  const int j       = blockIdx.x;  // panel column
  const int32_t src = rowmap[j];
  if (threadIdx.x < pwn) {
    const int w = threadIdx.x;
    word v = 0;
    if (src >= 0) {
      word piv = 0;  // the pivot columns of word w, below column j
      for (int b = 0; b < 64; ++b)
        if (64 * w + b < j && rowmap[64 * w + b] >= 0) piv |= (word)1 << b;
      v = A[(int64_t)src * stride + pw0 + w] & piv;
    }
    Lsq[(int64_t)j * pwn + w] = v;
  }
  const word *row = src >= 0 ? A + (int64_t)src * stride + pend : nullptr;
  for (int64_t w = threadIdx.x; w < ustride; w += 256) Ubuf[(int64_t)j * ustride + w] = (row && w < wtrail) ? row[w] : 0;
}

__global__ __launch_bounds__(256) void ple_panel_scatter_kernel(word *__restrict__ A, int64_t stride, int64_t pend, int64_t wtrail,
                                                                const int32_t *__restrict__ rowmap, const word *__restrict__ Ubuf, int64_t ustride) {
  const int32_t dst = rowmap[blockIdx.x];
  if (dst < 0) return;
  word *row = A + (int64_t)dst * stride + pend;
  for (int64_t w = threadIdx.x; w < wtrail; w += 256) row[w] = Ubuf[(int64_t)blockIdx.x * ustride + w];
}

#ifndef PLE_PANEL_WORDS
#define PLE_PANEL_WORDS 32  // words of a panel: 2048 columns (16 / 24 / 32 words: 106.9 / 118.5 / 100.5 ms for mzd_ple at 65536^2, 101.5 / 97.3 / 95.5 for the flat flavour; a width that is not a power of two sits badly in the recursion's nodes) (M4RI_AMD_PLE_PANEL overrides, up to PLE_PANEL_MAX)
#endif
#define PLE_PANEL_MAX 32

// ---- per-device scratch ------------------------------------------------------------------------------------------
struct Scratch {
  word *V = nullptr, *Mc = nullptr, *Lc = nullptr, *pivmask = nullptr;
  int32_t *Q = nullptr;
  PleBlock *blk = nullptr;
  PleBlock *hblk = nullptr;  // pinned host mirror
  int *lastrow = nullptr, *hlastrow = nullptr;
  word *Lsq = nullptr, *Ubuf = nullptr;           // the panel step: multipliers among the pivot rows, the pivot rows' trailing parts
  int32_t *rowmap = nullptr, *hrowmap = nullptr;  // panel column -> its pivot's row, or -1 (device; pinned host)
  int64_t ubuf_words = 0;
  int64_t lc_blocks = 0;          // Lc holds 64 words per 64-column block of the matrix (a block's triangle is read later, on the side stream)
  hipStream_t side = nullptr;     // the pivot rows' own solves run here, off the critical path
  hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_copy = nullptr;
  int64_t rows = 0, cols = 0;
};
std::mutex g_ple_mu;
Scratch g_scratch[16];

int reserve(Scratch &s, int64_t nrows, int64_t ncols) {
  if (!s.blk) {
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.blk), sizeof(PleBlock)));
    HIPTRY(hipHostMalloc(reinterpret_cast<void **>(&s.hblk), sizeof(PleBlock), hipHostMallocDefault));
    HIPTRY(hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
    HIPTRY(hipEventCreateWithFlags(&s.ev_main, hipEventDisableTiming));
    HIPTRY(hipEventCreateWithFlags(&s.ev_side, hipEventDisableTiming));
    HIPTRY(hipEventCreateWithFlags(&s.ev_copy, hipEventDisableTiming));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.lastrow), sizeof(int)));
    HIPTRY(hipHostMalloc(reinterpret_cast<void **>(&s.hlastrow), sizeof(int), hipHostMallocDefault));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.Lsq), (size_t)PLE_PANEL_MAX * 64 * PLE_PANEL_MAX * 8));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.rowmap), (size_t)PLE_PANEL_MAX * 64 * 4));
    HIPTRY(hipHostMalloc(reinterpret_cast<void **>(&s.hrowmap), (size_t)2 * PLE_PANEL_MAX * 64 * 4, hipHostMallocDefault));  // two: see close_panel
  }
  {
    const int64_t need = (int64_t)PLE_PANEL_MAX * 64 * ((words_of(ncols) + 1) & ~(int64_t)1);
    if (need > s.ubuf_words) {
      if (s.Ubuf) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.Ubuf)); }
      s.Ubuf = nullptr; s.ubuf_words = 0;
      HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.Ubuf), (size_t)need * 8));
      s.ubuf_words = need;
    }
  }
  if (nrows > s.rows) {
    if (s.V) { HIPTRY(hipFree(s.V)); HIPTRY(hipFree(s.Mc)); }
    s.V = s.Mc = nullptr; s.rows = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.V), (size_t)nrows * 8));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.Mc), (size_t)nrows * 8));
    s.rows = nrows;
  }
  if (words_of(ncols) > s.lc_blocks) {
    if (s.Lc) { HIPTRY(hipDeviceSynchronize()); HIPTRY(hipFree(s.Lc)); }
    s.Lc = nullptr; s.lc_blocks = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.Lc), (size_t)words_of(ncols) * 64 * 8));
    s.lc_blocks = words_of(ncols);
  }
  if (ncols > s.cols) {
    if (s.Q) { HIPTRY(hipFree(s.Q)); HIPTRY(hipFree(s.pivmask)); }
    s.Q = nullptr; s.pivmask = nullptr; s.cols = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.Q), (size_t)ncols * 4));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&s.pivmask), (size_t)words_of(ncols) * 8));
    s.cols = ncols;
  }
  return 0;
}

// ---- the driver ---------------------------------------------------------------------------------------------------
struct PleRun {
  word *A;
  int64_t stride, nrows, ncols, width;
  int32_t *P, *Q;
  hipStream_t st;
  Scratch *s;
  int64_t r0;      // rows finished so far = pivots found so far
  int64_t cutoff;  // __M4RI_PLE_CUTOFF of the reference build being matched (words); 0: no recursion
  bool side_used = false;
  // the open panel (see "the panel step"): first word, one past its last word, R.r0 when it was opened, and which of the two
  // host row maps it fills (the other one may still be on its way to the device).  It stays open across the nodes of the
  // recursion of _mzd_ple as long as they stay inside it.
  int64_t pw0 = -1, pend = 0, prow0 = 0;
  int which = 0;
};

// bring the words to the right of the finished panel up to date (see "the panel step" above)
int close_panel(PleRun &R) {
  if (R.pw0 < 0) return 0;
  Scratch &s = *R.s;
  word *A    = R.A;
  hipStream_t st = R.st;
  const int64_t stride = R.stride, p0 = R.pw0, pend = R.pend, rp = R.r0 - R.prow0, pwn = pend - p0, wtrail = R.width - pend;
  R.pw0 = -1;
  if (rp == 0 || wtrail <= 0) return 0;
  const int64_t ustride = (wtrail + 1) & ~(int64_t)1, trailcols = R.ncols - pend * 64;
  if (R.side_used) {  // the pivot rows' solves inside the panel write words the gather reads around
    HIPTRY(hipEventRecord(s.ev_side, s.side));
    HIPTRY(hipStreamWaitEvent(st, s.ev_side, 0));
  }
  HIPTRY(hipMemcpyAsync(s.rowmap, s.hrowmap + R.which * PLE_PANEL_MAX * 64, (size_t)pwn * 64 * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ple_panel_gather_kernel, dim3((unsigned)(pwn * 64)), dim3(256), 0, st, A, stride, p0, (int)pwn, pend, wtrail, s.rowmap, s.Lsq, s.Ubuf,
                     ustride);
  HIPTRY(hipGetLastError());
  HIPTRY(m4ri_amd_trsm_lower_left_dev(s.Lsq, pwn, s.Ubuf, ustride, pwn * 64, trailcols, 0, st));
  hipLaunchKernelGGL(ple_panel_scatter_kernel, dim3((unsigned)(pwn * 64)), dim3(256), 0, st, A, stride, pend, wtrail, s.rowmap, s.Ubuf, ustride);
  HIPTRY(hipGetLastError());
  if (R.nrows - R.r0 > 0)
    HIPTRY(m4ri_amd_mul_dev(A + R.r0 * stride + pend, stride, A + R.r0 * stride + p0, stride, s.Ubuf, ustride, R.nrows - R.r0, pwn * 64, trailcols, 1, 0, st));
  // the next panel fills the OTHER host map; this one is reused a panel later, by when at least one block's record has been
  // waited for that follows this copy in the stream
  R.which ^= 1;
  return 0;
}

// Columns [c0, c1) (c0 on a word boundary) in blocks of 64: the t-th pivot found goes to Q[c0 + t].
int ple_blocks(PleRun &R, int64_t c0, int64_t c1, int64_t *found) {
  Scratch &s = *R.s;
  word *A = R.A;
  const int64_t stride = R.stride, nrows = R.nrows, ncols = R.ncols, width = R.width, first = R.r0;
  hipStream_t st = R.st;
  const bool vec = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && stride % 2 == 0;
  static const bool wave_first = !(getenv("M4RI_AMD_PLE_WAVE") && atoi(getenv("M4RI_AMD_PLE_WAVE")) == 0);
  static const int variant     = getenv("M4RI_AMD_RU_TW") ? atoi(getenv("M4RI_AMD_RU_TW")) : RU_DEFAULT_TW;
  // panels pay from about 150 MiB of matrix on (a pass over the trailing matrix per block then costs more than the panel
  // step's fixed ~0.25 ms per 16 blocks): 16384^2 16.2 ms without / 19.1 with, 32768^2 41.1 / 41.4, 65536^2 155 / 106;
  // M4RI_AMD_PLE_PANELS=0 / 1 forces them off / on
  static const int panels_env = getenv("M4RI_AMD_PLE_PANELS") ? atoi(getenv("M4RI_AMD_PLE_PANELS")) : -1;
  const bool panels = panels_env >= 0 ? panels_env != 0 : nrows * width >= ((int64_t)3 << 23);
  static const int64_t panel_words = [] {
    const char *e = getenv("M4RI_AMD_PLE_PANEL");
    const int v   = e ? atoi(e) : PLE_PANEL_WORDS;
    return (int64_t)(v < 1 ? 1 : v > PLE_PANEL_MAX ? PLE_PANEL_MAX : v);
  }();
  for (int64_t wb = c0 / 64; wb * 64 < c1 && R.r0 < nrows; ++wb) {
    if (panels && (R.pw0 < 0 || wb >= R.pend)) {
      if (int rc = close_panel(R)) return rc;
      R.pw0   = wb;
      R.pend  = (wb / panel_words + 1) * panel_words;  // panels end on multiples of panel_words, whatever run of columns is being walked
      if (R.pend > width) R.pend = width;
      R.prow0 = R.r0;
      int32_t *hm = s.hrowmap + R.which * PLE_PANEL_MAX * 64;
      for (int64_t j = 0; j < (R.pend - R.pw0) * 64; ++j) hm[j] = -1;
    }
    const int64_t pend = panels ? R.pend : width;  // how far this block's update and its pivot rows' solve reach
    const int64_t r0 = R.r0;
    const int ncb = (int)((c1 - wb * 64) < 64 ? (c1 - wb * 64) : 64);
    const int64_t nleft = nrows - r0;
    word *Lc = s.Lc + wb * 64;
    const unsigned row_grid = (unsigned)((nleft + ROW_THREADS - 1) / ROW_THREADS);
    // rows below, words to the right: C ^= (M L^-1) * U*, inner dimension = the block's rank, U* the pivot rows as they are;
    // dev_rank: the kernel takes the rank from the block's record (it is launched before the host has read it)
    auto update = [&](int rank, bool dev_rank) -> int {
      if (wb + 1 >= pend || nleft - rank <= 0) return 0;
      const int64_t wfirst = vec ? ((wb + 1) & ~(int64_t)1) : (wb + 1);  // even tile origin; may take in word wb itself
      const int64_t wn     = pend - wfirst;                              // to the end of the panel (of the matrix without panels)
      const int skip       = (int)(wb + 1 - wfirst);
      word *C              = A + (r0 + rank) * stride + wfirst;
      const word *U        = A + r0 * stride + wfirst;
      if (vec) HIPTRY(launch_rank_update<true>(st, variant, C, stride, s.Mc, U, stride, nleft - rank, wn, rank, skip, dev_rank ? s.blk : nullptr));
      else HIPTRY(launch_rank_update<false>(st, variant, C, stride, s.Mc, U, stride, nleft - rank, wn, rank, skip, dev_rank ? s.blk : nullptr));
      return 0;
    };
    int rank = 0;
    bool done = false;
    word *col = A + r0 * stride + wb;  // the block's word of the rows from r0 on, in place
    if (wave_first) {
      // The one-wave search, its record on the way to the host, and -- without waiting for it -- everything that follows
      // from the record on the device: row moves, the slice pass, the trailing update.  The host only waits for the copy;
      // by the time it has the rank and queues the next block, the device is still busy with this one.
      hipLaunchKernelGGL(ple_pivots_wave_kernel, dim3(1), dim3(64), 0, st, nleft, r0, ncb, col, stride, s.blk, s.hblk);
      HIPTRY(hipGetLastError());
      HIPTRY(hipEventRecord(s.ev_copy, st));
      hipLaunchKernelGGL(ple_permute_rows_kernel, dim3((unsigned)((width + PERM_TW - 1) / PERM_TW)), dim3(ROW_THREADS), 0, st, A, stride, width, wb, r0, s.blk);
      hipLaunchKernelGGL(ple_finish_kernel, dim3(row_grid), dim3(ROW_THREADS), 0, st, A, stride, nrows, r0, wb, col, stride, s.blk, s.Mc, Lc);
      HIPTRY(hipGetLastError());
      if (int rc = update(0, true)) return rc;
      HIPTRY(hipEventRecord(s.ev_main, st));
      HIPTRY(hipEventSynchronize(s.ev_copy));
      rank = s.hblk->rank;
      done = true;
      if (s.hblk->deferred != 0) {
        // The search passed over columns in which none of its 128 rows had a pivot (the kernels queued above did nothing):
        // have the rows beyond looked at.  Nothing there -- the usual case, a column without a pivot -- and the block is
        // committed as found; otherwise the slice is untouched and the general search takes over.
        HIPTRY(hipMemsetAsync(&s.blk->missed, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(ple_verify_kernel, dim3((unsigned)((nleft - 128 + ROW_THREADS - 1) / ROW_THREADS)), dim3(ROW_THREADS), 0, st, col, stride, nleft, s.blk);
        HIPTRY(hipGetLastError());
        HIPTRY(hipMemcpyAsync(&s.hblk->missed, &s.blk->missed, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPTRY(hipStreamSynchronize(st));
        if (s.hblk->missed == 0) {
          hipLaunchKernelGGL(ple_commit_stage_kernel, dim3(1), dim3(128), 0, st, col, stride, s.blk);
          if (rank > 0) {
            hipLaunchKernelGGL(ple_permute_rows_kernel, dim3((unsigned)((width + PERM_TW - 1) / PERM_TW)), dim3(ROW_THREADS), 0, st, A, stride, width, wb, r0,
                               s.blk);
            hipLaunchKernelGGL(ple_finish_kernel, dim3(row_grid), dim3(ROW_THREADS), 0, st, A, stride, nrows, r0, wb, col, stride, s.blk, s.Mc, Lc);
            HIPTRY(hipGetLastError());
            if (int rc = update(rank, false)) return rc;
            HIPTRY(hipEventRecord(s.ev_main, st));
          }
        } else {
          done = false;
        }
      }
    }
    if (!done) {  // the general search works on a dense copy of the slice
      hipLaunchKernelGGL(ple_extract_kernel, dim3(row_grid), dim3(ROW_THREADS), 0, st, A, stride, nrows, r0, wb, s.V);
      hipLaunchKernelGGL(ple_pivots_kernel, dim3(1), dim3(SLICE_THREADS), 0, st, nleft, r0, ncb, s.V, s.blk);
      HIPTRY(hipGetLastError());
      HIPTRY(hipMemcpyAsync(s.hblk, s.blk, sizeof(PleBlock), hipMemcpyDeviceToHost, st));
      HIPTRY(hipStreamSynchronize(st));
      rank = s.hblk->rank;
      if (rank > 0) {
        hipLaunchKernelGGL(ple_swap_rows_kernel, dim3((unsigned)((width + ROW_THREADS - 1) / ROW_THREADS)), dim3(ROW_THREADS), 0, st, A, stride, width, wb, r0,
                           s.blk);
        hipLaunchKernelGGL(ple_finish_kernel, dim3(row_grid), dim3(ROW_THREADS), 0, st, A, stride, nrows, r0, wb, s.V, (int64_t)1, s.blk, s.Mc, Lc);
        HIPTRY(hipGetLastError());
        if (int rc = update(rank, false)) return rc;
        HIPTRY(hipEventRecord(s.ev_main, st));
      }
    }
    if (rank == 0) continue;  // nothing moved: the slice words are unchanged
    if (wb + 1 < pend) {
      // the pivot rows among themselves on the words to the right (ple_russian.c:306-325), as far as the panel goes: a
      // unit lower triangular solve with the <= 64 x 64 triangle of their multipliers -- after the update has read them
      // (ev_main), on the side stream: no later block looks at these words again
      const int64_t cend = pend * 64 < ncols ? pend * 64 : ncols;
      HIPTRY(hipStreamWaitEvent(s.side, s.ev_main, 0));
      HIPTRY(m4ri_amd_trsm_lower_left_dev(Lc, 1, A + r0 * stride + wb + 1, stride, rank, cend - (wb + 1) * 64, 0, s.side));
      R.side_used = true;
    }
    for (int t = 0; t < rank; ++t) {
      R.P[r0 + t]                = s.hblk->swaprow[t];
      R.Q[c0 + (r0 - first) + t] = (int32_t)(wb * 64 + s.hblk->pivcol[t]);
      if (R.pw0 >= 0) s.hrowmap[R.which * PLE_PANEL_MAX * 64 + (wb - R.pw0) * 64 + s.hblk->pivcol[t]] = (int32_t)(r0 + t);
    }
    R.r0 += rank;
  }
  *found = R.r0 - first;  // the panel stays open: the next run of columns may continue it (closed by ple_rec / the driver)
  return 0;
}

// mzd_first_zero_row (mzd.c:1826-1841) of rows [row0, row0 + R) x words [w0, w1): one past the last row with a set bit.
// Workgroup b takes the b-th chunk of 256 rows from the bottom and leaves at once when a chunk below it (an earlier
// workgroup) has already reported; a wave per row, lanes across the words.
__global__ __launch_bounds__(256) void ple_last_row_kernel(const word *__restrict__ A, int64_t stride, int64_t row0, int64_t R, int64_t w0, int64_t w1,
                                                           int *out) {
  const int64_t hi = R - (int64_t)blockIdx.x * 256, lo = hi > 256 ? hi - 256 : 0;
  if (*(volatile int *)out >= hi) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t rel = hi - 1 - wave; rel >= lo; rel -= 4) {
    const word *row = A + (row0 + rel) * stride;
    bool any = false;
    for (int64_t w = w0; w < w1 && !any; w += 64) {
      const word v = (w + lane < w1) ? row[w + lane] : 0;
      any = __ballot(v != 0) != 0;
    }
    if (any) {
      if (lane == 0) atomicMax(out, (int)(rel + 1));
      return;  // the wave's remaining rows lie further up in the matrix (smaller rel): nothing more to add
    }
  }
}

// The recursion of _mzd_ple (ple.c:62-171) as far as it can be seen from outside.  The elimination itself runs in blocks
// of 64 columns, left to right, always updating every column to the right, so when the walk reaches a node of the
// reference's recursion tree the node's window already holds the Schur complement the reference would see; what the
// tree decides is only the contents of Q BEHIND the rank: a node resets its part of Q to the identity (:68), a window
// without a set bit returns at once (:66-69), windows of <= 64 columns or <= cutoff words are one flat run (:74-81),
// the others split their columns in halves (:96) and afterwards copy the right half's pivots down behind the left
// half's (:146) while the right half's own entries stay.  rows: the node's A->nrows.
int ple_rec(PleRun &R, int64_t rows, int64_t c0, int64_t c1, int64_t *found) {
  const int64_t ncols = c1 - c0, width = words_of(ncols);
  for (int64_t c = c0; c < c1; ++c) R.Q[c] = (int32_t)c;
  *found = 0;
  if (rows <= 0) return 0;
  Scratch &s = *R.s;
  if (R.pw0 >= 0 && c0 / 64 + width > R.pend)  // the window reaches beyond the open panel, where the panel's update is still due
    if (int rc = close_panel(R)) return rc;
  HIPTRY(hipMemsetAsync(s.lastrow, 0, sizeof(int), R.st));
  hipLaunchKernelGGL(ple_last_row_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, R.st, R.A, R.stride, R.r0, rows, c0 / 64, c0 / 64 + width,
                     s.lastrow);
  HIPTRY(hipGetLastError());
  HIPTRY(hipMemcpyAsync(s.hlastrow, s.lastrow, sizeof(int), hipMemcpyDeviceToHost, R.st));
  HIPTRY(hipStreamSynchronize(R.st));
  const int64_t e = *s.hlastrow;
  if (e == 0) return 0;
  if (ncols <= 64 || width * rows <= R.cutoff) return ple_blocks(R, c0, c1, found);
  const int64_t n1 = (((ncols - 1) / 64 + 1) >> 1) * 64;
  int64_t r1 = 0, r2 = 0;
  if (int rc = ple_rec(R, e, c0, c0 + n1, &r1)) return rc;
  if (int rc = ple_rec(R, e - r1, c0 + n1, c1, &r2)) return rc;
  for (int64_t t = 0; t < r2; ++t) R.Q[c0 + r1 + t] = R.Q[c0 + n1 + t];
  *found = r1 + r2;
  return 0;
}

}  // namespace

extern "C" {

// PLE of the device matrix A (nrows x ncols, bits at column >= ncols zero) in place.  P (nrows entries) and Q
// (ncols entries) are HOST arrays; returns the rank in *rank_out.  Blocking (reads the pivots back per block).
// recursion_cutoff == 0: _mzd_ple_russian's Q (identity behind the rank); > 0: _mzd_ple's (ple.c:62-171), whose column
// halving leaves other values behind the rank -- see ple_rec above.
int m4ri_amd_ple_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, int32_t *P, int32_t *Q, int32_t *rank_out,
                     int64_t recursion_cutoff, void *stream) {
  if (nrows < 0 || ncols < 0 || !P || !Q || !rank_out || recursion_cutoff < 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i = 0; i < nrows; ++i) P[i] = (int32_t)i;  // ple_russian.c:412-414
  for (int64_t j = 0; j < ncols; ++j) Q[j] = (int32_t)j;
  *rank_out = 0;
  if (nrows == 0 || ncols == 0) return 0;
  std::lock_guard<std::mutex> lk(g_ple_mu);
  int dev = 0;
  HIPTRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  Scratch &s = g_scratch[dev];
  if (int rc = reserve(s, nrows, ncols)) return rc;
  const int64_t width = words_of(ncols);
  PleRun run{A, stride, nrows, ncols, width, P, Q, st, &s, 0, recursion_cutoff};
  int64_t found = 0;
  int rc_f = recursion_cutoff ? ple_rec(run, nrows, 0, ncols, &found) : ple_blocks(run, 0, ncols, &found);
  if (rc_f == 0) rc_f = close_panel(run);
  if (run.side_used) {  // the pivot rows' solves join here, also on an error path: the side stream works on the caller's matrix
    HIPTRY(hipEventRecord(s.ev_side, s.side));
    HIPTRY(hipStreamWaitEvent(st, s.ev_side, 0));
  }
  if (rc_f) return rc_f;
  const int rank = (int)run.r0;
  *rank_out      = rank;
  if (rank > 0) {
    std::vector<word> pm((size_t)width, 0);
    bool identity = true;
    for (int j = 0; j < rank; ++j) { pm[(size_t)(Q[j] >> 6)] |= (word)1 << (Q[j] & 63); identity = identity && Q[j] == j; }
    if (!identity) {  // pivots on the diagonal: L already sits where it belongs
      HIPTRY(hipMemcpyAsync(s.Q, Q, (size_t)rank * 4, hipMemcpyHostToDevice, st));
      HIPTRY(hipMemcpyAsync(s.pivmask, pm.data(), (size_t)width * 8, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(ple_compress_kernel, dim3((unsigned)nrows), dim3(ROW_THREADS), (size_t)((rank - 1) / 64 + 1) * 8, st, A, stride, nrows, width,
                         s.Q, s.pivmask, rank);
      HIPTRY(hipGetLastError());
      HIPTRY(hipStreamSynchronize(st));  // pm / Q are host temporaries of this call
    }
  }
  return 0;
}

// The column step of PLUQ on a device matrix: row r <- its columns under the transpositions (i, Q[i]), i > r, ascending
// (mzd_apply_p_right_trans_tri, mzp.c:279-293).  Q: HOST array of ncols entries with Q[i] >= i.  Blocking.
int m4ri_amd_apply_p_right_trans_tri_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, const int32_t *Q, void *stream) {
  if (nrows < 0 || ncols < 0 || !Q) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  int64_t top = -1, cmax = 0;
  for (int64_t i = 0; i < ncols; ++i) {
    if (Q[i] < i || Q[i] >= ncols) return (int)hipErrorInvalidValue;
    if (Q[i] != i) { top = i; if (Q[i] > cmax) cmax = Q[i]; }
  }
  const int64_t rows = nrows < top ? nrows : top;  // rows >= top take identity swaps only
  if (rows <= 0) return 0;
  const int64_t width = words_of(ncols), wend = cmax / 64 + 1;
  std::vector<int32_t> s((size_t)ncols), inv((size_t)ncols);
  for (int64_t c = 0; c < ncols; ++c) s[(size_t)c] = inv[(size_t)c] = (int32_t)c;
  std::vector<int2> writes;
  auto step = [&](int64_t k) {  // s <- (k Q[k]) o s
    const int32_t a = (int32_t)k, b = Q[k];
    if (a == b) return;
    const int32_t pa = inv[(size_t)a], pb = inv[(size_t)b];
    s[(size_t)pa] = b; s[(size_t)pb] = a;
    inv[(size_t)a] = pb; inv[(size_t)b] = pa;
    writes.push_back(make_int2(pa, b));
    writes.push_back(make_int2(pb, a));
  };
  for (int64_t k = top; k > rows - 1; --k) step(k);  // s = the map of row rows-1, the first one that exists
  writes.clear();
  int64_t G = ((int64_t)1 << 24) / ncols;
  G = G < 16 ? 16 : (G > 256 ? 256 : G);
  struct Group { int64_t hi; int g; size_t woff, coff; };
  std::vector<Group> groups;
  std::vector<int32_t> cnts;
  std::vector<int32_t> base0(s);  // map of the first group's top row
  for (int64_t hi = rows - 1; hi >= 0; hi -= G) {
    const int g = (int)(hi + 1 < G ? hi + 1 : G);
    Group gr{hi, g, writes.size(), cnts.size()};
    for (int i = 0; i <= g; ++i) {  // entry i: writes that take the base to row hi - i (entry g: to the next group's base)
      if (i > 0) step(hi - i + 1);
      cnts.push_back((int32_t)(writes.size() - gr.woff));
    }
    groups.push_back(gr);
  }
  uint32_t *d_base[2] = {nullptr, nullptr}, *d_S = nullptr;
  int2 *d_writes = nullptr;
  int32_t *d_cnt = nullptr;
  word *d_rowcopy = nullptr;
  const bool ldsrow = width * 8 <= 64 * 1024;
  int rc = 0;
  auto run = [&]() -> int {
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_base[0]), (size_t)ncols * 4));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_base[1]), (size_t)ncols * 4));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_S), (size_t)G * ncols * 4));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_writes), (writes.size() + 1) * sizeof(int2)));
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_cnt), cnts.size() * 4));
    if (!ldsrow) HIPTRY(hipMalloc(reinterpret_cast<void **>(&d_rowcopy), (size_t)G * width * 8));
    HIPTRY(hipMemcpyAsync(d_base[0], base0.data(), (size_t)ncols * 4, hipMemcpyHostToDevice, st));
    if (!writes.empty()) HIPTRY(hipMemcpyAsync(d_writes, writes.data(), writes.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    HIPTRY(hipMemcpyAsync(d_cnt, cnts.data(), cnts.size() * 4, hipMemcpyHostToDevice, st));
    int cur = 0;
    for (const Group &gr : groups) {
      hipLaunchKernelGGL(qtri_build_kernel, dim3((unsigned)gr.g + 1), dim3(QT_THREADS), 0, st, d_base[cur], d_S, d_base[cur ^ 1], ncols,
                         d_writes + gr.woff, d_cnt + gr.coff, gr.g);
      if (ldsrow) {
        hipLaunchKernelGGL((qtri_gather_kernel<true>), dim3((unsigned)gr.g), dim3(QT_THREADS), (size_t)width * 8, st, A, stride, width, wend, ncols, gr.hi,
                           d_S, nullptr);
      } else {
        const word *lo = A + (gr.hi - gr.g + 1) * stride;
        HIPTRY(hipMemcpy2DAsync(d_rowcopy, (size_t)width * 8, lo, (size_t)stride * 8, (size_t)width * 8, (size_t)gr.g, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL((qtri_gather_kernel<false>), dim3((unsigned)gr.g), dim3(QT_THREADS), 0, st, A, stride, width, wend, ncols, gr.hi, d_S,
                           d_rowcopy);
      }
      HIPTRY(hipGetLastError());
      cur ^= 1;
    }
    HIPTRY(hipStreamSynchronize(st));
    return 0;
  };
  rc = run();
  for (void *p : {(void *)d_base[0], (void *)d_base[1], (void *)d_S, (void *)d_writes, (void *)d_cnt, (void *)d_rowcopy})
    if (p) (void)hipFree(p);
  return rc;
}

// PLUQ in place (ple.c:50-60): the PLE above, then the column step on the first `rank` rows (all rows when the rank
// is 0 or full).
int m4ri_amd_pluq_dev(word *A, int64_t stride, int64_t nrows, int64_t ncols, int32_t *P, int32_t *Q, int32_t *rank_out,
                      int64_t recursion_cutoff, void *stream) {
  if (int rc = m4ri_amd_ple_dev(A, stride, nrows, ncols, P, Q, rank_out, recursion_cutoff, stream)) return rc;
  const int32_t r = *rank_out;
  return m4ri_amd_apply_p_right_trans_tri_dev(A, stride, (r && r < nrows) ? r : nrows, ncols, Q, stream);
}

}  // extern "C"
