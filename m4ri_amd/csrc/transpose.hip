// transpose.hip -- D = A^T for bit-packed matrices (the device side of mzd_transpose, /root/reference m4ri/mzd.c:1104-1139;
// the reference works through 64 x 64 blocks with masked swaps on the CPU, mzd.c:700-1100).
//
// Pure data movement: every word of A is read once and every word of D written once, so the kernel is bounded by HBM
// (algorithmic bytes = 8 * (nrows * W(ncols) + ncols * W(nrows))).  The shape of the work is what needs care: a 64 x 64
// bit block sits in ONE word per row, so reading a block the way it is transposed (a lane per row) fetches 8 bytes out
// of every 128-byte line, and so does writing it.  Both sides are made whole lines:
//   * a workgroup owns a tile of 1024 rows x 16 words (1024 x 1024 bits).  It walks the tile in 16 groups of 64 rows;
//     a group's 64 x 16 words are loaded along the rows (128 contiguous bytes per row, branch-free, TR_PREFETCH groups
//     ahead in registers) and staged in LDS (double-buffered, one barrier per group);
//   * wave w owns the word columns 2w, 2w+1: it takes its two 64 x 64 blocks of the group out of LDS (one ds_read_b128
//     per lane) and transposes them in registers -- six exchange stages, none through LDS (transpose_block below);
//   * every 8 groups lane j of the wave holds 8 consecutive words (64 bytes) of row 64 (2w + k) + j of the result; they
//     go through a wave-private LDS block so that a store instruction writes 16 x 64 contiguous bytes, the second half
//     of each 128-byte line a few microseconds after the first from the same CU.
// What the measurements said on the way (65536 x 65536, 1.07 GB of traffic, tools/transpose_probe.hip for the timelines):
//   2.6 TB/s  first version: ds_bpermute exchanges, loads behind bounds branches (waited for one by one), results
//             stored as 16-byte pieces from 64 accumulator registers; 181 registers = one workgroup per CU
//   2.6       loads branch-free and 4 groups ahead: no change -- the group loop was waiting for its own six dependent
//             LDS round trips per block, and a wave64 VALU operation issues over 4 cycles here
//   1.4       a whole tile of loads in flight (persistent workgroups): the array of loads went to scratch
//   2.7       DPP / v_permlane*_swap exchanges: the group loop is VALU-bound (0.8 us per group), but the kernel still
//             waits: 3 us to issue a half tile's stores (64 requests of 16 bytes per instruction)
//   3.4 .. 3.5  results through LDS (16 requests of 64 bytes), 8-group halves (32 accumulators, 112 registers: two
//             workgroups per CU); deeper prefetch (4, 6, 8 groups) and other tile orders change nothing any more: what
//             is left is what HBM gives 128-byte reads at an 8 KB stride against 64-byte writes (the same box copies
//             at 4.7 .. 5.2 TB/s); 16384^2 and 32768^2, which the caches help, run at 4.2 and 5.5 TB/s.
#include <hip/hip_runtime.h>
#include "gf2_common.h"

#ifdef TR_TIMING  // tools/transpose_probe.hip: a block's timeline in 100 MHz ticks
__device__ unsigned long long tr_times[8192][6];
#define TSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) tr_times[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

namespace {

#ifndef TR_WORDS
#define TR_WORDS 16                       // tile width in words of A: 128 bytes per row (32 = 256 bytes, 16 waves: 3.43 vs 3.24 TB/s at 65536^2 but 5.3 vs 5.9 at 32768^2)
#endif
constexpr int TR_THREADS = 32 * TR_WORDS;  // a wave per two word columns
constexpr int TR_GROUPS  = 16;   // 64-row groups per tile: 128 bytes per row of D
#ifndef TR_PREFETCH
#define TR_PREFETCH 3
#endif
#ifndef TR_WAVES_PER_EU
#define TR_WAVES_PER_EU 4
#endif
constexpr int TR_PITCH   = TR_WORDS + 2;  // LDS row pitch in words: 16-byte aligned, 36 (68) banks apart (b128 reads conflict-free)

// lane i holds row i of a 64 x 64 bit block in (lo, hi); on return lane j holds column j.  Six exchange stages (lane
// distance = bit distance = 32, 16, ..., 1), none of them through LDS -- with ds_bpermute the six dependent LDS round
// trips per block were what the kernel waited for:
//   32: lanes >= 32 of lo <-> lanes < 32 of hi: one v_permlane32_swap;
//   16: the low halves of both dwords in one register, the high halves in another (v_perm), v_permlane16_swap;
//   8: the same with bytes, two v_perm from the previous form, the exchange a DPP row_ror:8 and three selects, two v_perm back;
//   4, 2, 1: a lane keeps the bits K of its dwords (M for the lower lane of a pair, ~M for the upper one) and hands
//       the others to its partner, both dwords' worth packed into one dword that moves by DPP (row_half_mirror
//       + quad_perm[3,2,1,0] = xor 4; quad_perm for 2 and 1); branch-free, the two roles differ in K and two shift counts.
__device__ __forceinline__ uint32_t dpp_xor8(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_xor4(uint32_t v) {
  const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror: i -> i ^ 7
  return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, true);     // quad_perm [3,2,1,0]: i -> i ^ 3
}
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }

__device__ __forceinline__ void transpose_block(uint32_t &lo, uint32_t &hi, int lane) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    lo = r[0]; hi = r[1];
  }
  {
    // distance 16: y = the low halves of (lo, hi), x = the high halves; the lower lane of a pair keeps y and gets its
    // partner's y as its new x, the upper lane keeps x and gets its partner's x as its new y
    uint32_t y = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    uint32_t x = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
    const auto r = __builtin_amdgcn_permlane16_swap(y, x, false, false);  // odd rows of y <-> even rows of x
    y = r[0]; x = r[1];
    // distance 8, straight from that form: y8 = the even bytes of the (lo, hi) that (y, x) stand for, x8 = the odd ones
    const uint32_t y8 = __builtin_amdgcn_perm(x, y, 0x06020400u), x8 = __builtin_amdgcn_perm(x, y, 0x07030501u);
    const bool up = (lane & 8) != 0;
    const uint32_t got = dpp_xor8(up ? y8 : x8);
    const uint32_t ny = up ? got : y8, nx = up ? x8 : got;
    lo = __builtin_amdgcn_perm(nx, ny, 0x05010400u);
    hi = __builtin_amdgcn_perm(nx, ny, 0x07030602u);
  }
#define TR_STAGE(D, M, XCHG)                                                           \
  {                                                                                    \
    const uint32_t up = (uint32_t)lane & (D), K = up ? ~(M) : (M);                     \
    const uint32_t shr = (D) - up, shl = up; /* (D, 0) for the lower lane, (0, D) for the upper */ \
    const uint32_t s  = ((lo & ~K) >> shr) | ((hi & ~K) << shl);                       \
    const uint32_t r  = XCHG(s);                                                       \
    lo = (lo & K) | ((r & (M)) << shr);                                                \
    hi = (hi & K) | ((r & ~(M)) >> shl);                                               \
  }
  TR_STAGE(4, 0x0f0f0f0fu, dpp_xor4)
  TR_STAGE(2, 0x33333333u, dpp_xor2)
  TR_STAGE(1, 0x55555555u, dpp_xor1)
#undef TR_STAGE
}

// VEC / AVEC: the rows of D / of A are 16-byte aligned (even stride, aligned base) -> 16-byte stores / loads.
// The loads are branch-free and unmasked: addresses are clamped into the matrix (a load behind a branch would be waited
// for on the spot, and the kernel lives on having many in flight), and what a clamped or ragged load brings in beyond
// A's rows and columns ends up either in rows of D that do not exist (never stored) or in the bits of D's last word
// beyond column nrows (masked at the store).
template <bool VEC, bool AVEC>
__global__ __launch_bounds__(TR_THREADS) __attribute__((amdgpu_waves_per_eu(TR_WAVES_PER_EU))) void transpose_kernel(word *__restrict__ D, int64_t d_stride, const word *__restrict__ A,
                                                               int64_t a_stride, int64_t nrows, int64_t ncols, int64_t tiles_r,
                                                               int64_t ntiles) {
  __shared__ __attribute__((aligned(16))) word stage[2][64][TR_PITCH];
  __shared__ __attribute__((aligned(16))) word ostage[TR_THREADS / 64][64][TR_GROUPS / 2];
  const int64_t wa = (ncols + 63) >> 6, wd = (nrows + 63) >> 6;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = t / (TR_WORDS / 2), lw = (t % (TR_WORDS / 2)) * 2;  // this thread's row within a group and its pair of words
  const word dtail = (nrows & 63) ? ((~(word)0) >> (64 - (nrows & 63))) : ~(word)0;

  // a tile's load addresses: first row, and the thread's (clamped) word offsets
  struct Src { int64_t r0, c0, c1; };
  auto source = [&](int64_t tile) {
    Src s;
    const int64_t tr = tile % tiles_r, tc = tile / tiles_r;  // row tiles fastest: neighbouring workgroups write neighbouring
                                                            // pieces of the same rows of D (the other orders tried -- tile
                                                            // columns fastest, 2 / 4 / 8 / 16-wide super-tiles -- are slower)
    s.r0 = tr * 64 * TR_GROUPS;
    const int64_t w = tc * TR_WORDS + lw;
    s.c0 = AVEC ? (w < a_stride - 2 ? w : a_stride - 2) : (w < wa ? w : wa - 1);
    s.c1 = (w + 1 < wa) ? w + 1 : wa - 1;
    return s;
  };
  // TR_PREFETCH groups of loads are in flight ahead of the group being transposed.  The row pointer advances group by
  // group (and is made opaque to the compiler, which otherwise computes all sixteen up front and holds them).
  const int64_t tile = blockIdx.x;
  TSTAMP(0);
  const Src s0 = source(tile);
  const word *rowp  = A + (s0.r0 + lr) * a_stride + s0.c0;
  const word *lastp = A + (nrows - 1) * a_stride + s0.c0;
  const int64_t rows_left = nrows - (s0.r0 + lr);  // group g's row exists iff rows_left > 64 g
  const int64_t d1 = s0.c1 - s0.c0, step = 64 * a_stride;
  int fetched = 0;
  auto fetch = [&]() -> uint4 {
    const word *q = (rows_left > 64 * (int64_t)fetched) ? rowp : lastp;
    rowp += step;
    ++fetched;
    asm volatile("" : "+v"(rowp));
    if (AVEC) return *reinterpret_cast<const uint4 *>(q);
    const word a = q[0], b = q[d1];
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
  };
  uint4 xs[TR_PREFETCH];
#pragma unroll
  for (int p = 0; p < TR_PREFETCH; ++p) xs[p] = fetch();
  TSTAMP(1);
  const int64_t dw0 = (tile % tiles_r) * TR_GROUPS, w0 = (tile / tiles_r) * TR_WORDS;
  // Results leave in two halves of 8 groups (64 bytes per row of D each, the second half of a line a few microseconds
  // after the first, from the same CU): 32 accumulator registers instead of 64 -- the kernel fits 128 registers and two
  // workgroups share a CU, one transposing while the other waits for memory.
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t acc[2][TR_GROUPS / 2][2];
#pragma unroll
    for (int gg = 0; gg < TR_GROUPS / 2; ++gg) {
      const int g = h * (TR_GROUPS / 2) + gg;
      *reinterpret_cast<uint4 *>(&stage[g & 1][lr][lw]) = xs[g % TR_PREFETCH];
      __syncthreads();  // one barrier per group: the buffer written now was last read two groups ago
      if (g == 0) TSTAMP(2);
      if (g == 8) TSTAMP(3);
      if (g + TR_PREFETCH < TR_GROUPS) xs[g % TR_PREFETCH] = fetch();
      const uint4 v = *reinterpret_cast<const uint4 *>(&stage[g & 1][lane][2 * wave]);
      uint32_t a = v.x, b = v.y, c = v.z, d = v.w;
      transpose_block(a, b, lane);
      transpose_block(c, d, lane);
      acc[0][gg][0] = a; acc[0][gg][1] = b;
      acc[1][gg][0] = c; acc[1][gg][1] = d;
      __builtin_amdgcn_sched_barrier(0);  // keep each group's work where it is written
    }
    if (h == 1) TSTAMP(4);
    // Lane j holds 8 consecutive words of row 64 (2 wave + k) + j of the result.  Stored from there an instruction would
    // make 64 requests of 16 bytes (measured: the store issue alone took 3 us per half tile); through the wave's own LDS
    // block (64 rows x 64 bytes, the 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3 so that neither the row-wise
    // writes nor the 4-lanes-per-row reads conflict) it makes 16 requests of 64 bytes.
    const int64_t dw = dw0 + h * (TR_GROUPS / 2);
    const int orow = lane >> 2, ochunk = lane & 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      word (*blk)[TR_GROUPS / 2] = ostage[wave];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<uint4 *>(&blk[lane][2 * (c ^ ((lane >> 2) & 3))]) =
            make_uint4(acc[k][2 * c][0], acc[k][2 * c][1], acc[k][2 * c + 1][0], acc[k][2 * c + 1][1]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr      = 16 * i + orow;
        const int64_t row = (w0 + 2 * wave + k) * 64 + rr;  // row of D = column of A
        uint4 v = *reinterpret_cast<const uint4 *>(&blk[rr][2 * (ochunk ^ ((rr >> 2) & 3))]);
        const int64_t dword = dw + 2 * ochunk;
        if (row >= ncols || dword >= wd) continue;
        if (dword == wd - 1) { v.x &= (uint32_t)dtail; v.y &= (uint32_t)(dtail >> 32); }
        if (dword + 1 == wd - 1) { v.z &= (uint32_t)dtail; v.w &= (uint32_t)(dtail >> 32); }
        word *dst = D + row * d_stride + dword;
        if (VEC && dword + 1 < wd) {
          *reinterpret_cast<uint4 *>(dst) = v;
        } else {
          dst[0] = (word)v.x | ((word)v.y << 32);
          if (dword + 1 < wd) dst[1] = (word)v.z | ((word)v.w << 32);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the block is reused for k = 1 / the next half
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  TSTAMP(5);
}

}  // namespace

// D (ncols x nrows bits, d_stride words per row) <- A^T, A nrows x ncols.  D must not overlap A.  Bits of A's last word
// beyond ncols are ignored; D's bits beyond column nrows come out zero.  Asynchronous on `stream`.
extern "C" int m4ri_amd_transpose_dev(word *D, int64_t d_stride, const word *A, int64_t a_stride, int64_t nrows, int64_t ncols,
                                      void *stream) {
  if (nrows < 0 || ncols < 0) return (int)hipErrorInvalidValue;
  if (nrows == 0 || ncols == 0) return 0;
  const int64_t tiles_r = (nrows + 64 * TR_GROUPS - 1) / (64 * TR_GROUPS), tiles_c = (((ncols + 63) >> 6) + TR_WORDS - 1) / TR_WORDS;
  if (tiles_r * tiles_c > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const bool vec  = (d_stride % 2 == 0) && (reinterpret_cast<uintptr_t>(D) % 16 == 0);
  const bool avec = (a_stride % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
  const int64_t ntiles = tiles_r * tiles_c;
  const dim3 grid((unsigned)ntiles), block(TR_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (vec && avec)  hipLaunchKernelGGL((transpose_kernel<true, true>), grid, block, 0, st, D, d_stride, A, a_stride, nrows, ncols, tiles_r, ntiles);
  else if (vec)     hipLaunchKernelGGL((transpose_kernel<true, false>), grid, block, 0, st, D, d_stride, A, a_stride, nrows, ncols, tiles_r, ntiles);
  else if (avec)    hipLaunchKernelGGL((transpose_kernel<false, true>), grid, block, 0, st, D, d_stride, A, a_stride, nrows, ncols, tiles_r, ntiles);
  else              hipLaunchKernelGGL((transpose_kernel<false, false>), grid, block, 0, st, D, d_stride, A, a_stride, nrows, ncols, tiles_r, ntiles);
  return (int)hipGetLastError();
}
