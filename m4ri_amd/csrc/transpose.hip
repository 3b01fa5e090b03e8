// transpose.hip -- D = A^T for bit-packed matrices (the device side of mzd_transpose, /root/reference m4ri/mzd.c:1104-1139;
// the reference works through 64 x 64 blocks with masked swaps on the CPU, mzd.c:700-1100).
//
// Pure data movement: every word of A is read once and every word of D written once, so the kernel is bounded by HBM
// (algorithmic bytes = 8 * (nrows * W(ncols) + ncols * W(nrows))).  The shape of the work is what needs care: a 64 x 64
// bit block sits in ONE word per row, so reading a block the way it is transposed (a lane per row) fetches 8 bytes out
// of every 128-byte line, and so does writing it.  Both sides are made whole lines:
//   * a workgroup owns a tile of 1024 rows x 16 words (1024 x 1024 bits).  It walks the tile in 16 groups of 64 rows;
//     a group's 64 x 16 words are loaded along the rows (128 contiguous bytes per row) and staged in LDS;
//   * wave w owns the word columns 2w, 2w+1: it takes its two 64 x 64 blocks of the group out of LDS (one ds_read_b128
//     per lane), transposes them in registers -- six exchange stages, ONE cross-lane dword per stage: the stage with
//     distance 32 swaps whole dwords, the others pack the half each partner needs of both dwords into one -- and keeps
//     the results;
//   * after the 16 groups lane j of the wave holds 16 consecutive words of row (64 (2w + k) + j) of the result: a whole
//     128-byte line per lane, stored as 16-byte pieces (the L2 assembles the lines; the leaf's C tiles leave the same way).
// The loads of group g + 1 are in flight while group g is transposed (registers -> the other LDS buffer).
#include <hip/hip_runtime.h>
#include "gf2_common.h"

namespace {

constexpr int TR_THREADS = 512;  // 8 waves
constexpr int TR_WORDS   = 16;   // tile width in words of A: 128 bytes per row
constexpr int TR_GROUPS  = 16;   // 64-row groups per tile: 128 bytes per row of D
constexpr int TR_PITCH   = 18;   // LDS row pitch in words: 16-byte aligned, 36 banks apart (b128 reads conflict-free)

// lane i holds row i of a 64 x 64 bit block in (lo, hi); on return lane j holds column j
__device__ __forceinline__ void transpose_block(uint32_t &lo, uint32_t &hi, int lane) {
  {
    const bool up    = (lane & 32) != 0;
    const uint32_t r = (uint32_t)__shfl_xor((int)(up ? lo : hi), 32);
    if (up) lo = r; else hi = r;
  }
#define TR_STAGE(D, M)                                                                              \
  {                                                                                                 \
    const bool up    = (lane & (D)) != 0;                                                           \
    const uint32_t s = up ? ((lo & (M)) | ((hi & (M)) << (D))) : (((lo >> (D)) & (M)) | (hi & ~(M))); \
    const uint32_t r = (uint32_t)__shfl_xor((int)s, (D));                                           \
    if (up) { lo = (lo & ~(M)) | (r & (M));          hi = (hi & ~(M)) | ((r & ~(M)) >> (D)); }      \
    else    { lo = (lo & (M)) | ((r & (M)) << (D));  hi = (hi & (M)) | (r & ~(M)); }                \
  }
  TR_STAGE(16, 0x0000ffffu)
  TR_STAGE(8, 0x00ff00ffu)
  TR_STAGE(4, 0x0f0f0f0fu)
  TR_STAGE(2, 0x33333333u)
  TR_STAGE(1, 0x55555555u)
#undef TR_STAGE
}

// VEC: D's rows are 16-byte aligned (even stride, aligned base) -> 16-byte stores
template <bool VEC>
__global__ __launch_bounds__(TR_THREADS) void transpose_kernel(word *__restrict__ D, int64_t d_stride, const word *__restrict__ A,
                                                               int64_t a_stride, int64_t nrows, int64_t ncols, int64_t tiles_r) {
  __shared__ __attribute__((aligned(16))) word stage[2][64][TR_PITCH];
  const int64_t tr = (int64_t)blockIdx.x % tiles_r, tc = (int64_t)blockIdx.x / tiles_r;
  const int64_t r0 = tr * 64 * TR_GROUPS, w0 = tc * TR_WORDS;  // first row / first word of the tile in A
  const int64_t wa = (ncols + 63) >> 6, wd = (nrows + 63) >> 6;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = t >> 3, lw = (t & 7) * 2;  // this thread's row within a group and its pair of words
  const word tail = (ncols & 63) ? ((~(word)0) >> (64 - (ncols & 63))) : ~(word)0;

  auto fetch = [&](int g, word &x0, word &x1) {
    const int64_t r = r0 + 64 * g + lr, w = w0 + lw;
    x0 = 0; x1 = 0;
    if (r < nrows) {
      const word *row = A + r * a_stride;
      if (w < wa)     { x0 = row[w];     if (w == wa - 1) x0 &= tail; }
      if (w + 1 < wa) { x1 = row[w + 1]; if (w + 1 == wa - 1) x1 &= tail; }
    }
  };

  uint32_t acc[2][TR_GROUPS][2];
  word x0, x1;
  fetch(0, x0, x1);
#pragma unroll
  for (int g = 0; g < TR_GROUPS; ++g) {
    word *srow = &stage[g & 1][lr][lw];
    *reinterpret_cast<uint4 *>(srow) = make_uint4((uint32_t)x0, (uint32_t)(x0 >> 32), (uint32_t)x1, (uint32_t)(x1 >> 32));
    __syncthreads();  // one barrier per group: the buffer written now was last read two groups ago
    if (g + 1 < TR_GROUPS) {
      if (r0 + 64 * (g + 1) < nrows) fetch(g + 1, x0, x1);
      else { x0 = 0; x1 = 0; }
    }
    const uint4 v = *reinterpret_cast<const uint4 *>(&stage[g & 1][lane][2 * wave]);
    uint32_t a = v.x, b = v.y, c = v.z, d = v.w;
    transpose_block(a, b, lane);
    transpose_block(c, d, lane);
    acc[0][g][0] = a; acc[0][g][1] = b;
    acc[1][g][0] = c; acc[1][g][1] = d;
  }

  const int64_t dw0 = r0 >> 6;  // first word of the tile in D's rows
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int64_t row = (w0 + 2 * wave + k) * 64 + lane;  // row of D = column of A
    if (row >= ncols) continue;
    word *dst = D + row * d_stride + dw0;
#pragma unroll
    for (int g = 0; g < TR_GROUPS; g += 2) {
      if (VEC && dw0 + g + 1 < wd) {
        *reinterpret_cast<uint4 *>(dst + g) = make_uint4(acc[k][g][0], acc[k][g][1], acc[k][g + 1][0], acc[k][g + 1][1]);
      } else {
        if (dw0 + g < wd)     dst[g]     = (word)acc[k][g][0] | ((word)acc[k][g][1] << 32);
        if (dw0 + g + 1 < wd) dst[g + 1] = (word)acc[k][g + 1][0] | ((word)acc[k][g + 1][1] << 32);
      }
    }
  }
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
  const bool vec = (d_stride % 2 == 0) && (reinterpret_cast<uintptr_t>(D) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((transpose_kernel<true>), dim3((unsigned)(tiles_r * tiles_c)), dim3(TR_THREADS), 0, (hipStream_t)stream, D,
                       d_stride, A, a_stride, nrows, ncols, tiles_r);
  else
    hipLaunchKernelGGL((transpose_kernel<false>), dim3((unsigned)(tiles_r * tiles_c)), dim3(TR_THREADS), 0, (hipStream_t)stream, D,
                       d_stride, A, a_stride, nrows, ncols, tiles_r);
  return (int)hipGetLastError();
}
