// scheme_passes.hip -- the operand and result passes of TWO PAIRS of Strassen levels done with a rank-R scheme for the 4 x 4 x 4 block
// product over GF(2) (scheme444.h; R = 47 where Strassen's algorithm applied twice needs 49: 47^2 = 2209 leaf products instead of
// 7^4 = 2401, 8 % less leaf work AND 8 % less pass traffic).
//
// Replaces, for four fused levels, the reference's recursion   _mzd_mul_even   /root/reference m4ri/strassen.c:41-208   (7 products and
// 15 _mzd_add per level) by: ancestor = a 16 x 16 grid of blocks, block (4 i1 + i2, 4 j1 + j2); leaf (r1, r2), index r1 * R + r2, multiplies
//     sum_{(i1,j1) in U[r1]} sum_{(i2,j2) in U[r2]} A-block (4 i1 + i2, 4 j1 + j2)      by the same sum over V of B-blocks,
// and C-block (4 i1 + i2, 4 k1 + k2) is the sum of the products (r1, r2) with (i1, k1) in W[r1] and (i2, k2) in W[r2].  Any valid scheme
// gives the same bits as any other (exact arithmetic): the parity tests do not know which one ran.
//
// Three kernels, the shape of the four-level Winograd passes in aux_kernels.hip (ancestor through LDS once, nothing in between ever
// materialised), HBM-bound: a workgroup of 8 waves owns 64 word positions (lane = position), holds their 16 x 16 grid in LDS (128 KiB),
// and wave w works on the top-level products r1 = w, w + 8, ...: the OUTER application is driven by wave-uniform masks (scalar branches
// skip what a product does not use), the INNER one is unrolled with the scheme's masks as compile-time constants (16 words in registers).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "gf2_common.h"
#include "scheme444.h"

namespace {

constexpr int SP_THREADS = 512;   // 8 waves
constexpr int SP_POS     = 64;    // word positions (B, C side) or rows (packed A side) per workgroup: one per lane
constexpr int R444       = SCHEME444_R;

__constant__ uint16_t c_U[R444], c_V[R444], c_W[R444];

// grid block (row block a, column block b) of position `lane`
__device__ __forceinline__ int gidx(int a, int b, int lane) { return (a * 16 + b) * SP_POS + lane; }

// x[f] = sum over the coarse blocks c = (i1, j1) in `mask` of grid block (4 i1 + i2, 4 j1 + j2), f = 4 i2 + j2   (mask wave-uniform)
__device__ __forceinline__ void form_top(const word *grid, uint32_t mask, int lane, word (&x)[16]) {
#pragma unroll
  for (int f = 0; f < 16; ++f) x[f] = 0;
#pragma unroll 1
  for (int c = 0; c < 16; ++c) {
    if (!((mask >> c) & 1)) continue;   // scalar branch
    const int i1 = c >> 2, j1 = c & 3;
#pragma unroll
    for (int f = 0; f < 16; ++f) x[f] ^= grid[gidx(4 * i1 + (f >> 2), 4 * j1 + (f & 3), lane)];
  }
}

// the inner application of the operand side: child r2 = sum of the x[f] in MASKS[r2]  (compile-time masks: straight-line XORs)
template <bool BSIDE>
__device__ __forceinline__ word inner_child(const word (&x)[16], int r2) {
  const uint16_t m = BSIDE ? SCHEME444_V[r2] : SCHEME444_U[r2];
  word v = 0;
#pragma unroll
  for (int f = 0; f < 16; ++f)
    if ((m >> f) & 1) v ^= x[f];
  return v;
}

// ---- down, B side: descendants row-major ----------------------------------------------------------------------------------------
template <bool BSIDE>
__global__ __launch_bounds__(SP_THREADS) void scheme_down_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,  // ancestor: 16 * crows rows x 16 * cw words
    word *__restrict__ child, int64_t c_bs,                        // R * R descendants per ancestor, crows x cw words each, contiguous
    int64_t crows, int64_t cw) {                                   // cw % 64 == 0: the 64 positions of a workgroup lie in one row
  __shared__ word grid[256 * SP_POS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t i = (int64_t)blockIdx.x * SP_POS + lane, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  const word *p = anc + pi * p_bs + r * p_stride + w;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + wave;
    grid[blk * SP_POS + lane] = p[(int64_t)(blk >> 4) * crows * p_stride + (int64_t)(blk & 15) * cw];
  }
  __syncthreads();
  word *out = child + pi * (int64_t)(R444 * R444) * c_bs + r * cw + w;
  for (int r1 = wave; r1 < R444; r1 += 8) {
    const uint32_t mask = __builtin_amdgcn_readfirstlane((uint32_t)(BSIDE ? c_V[r1] : c_U[r1]));
    word x[16];
    form_top(grid, mask, lane, x);
    word *o = out + (int64_t)r1 * R444 * c_bs;
#pragma unroll
    for (int r2 = 0; r2 < R444; ++r2) o[(int64_t)r2 * c_bs] = inner_child<BSIDE>(x, r2);
  }
}

// ---- down, A side, written straight into the leaf's packed form (m4rm8q_leaf.hip: A4[chunk][row], index bytes rotated by (row >> 6) & 3) --
// A workgroup of 4 waves owns ONE word column and 32 consecutive ROWS (lane & 31 = row): the two 32-bit chunks of a lane's word land in the
// packed array as 32 consecutive dwords = one full 128-byte line per chunk and child, straight from the registers.  The grid is loaded
// 8 bytes per row -- a load phase four times as long as the coalesced one of the B side -- so this kernel keeps the grid at 64 KiB (32
// positions) and TWO workgroups per CU, one's loads under the other's stores; the price is that the two half-waves of a wave work on
// different top-level products: the outer application walks the union of their two masks and each lane keeps what its own mask names.
// The other 15 words of every loaded line belong to the workgroups of the neighbouring word columns, numbered 8 apart so that they run
// next to each other on one XCD and find the line in its L2 (the block map of winograd_down4_pack_lds_kernel).
constexpr int PK_THREADS = 256, PK_POS = 32;

__global__ __launch_bounds__(PK_THREADS) void scheme_down_pack_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,
    uint32_t *__restrict__ a4, int64_t a4_bs,       // packed descendants, a4_bs dwords each
    int64_t crows, int64_t cw) {                    // crows % 32 == 0, cw % 16 == 0
  __shared__ word grid[256 * PK_POS];
  const int tid = threadIdx.x, pp = tid & 31, hw = tid >> 5;   // hw: half-wave 0 .. 7
  const int64_t b = blockIdx.x, grp = b >> 7, xcd = b & 7, slot = (b >> 3) & 15;
  const int64_t wgroups = cw >> 4;
  const int64_t rb = (grp / wgroups) * 8 + xcd, wc = (grp % wgroups) * 16 + slot;
  if (rb * PK_POS >= crows) return;                  // (whole workgroups: the row blocks of the last group may not all exist)
  const int64_t r = rb * PK_POS + pp, pi = blockIdx.y;
  const word *p = anc + pi * p_bs + r * p_stride + wc;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + hw;
    grid[blk * PK_POS + pp] = p[(int64_t)(blk >> 4) * crows * p_stride + (int64_t)(blk & 15) * cw];
  }
  __syncthreads();
  const uint32_t rot = (uint32_t)((r >> 6) & 3);
  uint32_t *out = a4 + pi * (int64_t)(R444 * R444) * a4_bs + (2 * wc) * crows + r;
  const int hw_lo = (tid >> 6) * 2;                  // the wave's first half-wave
  for (int r0 = 0; r0 < R444; r0 += 8) {
    const int r1 = r0 + hw;
    // the masks of the wave's two half-waves (wave-uniform scalars), this lane's own among them
    const uint32_t ma = r0 + hw_lo < R444 ? (uint32_t)c_U[r0 + hw_lo] : 0u, mb = r0 + hw_lo + 1 < R444 ? (uint32_t)c_U[r0 + hw_lo + 1] : 0u;
    const uint32_t mine = (hw & 1) ? mb : ma, both = ma | mb;
    word x[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) x[f] = 0;
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
      if (!((both >> c) & 1)) continue;             // scalar branch
      const word keep = ((mine >> c) & 1) ? ~(word)0 : 0;
      const int i1 = c >> 2, j1 = c & 3;
#pragma unroll
      for (int f = 0; f < 16; ++f) x[f] ^= grid[((4 * i1 + (f >> 2)) * 16 + 4 * j1 + (f & 3)) * PK_POS + pp] & keep;
    }
    if (r1 >= R444) continue;                        // (the last round's idle half-wave)
    uint32_t *o = out + (int64_t)r1 * R444 * a4_bs;
#pragma unroll
    for (int r2 = 0; r2 < R444; ++r2) {
      const word v = inner_child<false>(x, r2);
      uint32_t w0 = (uint32_t)v, w1 = (uint32_t)(v >> 32);
      w0 = __builtin_amdgcn_alignbyte(w0, w0, rot);
      w1 = __builtin_amdgcn_alignbyte(w1, w1, rot);
      uint32_t *oo = o + (int64_t)r2 * a4_bs;
      oo[0]     = w0;
      oo[crows] = w1;
    }
  }
}

// ---- up: R * R products -> the 16 x 16 grid of the ancestor ------------------------------------------------------------------------
// Wave w folds the R sub-products of top-level product r1 into 16 fine words y[f] (inner application, compile-time masks), then adds
// y[f] to the coarse blocks of W[r1] in the grid held in LDS (ds_xor: the waves' products meet there); after ONE barrier the workgroup
// writes the grid out -- every word of C written once (read-modify-written once when accumulating).
template <bool ACC>
__global__ __launch_bounds__(SP_THREADS) void scheme_up_kernel(
    const word *__restrict__ prod, int64_t p_bs,  // R * R products per ancestor, crows x cw words each, contiguous
    word *anc, int64_t o_stride, int64_t o_bs,
    int64_t crows, int64_t cw) {                  // cw % 64 == 0
  __shared__ word grid[256 * SP_POS];
  unsigned long long *g = reinterpret_cast<unsigned long long *>(grid);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < 256 * SP_POS; k += SP_THREADS) g[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * SP_POS + lane, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  const word *q0 = prod + pi * (int64_t)(R444 * R444) * p_bs + r * cw + w;
  for (int r1 = wave; r1 < R444; r1 += 8) {
    const word *q = q0 + (int64_t)r1 * R444 * p_bs;
    word y[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) y[f] = 0;
    // the R products in groups of 8 loads, then folded: keeps the loads of a group in flight together without holding all R at once
#pragma unroll
    for (int g0 = 0; g0 < R444; g0 += 8) {
      word v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g0 + k < R444) v[k] = q[(int64_t)(g0 + k) * p_bs];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g0 + k < R444) {
          const uint16_t m = SCHEME444_W[g0 + k];
#pragma unroll
          for (int f = 0; f < 16; ++f)
            if ((m >> f) & 1) y[f] ^= v[k];
        }
    }
    const uint32_t mask = __builtin_amdgcn_readfirstlane((uint32_t)c_W[r1]);
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
      if (!((mask >> c) & 1)) continue;
      const int i1 = c >> 2, k1 = c & 3;
#pragma unroll
      for (int f = 0; f < 16; ++f) atomicXor(&g[gidx(4 * i1 + (f >> 2), 4 * k1 + (f & 3), lane)], (unsigned long long)y[f]);
    }
  }
  __syncthreads();
  word *o = anc + pi * o_bs + r * o_stride + w;
#pragma unroll 4
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + wave;
    word *oo      = o + (int64_t)(blk >> 4) * crows * o_stride + (int64_t)(blk & 15) * cw;
    const word v  = grid[blk * SP_POS + lane];
    *oo           = ACC ? (*oo ^ v) : v;
  }
}

// The same pass with 32 positions per workgroup (64 KiB of LDS, TWO workgroups per CU: one's clear and write-out under the other's loads);
// the two half-waves of a wave fold different top-level products, so the scatter into the coarse blocks is predicated per lane.
template <bool ACC>
__global__ __launch_bounds__(PK_THREADS) void scheme_up32_kernel(
    const word *__restrict__ prod, int64_t p_bs, word *anc, int64_t o_stride, int64_t o_bs, int64_t crows, int64_t cw) {  // cw % 32 == 0
  __shared__ unsigned long long g[256 * PK_POS];
  const int tid = threadIdx.x, pp = tid & 31, hw = tid >> 5;
  for (int k = tid; k < 256 * PK_POS; k += PK_THREADS) g[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * PK_POS + pp, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  const word *q0 = prod + pi * (int64_t)(R444 * R444) * p_bs + r * cw + w;
  for (int r0 = 0; r0 < R444; r0 += 8) {
    const int r1 = r0 + hw;
    if (r1 >= R444) continue;   // (the last round's idle half-wave; no barrier inside the loop)
    const word *q = q0 + (int64_t)r1 * R444 * p_bs;
    word y[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) y[f] = 0;
#pragma unroll
    for (int g0 = 0; g0 < R444; g0 += 8) {
      word v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g0 + k < R444) v[k] = q[(int64_t)(g0 + k) * p_bs];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g0 + k < R444) {
          const uint16_t m = SCHEME444_W[g0 + k];
#pragma unroll
          for (int f = 0; f < 16; ++f)
            if ((m >> f) & 1) y[f] ^= v[k];
        }
    }
    const uint32_t mine = (uint32_t)c_W[r1];
#pragma unroll 1
    for (int c = 0; c < 16; ++c) {
      if (!((mine >> c) & 1)) continue;   // per half-wave: the wave's other half may take the branch
      const int i1 = c >> 2, k1 = c & 3;
#pragma unroll
      for (int f = 0; f < 16; ++f) atomicXor(&g[((4 * i1 + (f >> 2)) * 16 + 4 * k1 + (f & 3)) * PK_POS + pp], (unsigned long long)y[f]);
    }
  }
  __syncthreads();
  word *o = anc + pi * o_bs + r * o_stride + w;
#pragma unroll 4
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + hw;
    word *oo      = o + (int64_t)(blk >> 4) * crows * o_stride + (int64_t)(blk & 15) * cw;
    const word v  = g[blk * PK_POS + pp];
    *oo           = ACC ? (*oo ^ v) : v;
  }
}

bool g_tables_up[16] = {};

hipError_t upload_tables() {  // the outer application's masks, once per device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (g_tables_up[dev]) return hipSuccess;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_U), SCHEME444_U, sizeof c_U)) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_V), SCHEME444_V, sizeof c_V)) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_W), SCHEME444_W, sizeof c_W)) != hipSuccess) return e;
  g_tables_up[dev] = true;
  return hipSuccess;
}

}  // namespace

// products per ancestor of the four-level scheme passes (R^2), and whether the passes can take these leaf shapes
extern "C" int gf2_scheme444_rank(void) { return R444; }

extern "C" int gf2_scheme444_ok(int64_t a_rows, int64_t a_cw, int64_t b_rows, int64_t b_cw) {  // leaf of A: a_rows x a_cw words; of B: b_rows x b_cw; C: a_rows x b_cw
  static const bool off = getenv("M4RI_AMD_SCHEME") && atoi(getenv("M4RI_AMD_SCHEME")) == 0;   // developer switch: four Winograd levels instead
  if (off) return 0;
  if (a_rows <= 0 || b_rows <= 0 || a_rows % PK_POS != 0 || a_cw % 16 != 0 || b_cw % SP_POS != 0) return 0;
  if ((a_rows * a_cw) / 8 > 0x7fffffffLL || (b_rows * b_cw) / SP_POS > 0x7fffffffLL || (a_rows * b_cw) / SP_POS > 0x7fffffffLL) return 0;
  return 1;
}

// Descendant r1 * R + r2 of ancestor i is stored at index R * R * i + that; descendants are crows x cw words, contiguous; an ancestor is
// 16 * crows rows x 16 * cw words with row stride p_stride.
extern "C" hipError_t gf2_launch_scheme_down(hipStream_t s, int bside, const word *anc, int64_t p_stride, int64_t p_bs, word *child,
                                             int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t c_bs = crows * cw;
  if (nparents * c_bs == 0) return hipSuccess;
  if (cw % SP_POS != 0 || nparents > 65535) return hipErrorInvalidValue;
  if (hipError_t e = upload_tables()) return e;
  const dim3 g((unsigned)(c_bs / SP_POS), (unsigned)nparents);
  if (bside) hipLaunchKernelGGL((scheme_down_kernel<true>), g, dim3(SP_THREADS), 0, s, anc, p_stride, p_bs, child, c_bs, crows, cw);
  else hipLaunchKernelGGL((scheme_down_kernel<false>), g, dim3(SP_THREADS), 0, s, anc, p_stride, p_bs, child, c_bs, crows, cw);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_scheme_down_pack(hipStream_t s, const word *anc, int64_t p_stride, int64_t p_bs, word *a4, int64_t nparents,
                                                  int64_t crows, int64_t cw) {
  if (nparents * crows * cw == 0) return hipSuccess;
  if (crows % PK_POS != 0 || cw % 16 != 0 || nparents > 65535) return hipErrorInvalidValue;
  if (hipError_t e = upload_tables()) return e;
  const int64_t groups = ((crows / PK_POS + 7) / 8) * (cw / 16);
  if (groups * 128 > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(scheme_down_pack_kernel, dim3((unsigned)(groups * 128), (unsigned)nparents), dim3(PK_THREADS), 0, s, anc, p_stride, p_bs,
                     reinterpret_cast<uint32_t *>(a4), crows * cw * 2, crows, cw);
  return hipGetLastError();
}

// anc (+)= the recombination of the R * R products per ancestor
extern "C" hipError_t gf2_launch_scheme_up(hipStream_t s, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs, int64_t nparents,
                                           int64_t crows, int64_t cw) {
  const int64_t p_bs = crows * cw;
  if (nparents * p_bs == 0) return hipSuccess;
  if (cw % SP_POS != 0 || nparents > 65535) return hipErrorInvalidValue;
  if (hipError_t e = upload_tables()) return e;
  static const bool wide = getenv("M4RI_AMD_SCHEME_UP") && atoi(getenv("M4RI_AMD_SCHEME_UP")) == 64;   // developer: the 64-position form
  if (!wide) {
    const dim3 g32((unsigned)(p_bs / PK_POS), (unsigned)nparents);
    if (acc) hipLaunchKernelGGL((scheme_up32_kernel<true>), g32, dim3(PK_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows, cw);
    else hipLaunchKernelGGL((scheme_up32_kernel<false>), g32, dim3(PK_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows, cw);
    return hipGetLastError();
  }
  const dim3 g((unsigned)(p_bs / SP_POS), (unsigned)nparents);
  if (acc) hipLaunchKernelGGL((scheme_up_kernel<true>), g, dim3(SP_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows, cw);
  else hipLaunchKernelGGL((scheme_up_kernel<false>), g, dim3(SP_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows, cw);
  return hipGetLastError();
}
