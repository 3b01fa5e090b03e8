// scheme_passes.hip -- the operand and result passes of 2, 3 or 4 fused Strassen levels whose LAST TWO levels are one application of a
// rank-R scheme for the 4 x 4 x 4 block product over GF(2) (scheme444.h; R < 49 where Strassen's algorithm applied twice needs 49).
//
// Replaces the reference's recursion   _mzd_mul_even   /root/reference m4ri/strassen.c:41-208   (7 products and 15 _mzd_add per level):
//   levels = 4   the scheme applied twice        R * R leaves per ancestor instead of 7^4 = 2401     ancestor = 16 x 16 grid of blocks
//   levels = 3   one Winograd level, then it     7 * R instead of 343                                 8 x 8
//   levels = 2   the scheme once                 R instead of 49                                      4 x 4
// An ancestor is a (4 G) x (4 G) grid of blocks, G = 4, 2, 1: block (4 i1 + i2, 4 j1 + j2).  Leaf (r1, r2), index r1 * R + r2, multiplies
//     sum_{(i1,j1) in OUTER_A[r1]} sum_{(i2,j2) in U[r2]} A-block (4 i1 + i2, 4 j1 + j2)      by the same sums of B-blocks (OUTER_B, V),
// and C-block (4 i1 + i2, 4 k1 + k2) is the sum of the products (r1, r2) with (i1, k1) in OUTER_C[r1] and (i2, k2) in W[r2]; the OUTER
// tables are the scheme's own (levels 4), Winograd's seven operand sums (levels 3) or the identity (levels 2).  Any valid scheme gives
// the same bits as any other (exact arithmetic): the parity tests do not know which one ran.
//
// Three kernels in the shape of the four-level Winograd passes of aux_kernels.hip -- ancestor through LDS once, nothing in between ever
// materialised, HBM-bound.  A workgroup owns POS word positions (B and C side) or POS rows of one word column (packed A side) and holds
// their grid in LDS; every UNIT of POS lanes works on the top-level products r1 = unit, unit + units, ...: the OUTER application walks its
// mask with scalar branches (a wave of two 32-lane units walks the union of both masks, each lane keeping what its own names), the INNER
// one is unrolled with the scheme's masks as compile-time constants (16 words in registers).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "gf2_common.h"
#include "scheme444.h"

namespace {

constexpr int R444 = SCHEME444_R;

// The outer applications, [levels - 2][side: 0 = A operands, 1 = B operands, 2 = results][r1] = mask over the G x G coarse blocks (bit
// G * i1 + j1), in constant memory, initialised at compile time: levels 4 the scheme's own tables; levels 3 Winograd's level in engine.hip's
// order -- A-side [A11, A12, S4, A22, S1, S2, S3], B-side [B11, B21, B22, T4, T1, T2, T3], product j to the quadrants winograd_scatter names
// (checked against the definition of the 2 x 2 product in tests/test_host_logic.py); levels 2 the identity.
struct Tables { uint16_t m[3][3][50]; };
constexpr Tables make_tables() {
  Tables t{};
  constexpr uint16_t WG[3][7] = {{0x1, 0x2, 0xF, 0x8, 0xC, 0xD, 0x5}, {0x1, 0x4, 0x8, 0xF, 0x3, 0xB, 0xA}, {0xF, 0x1, 0x2, 0x4, 0xA, 0xE, 0xC}};
  for (int side = 0; side < 3; ++side) {
    t.m[0][side][0] = 1;
    for (int r = 0; r < 7; ++r) t.m[1][side][r] = WG[side][r];
    for (int r = 0; r < R444; ++r) t.m[2][side][r] = side == 0 ? SCHEME444_U[r] : side == 1 ? SCHEME444_V[r] : SCHEME444_W[r];
  }
  return t;
}
__constant__ Tables c_tab = make_tables();

// what a kernel is told about its outer application: the number of top-level products and where their masks are
struct Outer { int32_t r1, lv, side; };
__device__ __forceinline__ uint32_t outer_mask(const Outer &o, int r1) { return (uint32_t)c_tab.m[o.lv][o.side][r1]; }

// the OR of `mine` over the wave's units (POS = 64: one unit per wave)
template <int POS>
__device__ __forceinline__ uint32_t wave_union(uint32_t mine) {
  if (POS == 64) return __builtin_amdgcn_readfirstlane(mine);
  return __builtin_amdgcn_readfirstlane(mine) | __builtin_amdgcn_readlane(mine, 32);
}

// x[f] = sum over the coarse blocks c = (i1, j1) in `mine` of grid block (4 i1 + i2, 4 j1 + j2), f = 4 i2 + j2
template <int G, int POS>
__device__ __forceinline__ void form_top(const word *grid, uint32_t mine, int pp, word (&x)[16]) {
  constexpr int GD = 4 * G;
  const uint32_t both = wave_union<POS>(mine);
#pragma unroll
  for (int f = 0; f < 16; ++f) x[f] = 0;
#pragma unroll 1
  for (int c = 0; c < G * G; ++c) {
    if (!((both >> c) & 1)) continue;   // scalar branch
    const word keep = (POS == 64 || ((mine >> c) & 1)) ? ~(word)0 : 0;
    const int i1 = c / G, j1 = c % G;
#pragma unroll
    for (int f = 0; f < 16; ++f) x[f] ^= grid[((4 * i1 + (f >> 2)) * GD + 4 * j1 + (f & 3)) * POS + pp] & keep;
  }
}

// the inner application of the operand side: child r2 = sum of the x[f] in U[r2] / V[r2]  (compile-time masks: straight-line XORs)
template <bool BSIDE>
__device__ __forceinline__ word inner_child(const word (&x)[16], int r2) {
  const uint16_t m = BSIDE ? SCHEME444_V[r2] : SCHEME444_U[r2];
  word v = 0;
#pragma unroll
  for (int f = 0; f < 16; ++f)
    if ((m >> f) & 1) v ^= x[f];
  return v;
}

// ---- down, row-major descendants (the B side) -------------------------------------------------------------------------------------
template <int G, int POS, int UNITS, bool BSIDE>
__global__ __launch_bounds__(POS * UNITS) void scheme_down_kernel(const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,  // ancestor: 4 G crows rows x 4 G cw words
                                   word *__restrict__ child, int64_t c_bs,                        // r1 * R descendants per ancestor, crows x cw words each
                                   int64_t crows, int64_t cw, Outer o) {                          // cw % POS == 0
  constexpr int GD = 4 * G, NB = GD * GD;
  __shared__ word grid[NB * POS];
  constexpr int units = UNITS;
  const int tid = threadIdx.x, pp = tid % POS, unit = tid / POS;
  const int64_t i = (int64_t)blockIdx.x * POS + pp, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  const word *p = anc + pi * p_bs + r * p_stride + w;
#pragma unroll 8
  for (int k = 0; k < (NB + UNITS - 1) / UNITS; ++k) {
    const int blk = k * UNITS + unit;
    if (NB % UNITS == 0 || blk < NB) grid[blk * POS + pp] = p[(int64_t)(blk / GD) * crows * p_stride + (int64_t)(blk % GD) * cw];
  }
  __syncthreads();
  word *out = child + pi * (int64_t)o.r1 * R444 * c_bs + r * cw + w;
  for (int r0 = 0; r0 < o.r1; r0 += units) {
    const int r1 = r0 + unit;
    const uint32_t mine = r1 < o.r1 ? outer_mask(o, r1) : 0u;
    word x[16];
    form_top<G, POS>(grid, mine, pp, x);
    if (r1 >= o.r1) continue;
    word *q = out + (int64_t)r1 * R444 * c_bs;
#pragma unroll
    for (int r2 = 0; r2 < R444; ++r2) q[(int64_t)r2 * c_bs] = inner_child<BSIDE>(x, r2);
  }
}

// ---- down, A side, written straight into the leaf's packed form (m4rm8q_leaf.hip: A4[chunk][row], index bytes rotated by (row >> 6) & 3) --
// A workgroup owns ONE word column and POS consecutive ROWS (lane = row): the two 32-bit chunks of a lane's word land in the packed array
// as POS consecutive dwords = whole 128-byte lines per chunk and child, straight from the registers.  The grid is loaded 8 bytes per row
// -- a load phase several times as long as the coalesced one of the B side -- so with the big grids POS is 32: 64 KiB of LDS, TWO
// workgroups per CU, one's loads under the other's stores.  The other 15 words of every loaded line belong to the workgroups of the
// neighbouring word columns, numbered 8 apart so that they run next to each other on one XCD and find the line in its L2 (the block map
// of winograd_down4_pack_lds_kernel).
template <int G, int POS, int UNITS>
__global__ __launch_bounds__(POS * UNITS) void scheme_down_pack_kernel(const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,
                                        uint32_t *__restrict__ a4, int64_t a4_bs,   // packed descendants, a4_bs dwords each
                                        int64_t crows, int64_t cw, Outer o) {       // crows % POS == 0, cw % 16 == 0
  constexpr int GD = 4 * G, NB = GD * GD;
  __shared__ word grid[NB * POS];
  constexpr int units = UNITS;
  const int tid = threadIdx.x, pp = tid % POS, unit = tid / POS;
  const int64_t b = blockIdx.x, grp = b >> 7, xcd = b & 7, slot = (b >> 3) & 15;
  const int64_t wgroups = cw >> 4;
  const int64_t rb = (grp / wgroups) * 8 + xcd, wc = (grp % wgroups) * 16 + slot;
  if (rb * POS >= crows) return;                     // (whole workgroups: the row blocks of the last group may not all exist)
  const int64_t r = rb * POS + pp, pi = blockIdx.y;
  const word *p = anc + pi * p_bs + r * p_stride + wc;
#pragma unroll 8
  for (int k = 0; k < (NB + UNITS - 1) / UNITS; ++k) {
    const int blk = k * UNITS + unit;
    if (NB % UNITS == 0 || blk < NB) grid[blk * POS + pp] = p[(int64_t)(blk / GD) * crows * p_stride + (int64_t)(blk % GD) * cw];
  }
  __syncthreads();
  const uint32_t rot = (uint32_t)((r >> 6) & 3);
  uint32_t *out = a4 + pi * (int64_t)o.r1 * R444 * a4_bs + (2 * wc) * crows + r;
  for (int r0 = 0; r0 < o.r1; r0 += units) {
    const int r1 = r0 + unit;
    const uint32_t mine = r1 < o.r1 ? outer_mask(o, r1) : 0u;
    word x[16];
    form_top<G, POS>(grid, mine, pp, x);
    if (r1 >= o.r1) continue;                        // (the last round's idle units)
    uint32_t *q = out + (int64_t)r1 * R444 * a4_bs;
#pragma unroll
    for (int r2 = 0; r2 < R444; ++r2) {
      const word v = inner_child<false>(x, r2);
      uint32_t w0 = (uint32_t)v, w1 = (uint32_t)(v >> 32);
      w0 = __builtin_amdgcn_alignbyte(w0, w0, rot);
      w1 = __builtin_amdgcn_alignbyte(w1, w1, rot);
      uint32_t *oo = q + (int64_t)r2 * a4_bs;
      oo[0]     = w0;
      oo[crows] = w1;
    }
  }
}

// ---- up: r1 * R products -> the grid of the ancestor ---------------------------------------------------------------------------------
// A unit folds the R sub-products of top-level product r1 into 16 fine words y[f] (inner application, compile-time masks), then adds
// y[f] to the coarse blocks of OUTER_C[r1] in the grid held in LDS (ds_xor: the units' products meet there); after ONE barrier the
// workgroup writes the grid out -- every word of C written once (read-modify-written once when accumulating).
template <int G, int POS, int UNITS, bool ACC>
__global__ __launch_bounds__(POS * UNITS) void scheme_up_kernel(const word *__restrict__ prod, int64_t p_bs,  // r1 * R products per ancestor, crows x cw words each
                                 word *anc, int64_t o_stride, int64_t o_bs, int64_t crows, int64_t cw, Outer o) {  // cw % POS == 0
  constexpr int GD = 4 * G, NB = GD * GD;
  __shared__ unsigned long long g[NB * POS];
  constexpr int units = UNITS;
  const int tid = threadIdx.x, pp = tid % POS, unit = tid / POS;
  for (int k = tid; k < NB * POS; k += POS * UNITS) g[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * POS + pp, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  const word *q0 = prod + pi * (int64_t)o.r1 * R444 * p_bs + r * cw + w;
  for (int r1 = unit; r1 < o.r1; r1 += units) {
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
    const uint32_t mine = outer_mask(o, r1);
#pragma unroll 1
    for (int c = 0; c < G * G; ++c) {
      if (!((mine >> c) & 1)) continue;   // per unit: the wave's other unit may take the branch
      const int i1 = c / G, k1 = c % G;
#pragma unroll
      for (int f = 0; f < 16; ++f) atomicXor(&g[((4 * i1 + (f >> 2)) * GD + 4 * k1 + (f & 3)) * POS + pp], (unsigned long long)y[f]);
    }
  }
  __syncthreads();
  word *out = anc + pi * o_bs + r * o_stride + w;
#pragma unroll 4
  for (int k = 0; k < (NB + UNITS - 1) / UNITS; ++k) {
    const int blk = k * UNITS + unit;
    if (NB % UNITS != 0 && blk >= NB) continue;
    word *oo     = out + (int64_t)(blk / GD) * crows * o_stride + (int64_t)(blk % GD) * cw;
    const word v = g[blk * POS + pp];
    *oo          = ACC ? (*oo ^ v) : v;
  }
}

Outer outer_for(int levels, int side) { return Outer{levels == 4 ? R444 : levels == 3 ? 7 : 1, levels - 2, side}; }

constexpr int64_t MAX_GRID_Y = 65535;  // ancestors per launch (they ride on gridDim.y)
constexpr int POS_B = 64;  // positions per workgroup of the B-side and result kernels' rule for the leaf shapes (cw % 64 == 0)

}  // namespace

extern "C" int gf2_scheme444_rank(void) { return R444; }

// leaves per ancestor of a `levels`-level scheme pass
extern "C" int64_t gf2_scheme444_leaves(int levels) { return levels == 4 ? (int64_t)R444 * R444 : levels == 3 ? 7 * (int64_t)R444 : levels == 2 ? R444 : 0; }

// can the scheme passes take these leaf shapes?  leaf of A: a_rows x a_cw words; of B: b_rows x b_cw words; of C: a_rows x b_cw
extern "C" int gf2_scheme444_ok(int levels, int64_t a_rows, int64_t a_cw, int64_t b_rows, int64_t b_cw) {
  // M4RI_AMD_SCHEME: unset = these passes wherever the table beats Strassen applied twice (R < 49: fewer leaves; with R = 49 the Winograd
  // passes of aux_kernels.hip are the same work and a little faster at two levels); 1 = wherever the shapes allow (how the tests reach
  // them whatever R is); 0 = never; 4 = only for four fused levels
  static const int sw = getenv("M4RI_AMD_SCHEME") ? atoi(getenv("M4RI_AMD_SCHEME")) : -1;
  if (sw == 0 || (sw == 4 && levels != 4) || (sw < 0 && R444 >= 49)) return 0;
  if (levels < 2 || levels > 4) return 0;
  if (a_rows <= 0 || b_rows <= 0 || a_rows % (levels == 2 ? 64 : 32) != 0 || a_cw % 16 != 0 || b_cw % POS_B != 0) return 0;
  if ((a_rows * a_cw) / 2 > 0x7fffffffLL || (b_rows * b_cw) / 32 > 0x7fffffffLL || (a_rows * b_cw) / 32 > 0x7fffffffLL) return 0;
  return 1;
}

#define SP_DISPATCH(LEVELS, CALL4, CALL2, CALL1) \
  do {                                           \
    if ((LEVELS) == 4) { CALL4; }                \
    else if ((LEVELS) == 3) { CALL2; }           \
    else { CALL1; }                              \
  } while (0)

// Descendant r1 * R + r2 of ancestor i is stored at index leaves * i + that; descendants are crows x cw words, contiguous; an ancestor is
// 2^levels * crows rows x 2^levels * cw words with row stride p_stride.
extern "C" hipError_t gf2_launch_scheme_down(hipStream_t s, int levels, int bside, const word *anc, int64_t p_stride, int64_t p_bs, word *child,
                                             int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t c_bs = crows * cw;
  if (nparents * c_bs == 0) return hipSuccess;
  if (levels < 2 || levels > 4 || cw % 64 != 0) return hipErrorInvalidValue;
  const Outer o = outer_for(levels, bside ? 1 : 0);
  const int64_t leaves = gf2_scheme444_leaves(levels);
  // ancestors ride on gridDim.y (at most 65535): more of them go in several launches
  for (int64_t p0 = 0; p0 < nparents; p0 += MAX_GRID_Y) {
    const int64_t np = nparents - p0 < MAX_GRID_Y ? nparents - p0 : MAX_GRID_Y;
    const word *an = anc + p0 * p_bs;
    word *ch       = child + p0 * leaves * c_bs;
    const dim3 g((unsigned)(c_bs / 64), (unsigned)np);
#define L_(G, T, BS) hipLaunchKernelGGL((scheme_down_kernel<G, 64, T / 64, BS>), g, dim3(T), 0, s, an, p_stride, p_bs, ch, c_bs, crows, cw, o)
    if (bside) SP_DISPATCH(levels, L_(4, 512, true), L_(2, 448, true), L_(1, 64, true));
    else SP_DISPATCH(levels, L_(4, 512, false), L_(2, 448, false), L_(1, 64, false));
#undef L_
  }
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_scheme_down_pack(hipStream_t s, int levels, const word *anc, int64_t p_stride, int64_t p_bs, word *a4,
                                                  int64_t nparents, int64_t crows, int64_t cw) {
  if (nparents * crows * cw == 0) return hipSuccess;
  const int pos = levels == 2 ? 64 : 32;
  if (levels < 2 || levels > 4 || crows % pos != 0 || cw % 16 != 0) return hipErrorInvalidValue;
  const Outer o = outer_for(levels, 0);
  const int64_t groups = ((crows / pos + 7) / 8) * (cw / 16);
  if (groups * 128 > 0x7fffffffLL) return hipErrorInvalidValue;
  const int64_t leaves = gf2_scheme444_leaves(levels);
  for (int64_t p0 = 0; p0 < nparents; p0 += MAX_GRID_Y) {
    const int64_t np = nparents - p0 < MAX_GRID_Y ? nparents - p0 : MAX_GRID_Y;
    const word *an = anc + p0 * p_bs;
    uint32_t *pk   = reinterpret_cast<uint32_t *>(a4) + p0 * leaves * crows * cw * 2;
    const dim3 g((unsigned)(groups * 128), (unsigned)np);
#define L_(G, P, T) hipLaunchKernelGGL((scheme_down_pack_kernel<G, P, T / P>), g, dim3(T), 0, s, an, p_stride, p_bs, pk, crows * cw * 2, crows, cw, o)
    SP_DISPATCH(levels, L_(4, 32, 256), L_(2, 32, 256), L_(1, 64, 64));
#undef L_
  }
  return hipGetLastError();
}

// anc (+)= the recombination of the products of every ancestor
extern "C" hipError_t gf2_launch_scheme_up(hipStream_t s, int levels, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs,
                                           int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t p_bs = crows * cw;
  if (nparents * p_bs == 0) return hipSuccess;
  if (levels < 2 || levels > 4 || cw % 64 != 0) return hipErrorInvalidValue;
  const Outer o = outer_for(levels, 2);
  const int pos = levels == 2 ? 64 : 32;
  const int64_t leaves = gf2_scheme444_leaves(levels);
  for (int64_t p0 = 0; p0 < nparents; p0 += MAX_GRID_Y) {
    const int64_t np = nparents - p0 < MAX_GRID_Y ? nparents - p0 : MAX_GRID_Y;
    const word *pr = prod + p0 * leaves * p_bs;
    word *an       = anc + p0 * o_bs;
    const dim3 g((unsigned)(p_bs / pos), (unsigned)np);
#define L_(G, P, T, AC) hipLaunchKernelGGL((scheme_up_kernel<G, P, T / P, AC>), g, dim3(T), 0, s, pr, p_bs, an, o_stride, o_bs, crows, cw, o)
    if (acc) SP_DISPATCH(levels, L_(4, 32, 256, true), L_(2, 32, 256, true), L_(1, 64, 64, true));
    else SP_DISPATCH(levels, L_(4, 32, 256, false), L_(2, 32, 256, false), L_(1, 64, 64, false));
#undef L_
  }
  return hipGetLastError();
}
