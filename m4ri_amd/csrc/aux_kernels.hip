// aux_kernels.hip -- the HBM-bound helpers around the M4RM leaf: the fused Strassen-Winograd passes
// (one, two or three levels per pass; the A side optionally written straight into the leaf's packed
// form), strided XOR / copy / masked copy, the fold of split leaf launches, tail masking and the
// deterministic fill.
//
// The passes replace the reference's 15 separate quadrant additions per recursion node
// (_mzd_add, /root/reference m4ri/mzd.c:1471-1583, called from m4ri/strassen.c:111-150) by one
// "down" pass per operand and one "up" pass: every operand word is read once and every result word
// written once per PASS (33 quadrant transfers per node instead of 45 with single-level passes;
// multi-level passes skip the intermediate levels altogether).
//
// All of these are streaming kernels: one 64-bit word (or a 16-byte pair) per lane, rows
// contiguous -- bounded by HBM bandwidth; only the packed-output passes tile (an LDS transpose).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "gf2_common.h"

namespace {

constexpr int AUX_THREADS = 256;
constexpr int PASS_NT_DEFAULT = 2;  // up3 reads its 343 products once: nontemporal loads measure 0.70 -> 0.60 ms at 65536^3; nontemporal stores in the down passes measure slower

// ---- Winograd operand combinations ----------------------------------------------------------
// With the parent split into quadrants X11 X12 / X21 X22 the 7 children of the A side are
//   [A11, A12, S4, A22, S1, S2, S3],  S1 = A21+A22, S2 = S1+A11, S3 = A11+A21, S4 = A12+S2
// and of the B side
//   [B11, B21, B22, T4, T1, T2, T3],  T1 = B12+B11, T2 = B22+T1, T3 = B22+B12, T4 = T2+B21
// so that product j = Achild_j * Bchild_j is P1..P7 of the Strassen-Winograd scheme.
template <typename V, bool BSIDE>
__global__ __launch_bounds__(AUX_THREADS) void winograd_down_kernel(
    const V *__restrict__ parent, int64_t p_stride, int64_t p_bs,  // parent array (units of V)
    V *__restrict__ child, int64_t c_bs,                           // child array, stride == cw
    int64_t nparents, int64_t crows, int64_t cw, int64_t qoff_rows, int64_t qoff_cols) {
  const int64_t per    = crows * cw;
  const int64_t total  = nparents * per;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t pi = i / per;
    const int64_t rm = i - pi * per;
    const int64_t r  = rm / cw;
    const int64_t w  = rm - r * cw;
    const V *p       = parent + pi * p_bs + r * p_stride + w;
    const V x11 = p[0], x12 = p[qoff_cols], x21 = p[qoff_rows * p_stride],
            x22 = p[qoff_rows * p_stride + qoff_cols];
    V *c = child + (pi * 7) * c_bs + r * cw + w;
    if (!BSIDE) {
      const V s1 = x21 ^ x22, s2 = s1 ^ x11, s3 = x11 ^ x21, s4 = x12 ^ s2;
      c[0 * c_bs] = x11; c[1 * c_bs] = x12; c[2 * c_bs] = s4; c[3 * c_bs] = x22;
      c[4 * c_bs] = s1;  c[5 * c_bs] = s2;  c[6 * c_bs] = s3;
    } else {
      const V t1 = x12 ^ x11, t2 = x22 ^ t1, t3 = x22 ^ x12, t4 = t2 ^ x21;
      c[0 * c_bs] = x11; c[1 * c_bs] = x21; c[2 * c_bs] = x22; c[3 * c_bs] = t4;
      c[4 * c_bs] = t1;  c[5 * c_bs] = t2;  c[6 * c_bs] = t3;
    }
  }
}

// ---- Winograd recombination ------------------------------------------------------------------
//   U2 = P1+P6, U3 = U2+P7, U4 = U2+P5
//   C11 = P1+P2, C12 = U4+P3, C21 = U3+P4, C22 = U3+P5        (ACC: parent ^= instead of =)
template <typename V, bool ACC>
__global__ __launch_bounds__(AUX_THREADS) void winograd_up_kernel(
    const V *__restrict__ prod, int64_t p_bs,                      // products, stride == cw
    V *__restrict__ parent, int64_t o_stride, int64_t o_bs,        // parent array
    int64_t nparents, int64_t crows, int64_t cw, int64_t qoff_rows, int64_t qoff_cols) {
  const int64_t per    = crows * cw;
  const int64_t total  = nparents * per;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t pi = i / per;
    const int64_t rm = i - pi * per;
    const int64_t r  = rm / cw;
    const int64_t w  = rm - r * cw;
    const V *q       = prod + (pi * 7) * p_bs + r * cw + w;
    const V p1 = q[0], p2 = q[p_bs], p3 = q[2 * p_bs], p4 = q[3 * p_bs], p5 = q[4 * p_bs],
            p6 = q[5 * p_bs], p7 = q[6 * p_bs];
    const V u2 = p1 ^ p6, u3 = u2 ^ p7, u4 = u2 ^ p5;
    V c11 = p1 ^ p2, c12 = u4 ^ p3, c21 = u3 ^ p4, c22 = u3 ^ p5;
    V *o = parent + pi * o_bs + r * o_stride + w;
    if (ACC) {
      c11 ^= o[0]; c12 ^= o[qoff_cols]; c21 ^= o[qoff_rows * o_stride];
      c22 ^= o[qoff_rows * o_stride + qoff_cols];
    }
    o[0]                                  = c11;
    o[qoff_cols]                          = c12;
    o[qoff_rows * o_stride]               = c21;
    o[qoff_rows * o_stride + qoff_cols]   = c22;
  }
}

// ---- two levels at once ------------------------------------------------------------------------
// The two deepest levels of the schedule fused: a grandparent is read once as a 4 x 4 grid of blocks
// and its 49 grandchildren are written directly (index 7*j1 + j2, exactly what two single-level
// passes produce), and on the way up 49 products become the 16 blocks of the grandparent.  Per
// grandparent that is 1 + 49/16 transfers instead of (1 + 7/4) + (7/4 + 49/16): 38 % less HBM
// traffic for the two largest levels of every pass.
template <typename V, bool BSIDE>
__device__ __forceinline__ void winograd_combos(const V x11, const V x12, const V x21, const V x22, V out[7]) {
  if (!BSIDE) {
    const V s1 = x21 ^ x22, s2 = s1 ^ x11, s3 = x11 ^ x21, s4 = x12 ^ s2;
    out[0] = x11; out[1] = x12; out[2] = s4; out[3] = x22; out[4] = s1; out[5] = s2; out[6] = s3;
  } else {
    const V t1 = x12 ^ x11, t2 = x22 ^ t1, t3 = x22 ^ x12, t4 = t2 ^ x21;
    out[0] = x11; out[1] = x21; out[2] = x22; out[3] = t4; out[4] = t1; out[5] = t2; out[6] = t3;
  }
}

template <typename V>
__device__ __forceinline__ void winograd_recombine(const V p[7], V &c11, V &c12, V &c21, V &c22) {
  const V u2 = p[0] ^ p[5], u3 = u2 ^ p[6], u4 = u2 ^ p[4];
  c11 = p[0] ^ p[1]; c12 = u4 ^ p[2]; c21 = u3 ^ p[3]; c22 = u3 ^ p[4];
}

template <typename V, bool BSIDE>
__global__ __launch_bounds__(AUX_THREADS) void winograd_down2_kernel(
    const V *__restrict__ gparent, int64_t p_stride, int64_t p_bs,  // grandparent array (units of V)
    V *__restrict__ gchild, int64_t c_bs,                           // grandchildren, stride == cw
    int64_t nparents, int64_t crows, int64_t cw) {                  // grandchild shape: crows x cw
  const int64_t per    = crows * cw;
  const int64_t total  = nparents * per;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t pi = i / per;
    const int64_t rm = i - pi * per;
    const int64_t r  = rm / cw;
    const int64_t w  = rm - r * cw;
    const V *p       = gparent + pi * p_bs + r * p_stride + w;
    V x[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) x[a][b] = p[(int64_t)a * crows * p_stride + (int64_t)b * cw];
    // level 1: child j1 of the grandparent, as its own 2 x 2 blocks y[.][a][b]
    V y[4][7];  // y[2a+b][j1]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) winograd_combos<V, BSIDE>(x[a][b], x[a][b + 2], x[a + 2][b], x[a + 2][b + 2], y[2 * a + b]);
    V *c = gchild + (pi * 49) * c_bs + r * cw + w;
#pragma unroll
    for (int j1 = 0; j1 < 7; ++j1) {
      V g[7];
      winograd_combos<V, BSIDE>(y[0][j1], y[1][j1], y[2][j1], y[3][j1], g);
#pragma unroll
      for (int j2 = 0; j2 < 7; ++j2) c[(int64_t)(7 * j1 + j2) * c_bs] = g[j2];
    }
  }
}

template <typename V, bool ACC>
__global__ __launch_bounds__(AUX_THREADS) void winograd_up2_kernel(
    const V *__restrict__ prod, int64_t p_bs,                       // 49 products per grandparent, stride == cw
    V *__restrict__ gparent, int64_t o_stride, int64_t o_bs,        // grandparent array
    int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t per    = crows * cw;
  const int64_t total  = nparents * per;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t pi = i / per;
    const int64_t rm = i - pi * per;
    const int64_t r  = rm / cw;
    const int64_t w  = rm - r * cw;
    const V *q       = prod + (pi * 49) * p_bs + r * cw + w;
    // level 2 -> 1: the four blocks of each of the 7 products of level 1
    V y[4][7];  // y[2a+b][j1]
#pragma unroll
    for (int j1 = 0; j1 < 7; ++j1) {
      V g[7];
#pragma unroll
      for (int j2 = 0; j2 < 7; ++j2) g[j2] = q[(int64_t)(7 * j1 + j2) * p_bs];
      winograd_recombine<V>(g, y[0][j1], y[1][j1], y[2][j1], y[3][j1]);
    }
    // level 1 -> 0: block (a, b) of each quadrant of the grandparent
    V *o = gparent + pi * o_bs + r * o_stride + w;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        V c11, c12, c21, c22;
        winograd_recombine<V>(y[2 * a + b], c11, c12, c21, c22);
        V *o11 = o + (int64_t)a * crows * o_stride + (int64_t)b * cw;
        V *o12 = o11 + 2 * cw, *o21 = o11 + (int64_t)2 * crows * o_stride, *o22 = o21 + 2 * cw;
        if (ACC) { c11 ^= *o11; c12 ^= *o12; c21 ^= *o21; c22 ^= *o22; }
        *o11 = c11; *o12 = c12; *o21 = c21; *o22 = c22;
      }
  }
}

// C = A ^ B (whole words) on strided views; op 1: C = A (copy); op 2: C = 0.  The last word of a row
// is merged under `mask` (bits outside it keep C's value, mzd.c:1489); mask == ~0 writes whole words.
// C may alias A and/or B (no __restrict__).
__global__ __launch_bounds__(AUX_THREADS) void rowwise_kernel(word *C, int64_t cs, const word *A, int64_t as,
                                                              const word *B, int64_t bs,
                                                              int64_t rows, int64_t w, int op, word mask) {
  const int64_t total  = rows * w;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / w, k = i - r * w;
    word v = 0;
    if (op == 0) v = A[r * as + k] ^ B[r * bs + k];
    else if (op == 1) v = A[r * as + k];
    word *c = C + r * cs + k;
    if (k == w - 1 && mask != ~(word)0) v = (*c & ~mask) | (v & mask);
    *c = v;
  }
}

// C = A on strided views, the last word of every row merged under `mask` (bits outside it keep C's
// value): the device twin of mzd_copy's masked last word (mzd.c:1363-1382), used to put a result into
// a window of a device-resident parent without touching the parent's other columns
__global__ __launch_bounds__(AUX_THREADS) void copy_masked_kernel(word *__restrict__ C, int64_t cs,
                                                                  const word *__restrict__ A, int64_t as,
                                                                  int64_t rows, int64_t w, word mask) {
  const int64_t total  = rows * w;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / w, k = i - r * w;
    const word v = A[r * as + k];
    word *c      = C + r * cs + k;
    *c           = (k == w - 1) ? ((*c & ~mask) | (v & mask)) : v;
  }
}

// fold the slabs of an inner-dimension split (leaf mode 2) into C: tile t of the launch has ks dense
// slabs of tile_rows x tw words at Cpart + ((t - tile_base) * ks + k) * tile_rows * tw; C (^)= their
// XOR, clipped to the matrix.  RP_CHUNKS workgroups per tile (a tile alone is too few workgroups to
// pull bandwidth: a split launch has at most a few hundred tiles).
constexpr int RP_CHUNKS = 32;
template <bool ACC>
__global__ __launch_bounds__(AUX_THREADS) void reduce_partials_kernel(word *__restrict__ C, int64_t cs, int64_t cbs, int64_t m,
                                                                      int64_t wn, int64_t tile_rows, int64_t tw, int64_t tiles_m,
                                                                      int64_t tiles_n, int64_t tile_base, int ks,
                                                                      const word *__restrict__ Cpart) {
  const int64_t lt = blockIdx.x / RP_CHUNKS, chunk = blockIdx.x % RP_CHUNKS;
  int64_t t = tile_base + lt;
  const int64_t tm = t % tiles_m; t /= tiles_m;
  const int64_t tn = t % tiles_n; t /= tiles_n;
  word *base        = C + t * cbs + tm * tile_rows * cs + tn * tw;
  const int64_t rows = (m - tm * tile_rows) < tile_rows ? (m - tm * tile_rows) : tile_rows;
  const int64_t w    = (wn - tn * tw) < tw ? (wn - tn * tw) : tw;
  const int64_t slab = tile_rows * tw;
  const word *src    = Cpart + lt * ks * slab;
  const int64_t per  = (tile_rows / RP_CHUNKS) * tw;  // words of the tile this workgroup folds
  const int64_t end  = (chunk + 1) * per < rows * tw ? (chunk + 1) * per : rows * tw;
  for (int64_t i = chunk * per + threadIdx.x; i < end; i += AUX_THREADS) {
    const int64_t r = i / tw, k = i - r * tw;
    if (k >= w) continue;
    word x = 0;
    for (int j = 0; j < ks; ++j) x ^= src[(int64_t)j * slab + i];
    word *c = base + r * cs + k;
    *c      = ACC ? (*c ^ x) : x;
  }
}

// zero the C tiles [tile_base, tile_base + ntiles) of a batched leaf launch (linear tile order:
// tile_m fastest, then tile_n, then batch member): what an inner-dimension split of only SOME tiles
// needs before its atomic XORs.  One workgroup per tile, rows x tw words each.
__global__ __launch_bounds__(AUX_THREADS) void zero_tiles_kernel(word *__restrict__ C, int64_t cs, int64_t cbs, int64_t m,
                                                                 int64_t wn, int64_t tile_rows, int64_t tw, int64_t tiles_m,
                                                                 int64_t tiles_n, int64_t tile_base) {
  int64_t t = tile_base + blockIdx.x;
  const int64_t tm = t % tiles_m; t /= tiles_m;
  const int64_t tn = t % tiles_n; t /= tiles_n;
  word *base        = C + t * cbs + tm * tile_rows * cs + tn * tw;
  const int64_t rows = (m - tm * tile_rows) < tile_rows ? (m - tm * tile_rows) : tile_rows;
  const int64_t w    = (wn - tn * tw) < tw ? (wn - tn * tw) : tw;
  for (int64_t i = threadIdx.x; i < rows * w; i += AUX_THREADS) base[(i / w) * cs + (i % w)] = 0;
}

// zero the bits at column >= ncols of the last valid word of every row (establishes the engine's
// "zero excess" invariant for operands uploaded from windows, mzd.h:117-123)
__global__ __launch_bounds__(AUX_THREADS) void mask_tail_kernel(word *__restrict__ M, int64_t stride,
                                                                int64_t rows, int64_t w, word mask) {
  const int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x;
  if (i < rows) M[i * stride + (w - 1)] &= mask;
}

// row r, word j of the matrix := splitmix64 output number r*w + j of the stream seeded `seed`,
// last word masked: the fill order of mzd_randomize_custom (mzd.c:1282-1292) with a counter-based
// generator, so device and host fills (m4ri_amd/mzd.py fill_splitmix) are bit-identical.
__global__ __launch_bounds__(AUX_THREADS) void fill_splitmix_kernel(word *__restrict__ M, int64_t stride,
                                                                    int64_t rows, int64_t w, word mask,
                                                                    uint64_t seed, int64_t first) {
  const int64_t total  = rows * w;
  const int64_t gstr   = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += gstr) {
    const int64_t r = i / w, j = i - r * w;
    uint64_t z = seed + (uint64_t)(first + i + 1) * 0x9E3779B97F4A7C15ull;  // first = row0 * w: rows of a larger matrix
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    if (j == w - 1) z &= mask;
    M[r * stride + j] = z;
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t g = (total + AUX_THREADS - 1) / AUX_THREADS;
  if (g > 256 * 16) {
    // at most 16 workgroups per CU, grid-stride the rest -- in EQUAL trips (a plain cap leaves 7168 workgroups' worth of words to
    // 4096 workgroups, three quarters of them looping twice).  Measured on the three-level passes of a four-level schedule, which
    // run 10 % below their three-level rate: down3 1.275 -> 1.278 ms, up3 1.159 -> 1.139 ms -- the trips are not the reason (the
    // 512-byte rows of the smaller children are), balanced trips are simply the tidier launch
    // (profiles/r03_depth4_trace.summary.txt, r03_depth4_trace_balanced_trips.summary.txt)
    const int64_t trips = (g + 256 * 16 - 1) / (256 * 16);
    g                   = (g + trips - 1) / trips;
  }
  if (g < 1) g = 1;
  return (unsigned)g;
}

// measurement knob: M4RI_AMD_PASS_NT = bit mask of three-level passes that use nontemporal accesses for the 343-way side
// (1 down3 stores, 2 up3 loads, 4 down3_pack stores)
inline int pass_nt() {
  static const int v = getenv("M4RI_AMD_PASS_NT") ? atoi(getenv("M4RI_AMD_PASS_NT")) : PASS_NT_DEFAULT;
  return v;
}

inline bool vec_ok(const void *p, int64_t stride, int64_t bs, int64_t cw, int64_t qcols) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (stride % 2 == 0) && (bs % 2 == 0) &&
         (cw % 2 == 0) && (qcols % 2 == 0);
}

}  // namespace

typedef unsigned long long __attribute__((ext_vector_type(2))) word2;

extern "C" hipError_t gf2_launch_winograd_down(hipStream_t s, int bside, const word *parent,
                                               int64_t p_stride, int64_t p_bs, word *child,
                                               int64_t nparents, int64_t crows, int64_t cw) {
  // child matrices: crows x cw words, contiguous (stride cw), batch stride crows*cw
  const int64_t c_bs = crows * cw;
  if (nparents * c_bs == 0) return hipSuccess;
  if (vec_ok(parent, p_stride, p_bs, cw, cw) && vec_ok(child, cw, c_bs, cw, cw)) {
    const int64_t total = nparents * crows * (cw / 2);
    if (bside)
      hipLaunchKernelGGL((winograd_down_kernel<word2, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(parent), p_stride / 2, p_bs / 2,
                         reinterpret_cast<word2 *>(child), c_bs / 2, nparents, crows, cw / 2, crows, cw / 2);
    else
      hipLaunchKernelGGL((winograd_down_kernel<word2, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(parent), p_stride / 2, p_bs / 2,
                         reinterpret_cast<word2 *>(child), c_bs / 2, nparents, crows, cw / 2, crows, cw / 2);
  } else {
    const int64_t total = nparents * crows * cw;
    if (bside)
      hipLaunchKernelGGL((winograd_down_kernel<word, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         parent, p_stride, p_bs, child, c_bs, nparents, crows, cw, crows, cw);
    else
      hipLaunchKernelGGL((winograd_down_kernel<word, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         parent, p_stride, p_bs, child, c_bs, nparents, crows, cw, crows, cw);
  }
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_winograd_up(hipStream_t s, int acc, const word *prod, word *parent,
                                             int64_t o_stride, int64_t o_bs, int64_t nparents,
                                             int64_t crows, int64_t cw) {
  const int64_t p_bs = crows * cw;
  if (nparents * p_bs == 0) return hipSuccess;
  if (vec_ok(prod, cw, p_bs, cw, cw) && vec_ok(parent, o_stride, o_bs, cw, cw)) {
    const int64_t total = nparents * crows * (cw / 2);
    if (acc)
      hipLaunchKernelGGL((winograd_up_kernel<word2, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(prod), p_bs / 2, reinterpret_cast<word2 *>(parent),
                         o_stride / 2, o_bs / 2, nparents, crows, cw / 2, crows, cw / 2);
    else
      hipLaunchKernelGGL((winograd_up_kernel<word2, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(prod), p_bs / 2, reinterpret_cast<word2 *>(parent),
                         o_stride / 2, o_bs / 2, nparents, crows, cw / 2, crows, cw / 2);
  } else {
    const int64_t total = nparents * crows * cw;
    if (acc)
      hipLaunchKernelGGL((winograd_up_kernel<word, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         prod, p_bs, parent, o_stride, o_bs, nparents, crows, cw, crows, cw);
    else
      hipLaunchKernelGGL((winograd_up_kernel<word, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         prod, p_bs, parent, o_stride, o_bs, nparents, crows, cw, crows, cw);
  }
  return hipGetLastError();
}

// op 0: C = A ^ B, 1: C = A, 2: C = 0  (whole words of `w` words per row)
extern "C" hipError_t gf2_launch_rowwise(hipStream_t s, int op, word *C, int64_t cs, const word *A,
                                         int64_t as, const word *B, int64_t bs, int64_t rows, int64_t w) {
  if (rows * w == 0) return hipSuccess;
  hipLaunchKernelGGL(rowwise_kernel, dim3(grid_for(rows * w)), dim3(AUX_THREADS), 0, s, C, cs, A, as, B, bs, rows, w, op, ~(word)0);
  return hipGetLastError();
}

// C = A ^ B on rows x ncols BITS with _mzd_add's edge rule (mzd.c:1471-1583, :1489): the last word of
// every row is written under the column mask, the other bits of C's last word are kept
extern "C" hipError_t gf2_launch_xor_masked(hipStream_t s, word *C, int64_t cs, const word *A, int64_t as, const word *B,
                                            int64_t bs, int64_t rows, int64_t ncols) {
  if (rows <= 0 || ncols <= 0) return hipSuccess;
  const int64_t w = words_of(ncols);
  const word mask = (ncols % 64) ? ((~(word)0) >> (64 - ncols % 64)) : ~(word)0;
  hipLaunchKernelGGL(rowwise_kernel, dim3(grid_for(rows * w)), dim3(AUX_THREADS), 0, s, C, cs, A, as, B, bs, rows, w, 0, mask);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_copy_masked(hipStream_t s, word *C, int64_t cs, const word *A, int64_t as, int64_t rows,
                                             int64_t ncols) {
  if (rows == 0 || ncols == 0) return hipSuccess;
  const int64_t w = words_of(ncols);
  const word mask = (ncols % 64) ? ((~(word)0) >> (64 - ncols % 64)) : ~(word)0;
  hipLaunchKernelGGL(copy_masked_kernel, dim3(grid_for(rows * w)), dim3(AUX_THREADS), 0, s, C, cs, A, as, rows, w, mask);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_reduce_partials(hipStream_t s, int acc, word *C, int64_t cs, int64_t cbs, int64_t m, int64_t wn,
                                                 int64_t tile_rows, int64_t tw, int64_t tiles_m, int64_t tiles_n,
                                                 int64_t tile_base, int64_t ntiles, int ks, const word *Cpart) {
  if (ntiles <= 0) return hipSuccess;
  if (acc)
    hipLaunchKernelGGL((reduce_partials_kernel<true>), dim3((unsigned)(ntiles * RP_CHUNKS)), dim3(AUX_THREADS), 0, s, C, cs, cbs, m, wn, tile_rows,
                       tw, tiles_m, tiles_n, tile_base, ks, Cpart);
  else
    hipLaunchKernelGGL((reduce_partials_kernel<false>), dim3((unsigned)(ntiles * RP_CHUNKS)), dim3(AUX_THREADS), 0, s, C, cs, cbs, m, wn, tile_rows,
                       tw, tiles_m, tiles_n, tile_base, ks, Cpart);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_zero_tiles(hipStream_t s, word *C, int64_t cs, int64_t cbs, int64_t m, int64_t wn,
                                            int64_t tile_rows, int64_t tw, int64_t tiles_m, int64_t tiles_n,
                                            int64_t tile_base, int64_t ntiles) {
  if (ntiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(zero_tiles_kernel, dim3((unsigned)ntiles), dim3(AUX_THREADS), 0, s, C, cs, cbs, m, wn, tile_rows, tw,
                     tiles_m, tiles_n, tile_base);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_mask_tail(hipStream_t s, word *M, int64_t stride, int64_t rows, int64_t ncols) {
  if (rows == 0 || ncols == 0 || (ncols % 64) == 0) return hipSuccess;
  const word mask = (~(word)0) >> (64 - ncols % 64);
  hipLaunchKernelGGL(mask_tail_kernel, dim3((unsigned)((rows + AUX_THREADS - 1) / AUX_THREADS)), dim3(AUX_THREADS), 0, s,
                     M, stride, rows, words_of(ncols), mask);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_fill_splitmix_rows(hipStream_t s, word *M, int64_t stride, int64_t row0, int64_t rows,
                                                    int64_t ncols, uint64_t seed) {
  if (rows == 0 || ncols == 0) return hipSuccess;
  const int64_t w = words_of(ncols);
  const word mask = (ncols % 64) ? ((~(word)0) >> (64 - ncols % 64)) : ~(word)0;
  hipLaunchKernelGGL(fill_splitmix_kernel, dim3(grid_for(rows * w)), dim3(AUX_THREADS), 0, s, M, stride, rows, w, mask, seed, row0 * w);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_fill_splitmix(hipStream_t s, word *M, int64_t stride, int64_t rows,
                                               int64_t ncols, uint64_t seed) {
  return gf2_launch_fill_splitmix_rows(s, M, stride, 0, rows, ncols, seed);
}

// ---- two levels down on the A side, written straight into the leaf's packed form ------------------
// The M4RM leaf (generations 3 and 4) reads A chunk-major: A4[q][r] = the q-th 32-bit chunk of row r
// (a4_pack.hip).  Producing that layout here saves the separate pack pass -- one read and one
// write of all 7^L leaf operands, the largest single item of the schedule's HBM traffic.  A
// workgroup owns a 32-row x 32-chunk tile of the grandchild grid; every one of its 49 outputs goes
// through a (double-buffered) LDS transpose so that both the reads of the grandparent (128 B per
// row and block) and the writes of the packed rows (128 B per chunk) are coalesced.
namespace {
constexpr int DP_ROWS = 32, DP_V = 8, DP_PITCH = 36;  // tile rows, 16-byte vectors per row, LDS row pitch (dwords)

template <int ROT>  // 0: plain chunk-major, 1: generation 4's byte rotation, 2: generation 5's rotation + chunk swap
__global__ __launch_bounds__(AUX_THREADS) void winograd_down2_pack_kernel(
    const word2 *__restrict__ gparent, int64_t p_stride, int64_t p_bs,  // grandparent array (units of word2)
    uint32_t *__restrict__ a4, int64_t a4_bs,                           // packed grandchildren, a4_bs dwords each
    int64_t crows, int64_t cw2, int64_t tiles_r, int64_t tiles_w) {     // grandchild: crows x cw2 word2
  __shared__ __attribute__((aligned(16))) uint32_t tile[2][DP_ROWS * DP_PITCH];
  int64_t bid      = blockIdx.x;
  const int64_t wt = bid % tiles_w; bid /= tiles_w;
  const int64_t rt = bid % tiles_r; bid /= tiles_r;
  const int64_t pi = bid;
  const int t = threadIdx.x, r = t >> 3, v = t & 7;
  const word2 *p = gparent + pi * p_bs + (rt * DP_ROWS + r) * p_stride + (wt * DP_V + v);
  word2 x[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) x[a][b] = p[(int64_t)a * crows * p_stride + (int64_t)b * cw2];
  word2 y[4][7];  // y[2a+b][j1]: block (a, b) of child j1
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) winograd_combos<word2, false>(x[a][b], x[a][b + 2], x[a + 2][b], x[a + 2][b + 2], y[2 * a + b]);
  // write side: thread -> chunk q (0..31 of the tile) and four consecutive rows
  const int q = t >> 3, r4 = (t & 7) * 4;
  // generation 4 wants the four index bytes of a dword rotated by (row >> 6) & 3 (m4rm8q_leaf.hip),
  // generation 5 by (row >> 7) & 3 with the two chunks of a word swapped when (row >> 5) & 1
  // (m4rm8o_leaf.hip); the four rows of one store share those values
  const int64_t orow = rt * DP_ROWS + r4;
  const uint32_t rot = ROT == 1 ? (uint32_t)((orow >> 6) & 3) : ROT == 2 ? (uint32_t)((orow >> 7) & 3) : 0u;
  const int64_t oq   = (wt * (DP_V * 4) + q) ^ (ROT == 2 ? ((orow >> 5) & 1) : 0);
  uint32_t *o = a4 + (pi * 49) * a4_bs + oq * crows + orow;
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word2 g[7];
    winograd_combos<word2, false>(y[0][j1], y[1][j1], y[2][j1], y[3][j1], g);
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      uint32_t *tb = tile[(7 * j1 + j2) & 1];
      *reinterpret_cast<word2 *>(tb + r * DP_PITCH + 4 * v) = g[j2];
      __syncthreads();  // one barrier per output: the other buffer is only rewritten after the next one
      uint4 w;
      w.x = tb[(r4 + 0) * DP_PITCH + q]; w.y = tb[(r4 + 1) * DP_PITCH + q];
      w.z = tb[(r4 + 2) * DP_PITCH + q]; w.w = tb[(r4 + 3) * DP_PITCH + q];
      if (ROT) {
        w.x = __builtin_amdgcn_alignbyte(w.x, w.x, rot); w.y = __builtin_amdgcn_alignbyte(w.y, w.y, rot);
        w.z = __builtin_amdgcn_alignbyte(w.z, w.z, rot); w.w = __builtin_amdgcn_alignbyte(w.w, w.w, rot);
      }
      *reinterpret_cast<uint4 *>(o + (int64_t)(7 * j1 + j2) * a4_bs) = w;
    }
  }
}
}  // namespace

// Grandchild i of the pass lands at a4 + i * (crows * cw * 2) dwords, laid out exactly as
// gf2_launch_a4_pack(_rot) would have packed the row-major grandchild.  Returns
// hipErrorInvalidValue when the shape does not tile (caller falls back to down2 + pack).
extern "C" int gf2_winograd_down2_pack_ok(const word *gparent, int64_t p_stride, int64_t p_bs, const word *a4,
                                          int64_t crows, int64_t cw) {
  return vec_ok(gparent, p_stride, p_bs, cw, cw) && crows > 0 && cw > 0 && crows % DP_ROWS == 0 && cw % (2 * DP_V) == 0 &&
         (reinterpret_cast<uintptr_t>(a4) & 15) == 0;
}

extern "C" hipError_t gf2_launch_winograd_down2_pack(hipStream_t s, const word *gparent, int64_t p_stride, int64_t p_bs,
                                                     word *a4, int64_t nparents, int64_t crows, int64_t cw, int rot) {
  if (nparents * crows * cw == 0) return hipSuccess;
  if (!gf2_winograd_down2_pack_ok(gparent, p_stride, p_bs, a4, crows, cw)) return hipErrorInvalidValue;
  const int64_t tiles_r = crows / DP_ROWS, tiles_w = (cw / 2) / DP_V;
  const int64_t grid = nparents * tiles_r * tiles_w;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
#define DP2_LAUNCH(R)                                                                                         \
  hipLaunchKernelGGL((winograd_down2_pack_kernel<R>), dim3((unsigned)grid), dim3(AUX_THREADS), 0, s,                \
                     reinterpret_cast<const word2 *>(gparent), p_stride / 2, p_bs / 2, reinterpret_cast<uint32_t *>(a4), \
                     crows * cw * 2, crows, cw / 2, tiles_r, tiles_w)
  if (rot == 2) DP2_LAUNCH(2); else if (rot == 1) DP2_LAUNCH(1); else DP2_LAUNCH(0);
#undef DP2_LAUNCH
  return hipGetLastError();
}

// ---- three levels at once ----------------------------------------------------------------------
// The whole schedule of a 3-level product in one pass each way: an ancestor is read once as an 8 x 8
// grid of blocks and its 343 great-grandchildren are written directly (index 49*j1 + 7*j2 + j3,
// exactly what three single-level passes produce); on the way up 343 products become 64 blocks.
// Neither intermediate level is materialised: per operand 1 + 343/64 block transfers instead of
// (1 + 7/4) + (7/4 + 343/64) with one single + one double pass.  One 64-bit word per lane (64
// operand words live); children are produced one at a time to keep the register count down.
namespace {

// child j (0..6) of a 2 x 2 split, in the order of winograd_combos
template <typename V, bool BSIDE>
__device__ __forceinline__ V winograd_child(const V x11, const V x12, const V x21, const V x22, int j) {
  if (!BSIDE) {  // [A11, A12, S4, A22, S1, S2, S3]
    switch (j) {
      case 0: return x11;
      case 1: return x12;
      case 2: return x12 ^ x21 ^ x22 ^ x11;
      case 3: return x22;
      case 4: return x21 ^ x22;
      case 5: return x21 ^ x22 ^ x11;
      default: return x11 ^ x21;
    }
  } else {       // [B11, B21, B22, T4, T1, T2, T3]
    switch (j) {
      case 0: return x11;
      case 1: return x21;
      case 2: return x22;
      case 3: return x22 ^ x12 ^ x11 ^ x21;
      case 4: return x12 ^ x11;
      case 5: return x22 ^ x12 ^ x11;
      default: return x22 ^ x12;
    }
  }
}

// product j's contribution to the four quadrants (the transpose of winograd_recombine):
// C11 = P1+P2, C12 = P1+P6+P5+P3, C21 = P1+P6+P7+P4, C22 = P1+P6+P7+P5
template <typename V>
__device__ __forceinline__ void winograd_scatter(const V pr, int j, V &c11, V &c12, V &c21, V &c22) {
  switch (j) {
    case 0: c11 ^= pr; c12 ^= pr; c21 ^= pr; c22 ^= pr; break;
    case 1: c11 ^= pr; break;
    case 2: c12 ^= pr; break;
    case 3: c21 ^= pr; break;
    case 4: c12 ^= pr; c22 ^= pr; break;
    case 5: c12 ^= pr; c21 ^= pr; c22 ^= pr; break;
    default: c21 ^= pr; c22 ^= pr; break;
  }
}

template <bool BSIDE, bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_down3_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,  // ancestor array
    word *__restrict__ gchild, int64_t c_bs,                       // great-grandchildren, stride == cw
    int64_t nparents, int64_t crows, int64_t cw) {                 // great-grandchild shape: crows x cw
  const int64_t per    = crows * cw;
  const int64_t total  = nparents * per;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t pi = i / per;
    const int64_t rm = i - pi * per;
    const int64_t r  = rm / cw;
    const int64_t w  = rm - r * cw;
    const word *p    = anc + pi * p_bs + r * p_stride + w;
    word x[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) x[a][b] = p[(int64_t)a * crows * p_stride + (int64_t)b * cw];
    word *c = gchild + (pi * 343) * c_bs + r * cw + w;
#pragma unroll
    for (int j1 = 0; j1 < 7; ++j1) {
      word c1[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) c1[a][b] = winograd_child<word, BSIDE>(x[a][b], x[a][b + 4], x[a + 4][b], x[a + 4][b + 4], j1);
#pragma unroll
      for (int j2 = 0; j2 < 7; ++j2) {
        word c2[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) c2[a][b] = winograd_child<word, BSIDE>(c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2], j2);
#pragma unroll
        for (int j3 = 0; j3 < 7; ++j3) {
          const word v = winograd_child<word, BSIDE>(c2[0][0], c2[0][1], c2[1][0], c2[1][1], j3);
          if (NT) __builtin_nontemporal_store(v, &c[(int64_t)(49 * j1 + 7 * j2 + j3) * c_bs]);
          else c[(int64_t)(49 * j1 + 7 * j2 + j3) * c_bs] = v;
        }
      }
    }
  }
}

template <bool ACC, bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_up3_kernel(
    const word *__restrict__ prod, int64_t p_bs,                   // 343 products per ancestor, stride == cw
    word *__restrict__ anc, int64_t o_stride, int64_t o_bs,        // ancestor array
    int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t per    = crows * cw;
  const int64_t total  = nparents * per;
  const int64_t stride = (int64_t)gridDim.x * AUX_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AUX_THREADS + threadIdx.x; i < total; i += stride) {
    const int64_t pi = i / per;
    const int64_t rm = i - pi * per;
    const int64_t r  = rm / cw;
    const int64_t w  = rm - r * cw;
    const word *q    = prod + (pi * 343) * p_bs + r * cw + w;
    word out[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) out[a][b] = 0;
#pragma unroll
    for (int j1 = 0; j1 < 7; ++j1) {
      word c1[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) c1[a][b] = 0;
#pragma unroll
      for (int j2 = 0; j2 < 7; ++j2) {
        word c2[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
        for (int j3 = 0; j3 < 7; ++j3)
          winograd_scatter<word>(NT ? __builtin_nontemporal_load(&q[(int64_t)(49 * j1 + 7 * j2 + j3) * p_bs]) : q[(int64_t)(49 * j1 + 7 * j2 + j3) * p_bs],
                                 j3, c2[0][0], c2[0][1], c2[1][0], c2[1][1]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) winograd_scatter<word>(c2[a][b], j2, c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2]);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) winograd_scatter<word>(c1[a][b], j1, out[a][b], out[a][b + 4], out[a + 4][b], out[a + 4][b + 4]);
    }
    word *o = anc + pi * o_bs + r * o_stride + w;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        word *oo = o + (int64_t)a * crows * o_stride + (int64_t)b * cw;
        *oo      = ACC ? (*oo ^ out[a][b]) : out[a][b];
      }
  }
}

// A side, written in the leaf's packed chunk-major form (see winograd_down2_pack_kernel): 512
// threads own a 32-row x 16-word tile of the great-grandchild grid; each of the 343 outputs goes
// through a double-buffered LDS transpose (8-byte rows in, 8-byte row pairs of one chunk out).
// LDS layout: 32 rows x 32 dwords, the 8-byte column slot of row r XOR-swizzled by (r >> 1) & 15.  A padded
// pitch (34) made the transposed reads 2-way conflicted: the 16 row pairs a 32-lane read group touches sit
// 68 dwords apart, i.e. on only 8 distinct banks (SQ_LDS_BANK_CONFLICT = 33 % of the kernel's LDS cycles,
// profiles/r01d_bench65536_pmc_lds.summary.txt).  With the swizzle the 16 row pairs of one chunk land in 16
// different slots and the group's two chunks (q even / odd) in the two dwords of a slot: 32 banks, no conflict;
// the writes (16 lanes = the 16 slots of one row) stay conflict-free and 8-byte aligned.
constexpr int DP3_ROWS = 32, DP3_W = 16, DP3_PITCH = 32, DP3_THREADS = 512;

template <int ROT, bool NT>
__global__ __launch_bounds__(DP3_THREADS) void winograd_down3_pack_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,
    uint32_t *__restrict__ a4, int64_t a4_bs,                       // packed great-grandchildren, a4_bs dwords each
    int64_t crows, int64_t cw, int64_t tiles_r, int64_t tiles_w) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[2][7 * DP3_ROWS * DP3_PITCH];  // two sets of seven 4 KiB tiles
  int64_t bid      = blockIdx.x;
  const int64_t wt = bid % tiles_w; bid /= tiles_w;
  const int64_t rt = bid % tiles_r; bid /= tiles_r;
  const int64_t pi = bid;
  const int t = threadIdx.x, r = t >> 4, v = t & 15;
  const word *p = anc + pi * p_bs + (rt * DP3_ROWS + r) * p_stride + (wt * DP3_W + v);
  word x[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) x[a][b] = p[(int64_t)a * crows * p_stride + (int64_t)b * cw];
  // write side: thread -> chunk q (0..31 of the tile) and two consecutive rows
  const int q = t >> 4, r2 = (t & 15) * 2;
  const int64_t orow = rt * DP3_ROWS + r2;  // both rows share the rotation / swap
  const uint32_t rot = ROT == 1 ? (uint32_t)((orow >> 6) & 3) : ROT == 2 ? (uint32_t)((orow >> 7) & 3) : 0u;
  const int64_t oq   = (wt * (DP3_W * 2) + q) ^ (ROT == 2 ? ((orow >> 5) & 1) : 0);
  uint32_t *o = a4 + (pi * 343) * a4_bs + oq * crows + orow;
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word c1[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) c1[a][b] = winograd_child<word, false>(x[a][b], x[a][b + 4], x[a + 4][b], x[a + 4][b + 4], j1);
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      word c2[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c2[a][b] = winograd_child<word, false>(c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2], j2);
      // the seven outputs of (j1, j2) go through seven tiles of one of two sets and ONE barrier: the other set is only rewritten after the
      // next barrier, when every thread has read this one (49 barriers per workgroup instead of 343: with one 512-thread workgroup per
      // CU a barrier per 4 KiB of output was a visible part of the pass)
      uint32_t *set = tile[(7 * j1 + j2) & 1];
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3)
        *reinterpret_cast<word *>(set + j3 * (DP3_ROWS * DP3_PITCH) + r * DP3_PITCH + 2 * (v ^ ((r >> 1) & 15))) =
            winograd_child<word, false>(c2[0][0], c2[0][1], c2[1][0], c2[1][1], j3);
      __syncthreads();
      const int rs = 2 * ((q >> 1) ^ ((r2 >> 1) & 15)) + (q & 1);  // rows r2 and r2 + 1 share the swizzle
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) {
        const int k        = 49 * j1 + 7 * j2 + j3;
        const uint32_t *tb = set + j3 * (DP3_ROWS * DP3_PITCH);
        uint32_t w0 = tb[(r2 + 0) * DP3_PITCH + rs], w1 = tb[(r2 + 1) * DP3_PITCH + rs];
        if (ROT) { w0 = __builtin_amdgcn_alignbyte(w0, w0, rot); w1 = __builtin_amdgcn_alignbyte(w1, w1, rot); }
        if (NT) __builtin_nontemporal_store((unsigned long long)w0 | ((unsigned long long)w1 << 32), reinterpret_cast<unsigned long long *>(o + (int64_t)k * a4_bs));
        else *reinterpret_cast<uint2 *>(o + (int64_t)k * a4_bs) = make_uint2(w0, w1);
      }
    }
  }
}
}  // namespace

extern "C" hipError_t gf2_launch_winograd_down3(hipStream_t s, int bside, const word *anc, int64_t p_stride, int64_t p_bs,
                                                word *gchild, int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t c_bs = crows * cw;
  if (nparents * c_bs == 0) return hipSuccess;
  const int64_t total = nparents * c_bs;
#define D3_LAUNCH(BS, NT)                                                                                                      \
  hipLaunchKernelGGL((winograd_down3_kernel<BS, NT>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s, anc, p_stride, p_bs, gchild, c_bs, \
                     nparents, crows, cw)
  if (pass_nt() & 1) { if (bside) D3_LAUNCH(true, true); else D3_LAUNCH(false, true); }
  else { if (bside) D3_LAUNCH(true, false); else D3_LAUNCH(false, false); }
#undef D3_LAUNCH
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_winograd_up3(hipStream_t s, int acc, const word *prod, word *anc, int64_t o_stride,
                                              int64_t o_bs, int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t p_bs = crows * cw;
  if (nparents * p_bs == 0) return hipSuccess;
  const int64_t total = nparents * p_bs;
#define U3_LAUNCH(AC, NT)                                                                                                            \
  hipLaunchKernelGGL((winograd_up3_kernel<AC, NT>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, nparents, \
                     crows, cw)
  if (pass_nt() & 2) { if (acc) U3_LAUNCH(true, true); else U3_LAUNCH(false, true); }
  else { if (acc) U3_LAUNCH(true, false); else U3_LAUNCH(false, false); }
#undef U3_LAUNCH
  return hipGetLastError();
}

extern "C" int gf2_winograd_down3_pack_ok(const word *a4, int64_t crows, int64_t cw) {
  return crows > 0 && cw > 0 && crows % DP3_ROWS == 0 && cw % DP3_W == 0 && (reinterpret_cast<uintptr_t>(a4) & 7) == 0;
}

// Great-grandchild i of the pass lands at a4 + i * (crows * cw * 2) dwords, laid out exactly as
// gf2_launch_a4_pack(_rot) would have packed the row-major great-grandchild.
extern "C" hipError_t gf2_launch_winograd_down3_pack(hipStream_t s, const word *anc, int64_t p_stride, int64_t p_bs,
                                                     word *a4, int64_t nparents, int64_t crows, int64_t cw, int rot) {
  if (nparents * crows * cw == 0) return hipSuccess;
  if (!gf2_winograd_down3_pack_ok(a4, crows, cw)) return hipErrorInvalidValue;
  const int64_t tiles_r = crows / DP3_ROWS, tiles_w = cw / DP3_W;
  const int64_t grid = nparents * tiles_r * tiles_w;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
#define DP3_LAUNCH(R, NT)                                                                                              \
  hipLaunchKernelGGL((winograd_down3_pack_kernel<R, NT>), dim3((unsigned)grid), dim3(DP3_THREADS), 0, s, anc, p_stride, p_bs, \
                     reinterpret_cast<uint32_t *>(a4), crows * cw * 2, crows, cw, tiles_r, tiles_w)
  if (pass_nt() & 4) { if (rot == 2) DP3_LAUNCH(2, true); else if (rot == 1) DP3_LAUNCH(1, true); else DP3_LAUNCH(0, true); }
  else { if (rot == 2) DP3_LAUNCH(2, false); else if (rot == 1) DP3_LAUNCH(1, false); else DP3_LAUNCH(0, false); }
#undef DP3_LAUNCH
  return hipGetLastError();
}

// ---- four levels at once ---------------------------------------------------------------------------
// The three-level passes with ONE MORE level on top, formed on the fly: the ancestor is a 16 x 16 grid of blocks; workgroup
// (position block, j0) builds block (a, b), a, b < 8, of top-level child j0 from the up to four grid blocks (a, b), (a, b + 8),
// (a + 8, b), (a + 8, b + 8) while it loads, then runs the three-level expansion of that child in registers and writes its 343
// outputs at index 343 * j0 + 49 * j1 + 7 * j2 + j3 -- exactly what a single-level pass followed by a three-level pass produces, without
// the 7/4-size intermediate level ever being written or read.  The seven workgroups of one position block read overlapping
// grid blocks (14 quadrant reads for 4 quadrants); they are numbered 8 apart, so they run on the SAME XCD (workgroup b -> XCD
// b % 8) within a few dispatches of each other and meet in its L2: HBM sees every ancestor word about once.
// Up: the three-level recombination of the 343 products of top-level product j0 gives its 8 x 8 blocks, which are scattered into the
// quadrants that product belongs to (winograd_scatter) by no-return L2 atomic XOR into a zeroed (or, accumulating, the caller's) C.
namespace {

// blockIdx.x -> (position block, j0): b = (hi * 7 + j0) * 8 + lo, position block = hi * 8 + lo
__device__ __forceinline__ void pass4_block(int64_t b, int64_t &pb, int &j0) {
  const int64_t lo = b & 7, rest = b >> 3;
  j0 = (int)(rest % 7);
  pb = (rest / 7) * 8 + lo;
}
inline int64_t pass4_grid(int64_t nposblocks) { return ((nposblocks + 7) / 8) * 7 * 8; }

// block (a, b) of top-level child j (wave-uniform) of the 2 x 2 split whose quadrants are `qr` rows / `qc` words apart: only the
// quadrants the child needs are loaded
template <bool BSIDE>
__device__ __forceinline__ word load_top_child(const word *q11, int64_t qr, int64_t qc, int j) {
  const word *q12 = q11 + qc, *q21 = q11 + qr, *q22 = q11 + qr + qc;
  if (!BSIDE) {  // [A11, A12, S4, A22, S1, S2, S3]
    switch (j) {
      case 0: return *q11;
      case 1: return *q12;
      case 2: return *q12 ^ *q21 ^ *q22 ^ *q11;
      case 3: return *q22;
      case 4: return *q21 ^ *q22;
      case 5: return *q21 ^ *q22 ^ *q11;
      default: return *q11 ^ *q21;
    }
  } else {       // [B11, B21, B22, T4, T1, T2, T3]
    switch (j) {
      case 0: return *q11;
      case 1: return *q21;
      case 2: return *q22;
      case 3: return *q22 ^ *q12 ^ *q11 ^ *q21;
      case 4: return *q12 ^ *q11;
      case 5: return *q22 ^ *q12 ^ *q11;
      default: return *q22 ^ *q12;
    }
  }
}

// (Tried and measured, profiles/r04_depth4_fused/: ONE workgroup of seven waves per 64 positions, wave = j0, so that the siblings'
// loads are issued at the same moment -- slower, down4 1.37 -> 1.46 ms, up4 1.34 -> 1.61 ms: one 448-thread workgroup per CU.)
template <bool BSIDE, bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_down4_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,  // ancestor array: 16 * crows rows x 16 * cw words each
    word *__restrict__ gchild, int64_t c_bs,                       // 2401 descendants per ancestor, stride == cw
    int64_t crows, int64_t cw, int64_t nposblocks) {
  int64_t pb;
  int j0;
  pass4_block(blockIdx.x, pb, j0);
  const int64_t i = pb * AUX_THREADS + threadIdx.x, pi = blockIdx.y;
  if (pb >= nposblocks || i >= crows * cw) return;
  const int64_t r = i / cw, w = i - r * cw;
  const word *p   = anc + pi * p_bs + r * p_stride + w;
  word x[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) x[a][b] = load_top_child<BSIDE>(p + (int64_t)a * crows * p_stride + (int64_t)b * cw, 8 * crows * p_stride, 8 * cw, j0);
  word *c = gchild + (pi * 2401 + 343 * j0) * c_bs + r * cw + w;
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word c1[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) c1[a][b] = winograd_child<word, BSIDE>(x[a][b], x[a][b + 4], x[a + 4][b], x[a + 4][b + 4], j1);
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      word c2[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c2[a][b] = winograd_child<word, BSIDE>(c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2], j2);
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) {
        const word v = winograd_child<word, BSIDE>(c2[0][0], c2[0][1], c2[1][0], c2[1][1], j3);
        if (NT) __builtin_nontemporal_store(v, &c[(int64_t)(49 * j1 + 7 * j2 + j3) * c_bs]);
        else c[(int64_t)(49 * j1 + 7 * j2 + j3) * c_bs] = v;
      }
    }
  }
}

// out[a][b] ^-> block (a, b) of the quadrants T11 ... T22 of the ancestor (no-return L2 atomics)
template <bool T11, bool T12, bool T21, bool T22>
__device__ __forceinline__ void up4_scatter(const word (&out)[8][8], char *ob, uint32_t lane_b, int64_t qr, int64_t qc, int64_t br, int64_t bc) {
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      char *bb = ob + a * br + b * bc;  // wave-uniform
      const unsigned long long v = out[a][b];
      if (T11) atomicXor(reinterpret_cast<unsigned long long *>(bb + lane_b), v);
      if (T12) atomicXor(reinterpret_cast<unsigned long long *>(bb + qc + lane_b), v);
      if (T21) atomicXor(reinterpret_cast<unsigned long long *>(bb + qr + lane_b), v);
      if (T22) atomicXor(reinterpret_cast<unsigned long long *>(bb + qr + qc + lane_b), v);
    }
}

template <bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_up4_kernel(
    const word *__restrict__ prod, int64_t p_bs,            // 2401 products per ancestor, stride == cw
    word *__restrict__ anc, int64_t o_stride, int64_t o_bs, // ancestor array, zeroed (or holding the matrix to accumulate onto)
    int64_t crows, int64_t cw, int64_t nposblocks) {
  int64_t pb;
  int j0;
  pass4_block(blockIdx.x, pb, j0);
  const int64_t i = pb * AUX_THREADS + threadIdx.x, pi = blockIdx.y;
  if (pb >= nposblocks || i >= crows * cw) return;
  const int64_t r = i / cw, w = i - r * cw;
  const word *q   = prod + (pi * 2401 + 343 * j0) * p_bs + r * cw + w;
  word out[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) out[a][b] = 0;
  // The 343 products are consumed in 49 groups of 7 (one j2 each), software-pipelined by hand: group g + 1 is loaded, then group g
  // folded.  The scheduling barriers keep the memory operations in this order -- left alone, the compiler issues all 343 loads up
  // front and spills most of them (it does not in the three-level kernel, whose tail is plain stores)
  word buf[2][7];
  auto load_group = [&](int g, word (&dst)[7]) {
#pragma unroll
    for (int j3 = 0; j3 < 7; ++j3) dst[j3] = NT ? __builtin_nontemporal_load(&q[(int64_t)(7 * g + j3) * p_bs]) : q[(int64_t)(7 * g + j3) * p_bs];
  };
  load_group(0, buf[0]);
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word c1[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) c1[a][b] = 0;
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      const int g = 7 * j1 + j2;
      if (g + 1 < 49) load_group(g + 1, buf[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // nothing crosses: the loads of group g + 1 are in flight while group g is folded
      word c2[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) winograd_scatter<word>(buf[g & 1][j3], j3, c2[0][0], c2[0][1], c2[1][0], c2[1][1]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) winograd_scatter<word>(c2[a][b], j2, c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) winograd_scatter<word>(c1[a][b], j1, out[a][b], out[a][b + 4], out[a + 4][b], out[a + 4][b + 4]);
  }
  // (the folds stay HERE: without the pins the optimiser sinks the whole XOR tree into each of the seven cases below and keeps all 343
  // loaded products alive across the branch)
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(out[a][b]));
  // top level: product j0 belongs to the quadrants winograd_scatter names; seven workgroups add into every word.  Addresses are a
  // wave-uniform base (block, quadrant) + ONE 32-bit per-lane byte offset, so the 64 x (1 ... 4) atomics need no address registers
  const uint32_t lane_b = (uint32_t)((r * o_stride + w) * 8);  // the launcher keeps one quadrant row range below 2 GiB
  char *ob              = reinterpret_cast<char *>(anc + pi * o_bs);
  const int64_t qr = 8 * crows * o_stride * 8, qc = 8 * cw * 8, br = crows * o_stride * 8, bc = cw * 8;  // bytes
  switch (j0) {
    case 0: up4_scatter<true, true, true, true>(out, ob, lane_b, qr, qc, br, bc); break;
    case 1: up4_scatter<true, false, false, false>(out, ob, lane_b, qr, qc, br, bc); break;
    case 2: up4_scatter<false, true, false, false>(out, ob, lane_b, qr, qc, br, bc); break;
    case 3: up4_scatter<false, false, true, false>(out, ob, lane_b, qr, qc, br, bc); break;
    case 4: up4_scatter<false, true, false, true>(out, ob, lane_b, qr, qc, br, bc); break;
    case 5: up4_scatter<false, true, true, true>(out, ob, lane_b, qr, qc, br, bc); break;
    default: up4_scatter<false, false, true, true>(out, ob, lane_b, qr, qc, br, bc); break;
  }
}

// Down, with the ancestor read ONCE: a workgroup owns 32 word positions, loads their 16 x 16 grid into LDS (64 KiB, every load 256
// contiguous bytes), and after one barrier half-wave g (of 8; the last idles) forms top-level child g from the grid -- up to four
// LDS reads per word, chosen by per-lane masks -- and expands it through the three levels in registers.  (The form above leaves the
// siblings' re-reads to the L2, which absorbs half of them: 0.97 GB read for a 0.54 GB ancestor.)
template <bool BSIDE, bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_down4_lds_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,  // ancestor array: 16 * crows rows x 16 * cw words each
    word *__restrict__ gchild, int64_t c_bs,                       // 2401 descendants per ancestor, stride == cw
    int64_t crows, int64_t cw) {                                   // cw % 32 == 0
  __shared__ word grid[256 * 32];
  const int tid = threadIdx.x, pp = tid & 31, g = tid >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + pp, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  const word *p   = anc + pi * p_bs + r * p_stride + w;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + g;
    grid[blk * 32 + pp] = p[(int64_t)(blk >> 4) * crows * p_stride + (int64_t)(blk & 15) * cw];
  }
  __syncthreads();
  if (g >= 7) return;
  // which quadrants top-level child g is made of (winograd_child's table), as all-ones / zero masks per lane
  const int j0 = g;
  bool u11, u12, u21, u22;
  if (!BSIDE) {  // [A11, A12, S4, A22, S1, S2, S3]
    u11 = j0 == 0 || j0 == 2 || j0 == 5 || j0 == 6; u12 = j0 == 1 || j0 == 2; u21 = j0 == 2 || j0 == 4 || j0 == 5 || j0 == 6; u22 = j0 >= 2 && j0 <= 5;
  } else {       // [B11, B21, B22, T4, T1, T2, T3]
    u11 = j0 == 0 || j0 == 3 || j0 == 4 || j0 == 5; u12 = j0 >= 3; u21 = j0 == 1 || j0 == 3; u22 = j0 == 2 || j0 == 3 || j0 == 5 || j0 == 6;
  }
  const word m11 = u11 ? ~(word)0 : 0, m12 = u12 ? ~(word)0 : 0, m21 = u21 ? ~(word)0 : 0, m22 = u22 ? ~(word)0 : 0;
  word x[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
      x[a][b] = (grid[(a * 16 + b) * 32 + pp] & m11) ^ (grid[(a * 16 + b + 8) * 32 + pp] & m12) ^ (grid[((a + 8) * 16 + b) * 32 + pp] & m21) ^
                (grid[((a + 8) * 16 + b + 8) * 32 + pp] & m22);
  word *c = gchild + (pi * 2401 + 343 * j0) * c_bs + r * cw + w;
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word c1[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) c1[a][b] = winograd_child<word, BSIDE>(x[a][b], x[a][b + 4], x[a + 4][b], x[a + 4][b + 4], j1);
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      word c2[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c2[a][b] = winograd_child<word, BSIDE>(c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2], j2);
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) {
        const word v = winograd_child<word, BSIDE>(c2[0][0], c2[0][1], c2[1][0], c2[1][1], j3);
        if (NT) __builtin_nontemporal_store(v, &c[(int64_t)(49 * j1 + 7 * j2 + j3) * c_bs]);
        else c[(int64_t)(49 * j1 + 7 * j2 + j3) * c_bs] = v;
      }
    }
  }
}

// The A side in the leaf's packed form, the same way -- and without the transpose.  A workgroup owns ONE word column and 32 consecutive
// ROWS: lane = row, so that the two 32-bit chunks of a lane's word land in the packed array A4[chunk][row] as 32 consecutive dwords
// = one full 128-byte line per chunk and child, straight from the registers (no LDS transpose, no barrier per output).  The price is
// paid where it is cheap: the grid is LOADED 8 bytes per row (32 lines touched per block), and the other 15 words of every line belong
// to the workgroups of the neighbouring word columns, which are numbered 8 apart so that they run on the same XCD next to each other and
// find the line in its L2.
template <int ROT, bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_down4_pack_lds_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,
    uint32_t *__restrict__ a4, int64_t a4_bs,       // packed descendants, a4_bs dwords each
    int64_t crows, int64_t cw) {                    // crows % 32 == 0, cw % 16 == 0
  __shared__ word grid[256 * 32];
  const int tid = threadIdx.x, pp = tid & 31, g = tid >> 5;
  // blockIdx -> (row block, word column): 8 row blocks x 16 word columns per group of 128 workgroups, the 16 sharers of a line 8 apart
  const int64_t b = blockIdx.x, grp = b >> 7, xcd = b & 7, slot = (b >> 3) & 15;
  const int64_t wgroups = cw >> 4;                   // groups of 16 word columns per row block
  const int64_t rb = (grp / wgroups) * 8 + xcd, wc = (grp % wgroups) * 16 + slot;
  if (rb * 32 >= crows) return;                      // (whole workgroups: the row blocks of the last group may not all exist)
  const int64_t r = rb * 32 + pp, pi = blockIdx.y;
  const word *p   = anc + pi * p_bs + r * p_stride + wc;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + g;
    grid[blk * 32 + pp] = p[(int64_t)(blk >> 4) * crows * p_stride + (int64_t)(blk & 15) * cw];
  }
  __syncthreads();
  if (g >= 7) return;
  const int j0 = g;   // [A11, A12, S4, A22, S1, S2, S3]
  const bool u11 = j0 == 0 || j0 == 2 || j0 == 5 || j0 == 6, u12 = j0 == 1 || j0 == 2, u21 = j0 == 2 || j0 == 4 || j0 == 5 || j0 == 6, u22 = j0 >= 2 && j0 <= 5;
  const word m11 = u11 ? ~(word)0 : 0, m12 = u12 ? ~(word)0 : 0, m21 = u21 ? ~(word)0 : 0, m22 = u22 ? ~(word)0 : 0;
  word x[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int bb = 0; bb < 8; ++bb)
      x[a][bb] = (grid[(a * 16 + bb) * 32 + pp] & m11) ^ (grid[(a * 16 + bb + 8) * 32 + pp] & m12) ^ (grid[((a + 8) * 16 + bb) * 32 + pp] & m21) ^
                 (grid[((a + 8) * 16 + bb + 8) * 32 + pp] & m22);
  const uint32_t rot = ROT == 1 ? (uint32_t)((r >> 6) & 3) : 0u;   // generation 4's byte rotation of the index dwords (m4rm8q_leaf.hip)
  uint32_t *o = a4 + (pi * 2401 + 343 * j0) * a4_bs + (2 * wc) * crows + r;
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word c1[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) c1[a][bb] = winograd_child<word, false>(x[a][bb], x[a][bb + 4], x[a + 4][bb], x[a + 4][bb + 4], j1);
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      word c2[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) c2[a][bb] = winograd_child<word, false>(c1[a][bb], c1[a][bb + 2], c1[a + 2][bb], c1[a + 2][bb + 2], j2);
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) {
        const word v = winograd_child<word, false>(c2[0][0], c2[0][1], c2[1][0], c2[1][1], j3);
        uint32_t w0 = (uint32_t)v, w1 = (uint32_t)(v >> 32);
        if (ROT) { w0 = __builtin_amdgcn_alignbyte(w0, w0, rot); w1 = __builtin_amdgcn_alignbyte(w1, w1, rot); }
        uint32_t *oo = o + (int64_t)(49 * j1 + 7 * j2 + j3) * a4_bs;
        if (NT) { __builtin_nontemporal_store(w0, oo); __builtin_nontemporal_store(w1, oo + crows); }
        else { oo[0] = w0; oo[crows] = w1; }
      }
    }
  }
}

// The same pass with the seven top-level products of a word meeting in LDS instead of in HBM: a workgroup owns 32 word positions,
// half-wave g (of 8; the last idles through the fold) folds the 343 products of top-level product g for them, scatters its 8 x 8
// blocks into the 16 x 16 grid held in LDS (64 KiB, ds_xor) and, after ONE barrier, the workgroup writes the grid out -- every word of
// C written once (or read-modify-written once when accumulating), no clear, no atomics beyond the CU.  By the counters the atomic form
// writes 3.5 x the result (every atomic reaches HBM) on top of the clear.
constexpr int U4_POS = 32;

template <bool ACC, bool NT>
__global__ __launch_bounds__(AUX_THREADS) void winograd_up4_lds_kernel(
    const word *__restrict__ prod, int64_t p_bs,  // 2401 products per ancestor, stride == cw
    word *anc, int64_t o_stride, int64_t o_bs,    // ancestor array
    int64_t crows, int64_t cw) {                  // cw % U4_POS == 0: the 32 positions of a workgroup lie in one row
  __shared__ unsigned long long grid[256 * U4_POS];
  const int tid = threadIdx.x, pp = tid & (U4_POS - 1), j0 = tid >> 5;
  for (int k = tid; k < 256 * U4_POS; k += AUX_THREADS) grid[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * U4_POS + pp, pi = blockIdx.y;
  const int64_t r = i / cw, w = i - r * cw;
  if (j0 < 7) {
    const word *q = prod + (pi * 2401 + 343 * j0) * p_bs + r * cw + w;
    word out[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) out[a][b] = 0;
    word buf[2][7];
    auto load_group = [&](int g, word (&dst)[7]) {
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) dst[j3] = NT ? __builtin_nontemporal_load(&q[(int64_t)(7 * g + j3) * p_bs]) : q[(int64_t)(7 * g + j3) * p_bs];
    };
    load_group(0, buf[0]);
#pragma unroll
    for (int j1 = 0; j1 < 7; ++j1) {
      word c1[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) c1[a][b] = 0;
#pragma unroll
      for (int j2 = 0; j2 < 7; ++j2) {
        const int g = 7 * j1 + j2;
        if (g + 1 < 49) load_group(g + 1, buf[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        word c2[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
        for (int j3 = 0; j3 < 7; ++j3) winograd_scatter<word>(buf[g & 1][j3], j3, c2[0][0], c2[0][1], c2[1][0], c2[1][1]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) winograd_scatter<word>(c2[a][b], j2, c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) winograd_scatter<word>(c1[a][b], j1, out[a][b], out[a][b + 4], out[a + 4][b], out[a + 4][b + 4]);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(out[a][b]));  // the folds stay above the predicated stores (see winograd_up4_kernel)
    // top level, per lane (the two half-waves of a wave fold different products): quadrant (qa, qb) of the grid gets the block when
    // the product belongs to it (winograd_scatter's table)
    const bool t11 = j0 <= 1, t12 = j0 == 0 || j0 == 2 || j0 == 4 || j0 == 5, t21 = j0 == 0 || j0 == 3 || j0 == 5 || j0 == 6,
               t22 = j0 == 0 || j0 == 4 || j0 == 5 || j0 == 6;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long v = out[a][b];
        if (t11) atomicXor(&grid[((a) * 16 + b) * U4_POS + pp], v);
        if (t12) atomicXor(&grid[((a) * 16 + b + 8) * U4_POS + pp], v);
        if (t21) atomicXor(&grid[((a + 8) * 16 + b) * U4_POS + pp], v);
        if (t22) atomicXor(&grid[((a + 8) * 16 + b + 8) * U4_POS + pp], v);
      }
  }
  __syncthreads();
  // write-out: grid block k * 8 + (tid >> 5) for k = 0 .. 31, the 32 lanes of a half-wave = the 32 positions = 256 contiguous bytes
  word *o = anc + pi * o_bs + r * o_stride + w;
#pragma unroll 4
  for (int k = 0; k < 32; ++k) {
    const int blk = k * 8 + (tid >> 5), ga = blk >> 4, gb = blk & 15;
    word *oo      = o + (int64_t)ga * crows * o_stride + (int64_t)gb * cw;
    const word v  = grid[blk * U4_POS + pp];
    *oo           = ACC ? (*oo ^ v) : v;
  }
}

template <int ROT, bool NT>
__global__ __launch_bounds__(DP3_THREADS) void winograd_down4_pack_kernel(
    const word *__restrict__ anc, int64_t p_stride, int64_t p_bs,
    uint32_t *__restrict__ a4, int64_t a4_bs,                       // packed descendants, a4_bs dwords each
    int64_t crows, int64_t cw, int64_t tiles_r, int64_t tiles_w) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[2][7 * DP3_ROWS * DP3_PITCH];  // two sets of seven 4 KiB tiles
  int64_t tb_;
  int j0;
  pass4_block(blockIdx.x, tb_, j0);
  if (tb_ >= tiles_r * tiles_w) return;   // (whole workgroups leave: no barrier is missed)
  const int64_t wt = tb_ % tiles_w, rt = tb_ / tiles_w, pi = blockIdx.y;
  const int t = threadIdx.x, r = t >> 4, v = t & 15;
  const word *p = anc + pi * p_bs + (rt * DP3_ROWS + r) * p_stride + (wt * DP3_W + v);
  word x[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) x[a][b] = load_top_child<false>(p + (int64_t)a * crows * p_stride + (int64_t)b * cw, 8 * crows * p_stride, 8 * cw, j0);
  const int q = t >> 4, r2 = (t & 15) * 2;
  const int64_t orow = rt * DP3_ROWS + r2;
  const uint32_t rot = ROT == 1 ? (uint32_t)((orow >> 6) & 3) : ROT == 2 ? (uint32_t)((orow >> 7) & 3) : 0u;
  const int64_t oq   = (wt * (DP3_W * 2) + q) ^ (ROT == 2 ? ((orow >> 5) & 1) : 0);
  uint32_t *o = a4 + (pi * 2401 + 343 * j0) * a4_bs + oq * crows + orow;
#pragma unroll
  for (int j1 = 0; j1 < 7; ++j1) {
    word c1[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) c1[a][b] = winograd_child<word, false>(x[a][b], x[a][b + 4], x[a + 4][b], x[a + 4][b + 4], j1);
#pragma unroll
    for (int j2 = 0; j2 < 7; ++j2) {
      word c2[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c2[a][b] = winograd_child<word, false>(c1[a][b], c1[a][b + 2], c1[a + 2][b], c1[a + 2][b + 2], j2);
      // the seven outputs of (j1, j2) go through seven tiles of one of two sets and ONE barrier: the other set is only rewritten after the
      // next barrier, when every thread has read this one (49 barriers per workgroup instead of 343: with one 512-thread workgroup per
      // CU a barrier per 4 KiB of output was a visible part of the pass)
      uint32_t *set = tile[(7 * j1 + j2) & 1];
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3)
        *reinterpret_cast<word *>(set + j3 * (DP3_ROWS * DP3_PITCH) + r * DP3_PITCH + 2 * (v ^ ((r >> 1) & 15))) =
            winograd_child<word, false>(c2[0][0], c2[0][1], c2[1][0], c2[1][1], j3);
      __syncthreads();
      const int rs = 2 * ((q >> 1) ^ ((r2 >> 1) & 15)) + (q & 1);  // rows r2 and r2 + 1 share the swizzle
#pragma unroll
      for (int j3 = 0; j3 < 7; ++j3) {
        const int k        = 49 * j1 + 7 * j2 + j3;
        const uint32_t *tb = set + j3 * (DP3_ROWS * DP3_PITCH);
        uint32_t w0 = tb[(r2 + 0) * DP3_PITCH + rs], w1 = tb[(r2 + 1) * DP3_PITCH + rs];
        if (ROT) { w0 = __builtin_amdgcn_alignbyte(w0, w0, rot); w1 = __builtin_amdgcn_alignbyte(w1, w1, rot); }
        if (NT) __builtin_nontemporal_store((unsigned long long)w0 | ((unsigned long long)w1 << 32), reinterpret_cast<unsigned long long *>(o + (int64_t)k * a4_bs));
        else *reinterpret_cast<uint2 *>(o + (int64_t)k * a4_bs) = make_uint2(w0, w1);
      }
    }
  }
}
}  // namespace

// Descendant 343 * j0 + 49 * j1 + 7 * j2 + j3 of ancestor i is stored at index 2401 * i + that; descendants are crows x cw words,
// contiguous; an ancestor is 16 * crows rows x 16 * cw words with row stride p_stride.
extern "C" hipError_t gf2_launch_winograd_down4(hipStream_t s, int bside, const word *anc, int64_t p_stride, int64_t p_bs,
                                                word *gchild, int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t c_bs = crows * cw;
  if (nparents * c_bs == 0) return hipSuccess;
  // the form that reads the ancestor once through LDS needs the 32 positions of a workgroup inside one descendant row: 0.92 ms against
  // 1.26 ... 1.38 ms for the B side of 65536^3 (M4RI_AMD_DOWN4=direct selects the other form, for measurements)
  static const bool want_lds = !(getenv("M4RI_AMD_DOWN4") && getenv("M4RI_AMD_DOWN4")[0] == 'd');
  if (want_lds && cw % 32 == 0 && c_bs / 32 <= 0x7fffffffLL && nparents <= 65535) {
    const dim3 g((unsigned)(c_bs / 32), (unsigned)nparents);
#define D4L_LAUNCH(BS, NT) hipLaunchKernelGGL((winograd_down4_lds_kernel<BS, NT>), g, dim3(AUX_THREADS), 0, s, anc, p_stride, p_bs, gchild, c_bs, crows, cw)
    if (pass_nt() & 1) { if (bside) D4L_LAUNCH(true, true); else D4L_LAUNCH(false, true); }
    else { if (bside) D4L_LAUNCH(true, false); else D4L_LAUNCH(false, false); }
#undef D4L_LAUNCH
    return hipGetLastError();
  }
  const int64_t nposblocks = (c_bs + AUX_THREADS - 1) / AUX_THREADS, grid = pass4_grid(nposblocks);
  if (grid > 0x7fffffffLL || nparents > 65535) return hipErrorInvalidValue;
#define D4_LAUNCH(BS, NT)                                                                                                             \
  hipLaunchKernelGGL((winograd_down4_kernel<BS, NT>), dim3((unsigned)grid, (unsigned)nparents), dim3(AUX_THREADS), 0, s, anc, p_stride, p_bs, gchild, \
                     c_bs, crows, cw, nposblocks)
  if (pass_nt() & 1) { if (bside) D4_LAUNCH(true, true); else D4_LAUNCH(false, true); }
  else { if (bside) D4_LAUNCH(true, false); else D4_LAUNCH(false, false); }
#undef D4_LAUNCH
  return hipGetLastError();
}

// anc (+)= the recombination of the 2401 products per ancestor.  acc == 0: the ancestor's 16 crows x 16 cw words are zeroed first (the
// seven top-level products of a word meet in it by atomic XOR); acc != 0: they are added onto what is there.
extern "C" hipError_t gf2_launch_winograd_up4(hipStream_t s, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs,
                                              int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t p_bs = crows * cw;
  if (nparents * p_bs == 0) return hipSuccess;
  // the form that combines the seven top-level products in LDS needs the 32 positions of a workgroup inside one row
  // (M4RI_AMD_UP4=atomic selects the other form: products meeting in HBM by atomic XOR, for measurements)
  static const bool want_lds = !(getenv("M4RI_AMD_UP4") && getenv("M4RI_AMD_UP4")[0] == 'a');
  if (want_lds && cw % U4_POS == 0 && p_bs / U4_POS <= 0x7fffffffLL && nparents <= 65535) {
    const dim3 g((unsigned)(p_bs / U4_POS), (unsigned)nparents);
#define U4L_LAUNCH(AC, NT) hipLaunchKernelGGL((winograd_up4_lds_kernel<AC, NT>), g, dim3(AUX_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows, cw)
    if (pass_nt() & 2) { if (acc) U4L_LAUNCH(true, true); else U4L_LAUNCH(false, true); }
    else { if (acc) U4L_LAUNCH(true, false); else U4L_LAUNCH(false, false); }
#undef U4L_LAUNCH
    return hipGetLastError();
  }
  const int64_t nposblocks = (p_bs + AUX_THREADS - 1) / AUX_THREADS, grid = pass4_grid(nposblocks);
  if (grid > 0x7fffffffLL || nparents > 65535 || (crows * o_stride + cw) * 8 >= (1ll << 31)) return hipErrorInvalidValue;
  if (!acc)
    for (int64_t i = 0; i < nparents; ++i) {
      const hipError_t e = gf2_launch_rowwise(s, 2, anc + i * o_bs, o_stride, nullptr, 0, nullptr, 0, 16 * crows, 16 * cw);
      if (e != hipSuccess) return e;
    }
  if (pass_nt() & 2)
    hipLaunchKernelGGL((winograd_up4_kernel<true>), dim3((unsigned)grid, (unsigned)nparents), dim3(AUX_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows,
                       cw, nposblocks);
  else
    hipLaunchKernelGGL((winograd_up4_kernel<false>), dim3((unsigned)grid, (unsigned)nparents), dim3(AUX_THREADS), 0, s, prod, p_bs, anc, o_stride, o_bs, crows,
                       cw, nposblocks);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_winograd_down4_pack(hipStream_t s, const word *anc, int64_t p_stride, int64_t p_bs, word *a4,
                                                     int64_t nparents, int64_t crows, int64_t cw, int rot) {
  if (nparents * crows * cw == 0) return hipSuccess;
  if (!gf2_winograd_down3_pack_ok(a4, crows, cw)) return hipErrorInvalidValue;
  // the form without the transpose (one word column x 32 rows per workgroup, the grid through LDS) wherever the shape allows it:
  // 1.13 against 1.33 ms at 65536^3; M4RI_AMD_DOWN4_PACK=transpose keeps the other
  static const bool want_lds = !(getenv("M4RI_AMD_DOWN4_PACK") && getenv("M4RI_AMD_DOWN4_PACK")[0] == 't');
  if (want_lds && rot != 2 && crows % 32 == 0 && cw % 16 == 0 && nparents <= 65535) {
    const int64_t groups = ((crows / 32 + 7) / 8) * (cw / 16);
    if (groups * 128 <= 0x7fffffffLL) {
      const dim3 g((unsigned)(groups * 128), (unsigned)nparents);
#define DP4L_LAUNCH(R, NT) hipLaunchKernelGGL((winograd_down4_pack_lds_kernel<R, NT>), g, dim3(AUX_THREADS), 0, s, anc, p_stride, p_bs, reinterpret_cast<uint32_t *>(a4), crows * cw * 2, crows, cw)
      if (pass_nt() & 4) { if (rot == 1) DP4L_LAUNCH(1, true); else DP4L_LAUNCH(0, true); }
      else { if (rot == 1) DP4L_LAUNCH(1, false); else DP4L_LAUNCH(0, false); }
#undef DP4L_LAUNCH
      return hipGetLastError();
    }
  }
  const int64_t tiles_r = crows / DP3_ROWS, tiles_w = cw / DP3_W;
  const int64_t grid = pass4_grid(tiles_r * tiles_w);
  if (grid > 0x7fffffffLL || nparents > 65535) return hipErrorInvalidValue;
#define DP4_LAUNCH(R, NT)                                                                                                                     \
  hipLaunchKernelGGL((winograd_down4_pack_kernel<R, NT>), dim3((unsigned)grid, (unsigned)nparents), dim3(DP3_THREADS), 0, s, anc, p_stride, p_bs, \
                     reinterpret_cast<uint32_t *>(a4), crows * cw * 2, crows, cw, tiles_r, tiles_w)
  if (pass_nt() & 4) { if (rot == 2) DP4_LAUNCH(2, true); else if (rot == 1) DP4_LAUNCH(1, true); else DP4_LAUNCH(0, true); }
  else { if (rot == 2) DP4_LAUNCH(2, false); else if (rot == 1) DP4_LAUNCH(1, false); else DP4_LAUNCH(0, false); }
#undef DP4_LAUNCH
  return hipGetLastError();
}

// Two levels per pass.  Grandchildren are crows x cw words, contiguous; a grandparent is 4*crows rows
// x 4*cw words with row stride p_stride; grandchild 7*j1 + j2 of grandparent i is stored at index
// 49*i + 7*j1 + j2.
extern "C" hipError_t gf2_launch_winograd_down2(hipStream_t s, int bside, const word *gparent, int64_t p_stride,
                                                int64_t p_bs, word *gchild, int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t c_bs = crows * cw;
  if (nparents * c_bs == 0) return hipSuccess;
  if (vec_ok(gparent, p_stride, p_bs, cw, cw) && vec_ok(gchild, cw, c_bs, cw, cw)) {
    const int64_t total = nparents * crows * (cw / 2);
    if (bside)
      hipLaunchKernelGGL((winograd_down2_kernel<word2, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(gparent), p_stride / 2, p_bs / 2, reinterpret_cast<word2 *>(gchild),
                         c_bs / 2, nparents, crows, cw / 2);
    else
      hipLaunchKernelGGL((winograd_down2_kernel<word2, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(gparent), p_stride / 2, p_bs / 2, reinterpret_cast<word2 *>(gchild),
                         c_bs / 2, nparents, crows, cw / 2);
  } else {
    const int64_t total = nparents * crows * cw;
    if (bside)
      hipLaunchKernelGGL((winograd_down2_kernel<word, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s, gparent,
                         p_stride, p_bs, gchild, c_bs, nparents, crows, cw);
    else
      hipLaunchKernelGGL((winograd_down2_kernel<word, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s, gparent,
                         p_stride, p_bs, gchild, c_bs, nparents, crows, cw);
  }
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_winograd_up2(hipStream_t s, int acc, const word *prod, word *gparent, int64_t o_stride,
                                              int64_t o_bs, int64_t nparents, int64_t crows, int64_t cw) {
  const int64_t p_bs = crows * cw;
  if (nparents * p_bs == 0) return hipSuccess;
  if (vec_ok(prod, cw, p_bs, cw, cw) && vec_ok(gparent, o_stride, o_bs, cw, cw)) {
    const int64_t total = nparents * crows * (cw / 2);
    if (acc)
      hipLaunchKernelGGL((winograd_up2_kernel<word2, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(prod), p_bs / 2, reinterpret_cast<word2 *>(gparent), o_stride / 2,
                         o_bs / 2, nparents, crows, cw / 2);
    else
      hipLaunchKernelGGL((winograd_up2_kernel<word2, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s,
                         reinterpret_cast<const word2 *>(prod), p_bs / 2, reinterpret_cast<word2 *>(gparent), o_stride / 2,
                         o_bs / 2, nparents, crows, cw / 2);
  } else {
    const int64_t total = nparents * crows * cw;
    if (acc)
      hipLaunchKernelGGL((winograd_up2_kernel<word, true>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s, prod, p_bs,
                         gparent, o_stride, o_bs, nparents, crows, cw);
    else
      hipLaunchKernelGGL((winograd_up2_kernel<word, false>), dim3(grid_for(total)), dim3(AUX_THREADS), 0, s, prod, p_bs,
                         gparent, o_stride, o_bs, nparents, crows, cw);
  }
  return hipGetLastError();
}
