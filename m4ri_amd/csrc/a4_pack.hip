// a4_pack.hip -- the packed form of A that the M4RM leaf (generation 4, m4rm8q_leaf.hip) reads: the dwords of A transposed
// to CHUNK-major, A4[q][r] with m_pad (multiple of 4) rows per 32-bit chunk q, so that the four consecutive rows a lane
// handles per read group are one 16-byte load, the index bytes pre-rotated the way the leaf's constant selectors want
// them (`rot`).  The fused Winograd passes (aux_kernels.hip) write the same form directly; this kernel serves the products
// whose A does not come out of a pass.  (The file used to hold the generation-3 leaf as well -- 8-bit tables with 128-byte
// entries, 2048 x 1024 tiles -- retired in round 2 when generation 4 turned out to be at least as fast on every shape,
// short tiles included: DESIGN.md 3.1.)
#include <hip/hip_runtime.h>
#include "gf2_common.h"

namespace {

// ---- A -> A4: the dwords of A transposed to CHUNK-major, A4[q][r] with m_pad (multiple of 4) rows per
// 32-bit chunk q, so that the four consecutive rows a lane handles per read group are one 16-byte
// load.  Through LDS: a workgroup reads a 64-row x 32-word tile along the rows of A and writes 64
// chunks x 64 rows along the rows of A4 -- both sides coalesced.
constexpr int PK_ROWS = 64, PK_WORDS = 32;
__global__ __launch_bounds__(256) void a4_pack_kernel(const word *__restrict__ A, int64_t a_stride, int64_t a_bs,
                                                      uint32_t *__restrict__ A4, int64_t m_pad, int64_t a4_bs,
                                                      int64_t m, int64_t l, int64_t row_tiles, int64_t word_tiles, int rot) {
  __shared__ uint32_t tile[PK_ROWS][2 * PK_WORDS + 1];
  const int64_t nq = 2 * ((l + 63) / 64);  // two 32-bit chunks per word of A (always even)
  const int64_t wa = (l + 63) >> 6;
  int64_t bid      = blockIdx.x;
  const int64_t wt = bid % word_tiles; bid /= word_tiles;
  const int64_t rt = bid % row_tiles;  bid /= row_tiles;
  const int64_t b  = bid;
  const int64_t r0 = rt * PK_ROWS, w0 = wt * PK_WORDS;
  for (int i = threadIdx.x; i < PK_ROWS * PK_WORDS; i += 256) {
    const int r = i / PK_WORDS, w = i - r * PK_WORDS;
    word v = 0;
    if (r0 + r < m && w0 + w < wa) v = A[b * a_bs + (r0 + r) * a_stride + (w0 + w)];
    const int64_t bit0 = (w0 + w) * 64;  // bits >= l never reach the tables, but keep them 0 anyway
    if (bit0 + 64 > l) v = (bit0 >= l) ? 0 : (v & ((~(word)0) >> (64 - (l - bit0))));
    tile[r][2 * w]     = (uint32_t)v;
    tile[r][2 * w + 1] = (uint32_t)(v >> 32);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PK_ROWS * 2 * PK_WORDS; i += 256) {
    const int r = i % PK_ROWS, ql = i / PK_ROWS;
    const int64_t q = 2 * w0 + ql;
    if (q >= nq || r0 + r >= m_pad) continue;
    uint32_t v = tile[r][ql];
    // the leaf reads table (rot + i) & 3 in a row's i-th gather, rot = (row >> 6) & 3 (its lane geometry): store the
    // four index bytes pre-rotated so that byte i IS the i-th gather's index and the kernel's v_perm selectors are
    // compile-time constants (rot == 0: plain bytes, kept for tools/leaf_check.cpp)
    if (rot == 1) v = __builtin_amdgcn_alignbyte(v, v, (uint32_t)(((r0 + r) >> 6) & 3));
    A4[b * a4_bs + q * m_pad + r0 + r] = v;  // rows m .. m_pad-1 come out 0 (index 0 = zero entries)
  }
}

}  // namespace

// words of workspace the packed copy of A needs for a launch (uint32 units rounded to 64-bit words)
extern "C" int64_t gf2_m4rm8_a4_words(int64_t m, int64_t l, int64_t batch) {
  const int64_t nq = 2 * ((l + 63) / 64);  // two 32-bit chunks per word of A (always even)
  return (batch * ((m + 3) & ~(int64_t)3) * nq + 1) / 2;
}

// Host launcher: gf2_launch_a4_pack_rot fills `a4_ws` (gf2_m4rm8_a4_words words) with the packed copy of A.
static bool k8_geometry(LeafArgs &a, word *a4_ws, int64_t &nq, int64_t &m_pad) {
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0 || a.l <= 0) return false;
  nq          = 2 * (((int64_t)a.l + 63) / 64);
  m_pad       = ((int64_t)a.m + 3) & ~(int64_t)3;
  a.Apk        = reinterpret_cast<const uint32_t *>(a4_ws);
  a.apk_stride = m_pad;
  a.apk_bs     = m_pad * nq;
  return true;
}

extern "C" hipError_t gf2_launch_a4_pack_rot(hipStream_t stream, LeafArgs a, word *a4_ws, int rot) {
  int64_t nq, m_pad;
  if (!k8_geometry(a, a4_ws, nq, m_pad)) return hipSuccess;
  if ((uint64_t)m_pad * (uint64_t)nq * 4 >= (1ull << 32)) return hipErrorInvalidValue;
  const int64_t wa = ((int64_t)a.l + 63) >> 6;
  const int64_t row_tiles = (m_pad + PK_ROWS - 1) / PK_ROWS, chunk_tiles = (wa + PK_WORDS - 1) / PK_WORDS;
  const int64_t g = row_tiles * chunk_tiles * a.batch;
  if (g > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(a4_pack_kernel, dim3((unsigned)g), dim3(256), 0, stream, a.A, a.a_stride, a.a_bs,
                     reinterpret_cast<uint32_t *>(a4_ws), m_pad, a.apk_bs, (int64_t)a.m, (int64_t)a.l, row_tiles, chunk_tiles, rot);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_a4_pack(hipStream_t stream, LeafArgs a, word *a4_ws) {
  return gf2_launch_a4_pack_rot(stream, a, a4_ws, 0);
}
