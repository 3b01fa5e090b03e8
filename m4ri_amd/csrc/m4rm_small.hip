// m4rm_small.hip -- M4RM leaf for SMALL products: one launch, no packed A, no slabs.
//
// The generation-4 leaf (m4rm8q_leaf.hip) is built for throughput: 4096 x 512 tiles, A repacked by a pass of its own, short launches
// split along the inner dimension into slabs that a third kernel folds.  A product of a few hundred to a few thousand rows pays for
// that machinery with three dependent launches -- 512^3: 8 + 14 + 6 us of kernels (profiles/r06_small_gpu_path_zero_copy_experiment.log),
// 29 us resident; the depth model's "45 us per short launch".  This kernel is the opposite trade: lighter tables, smaller tiles, everything
// in ONE launch.
//
//   * a workgroup of 256 threads owns 256 rows x 512 columns of C, one ROW PER THREAD, the row's 8 words in registers;
//   * the inner dimension in steps of 64 bits = one word of A per row: the 64 rows of B of the step are staged in LDS (4 KiB), SIXTEEN
//     4-bit tables (16 entries of 64 bytes each, 16 KiB) are built from them -- thread (t, e) forms entry e of table t from at most four
//     staged rows -- and every row adds its sixteen entries: 64 ds_read_b128 per row and step;
//   * entries are stored with their four 16-byte slots XOR-swizzled by (entry >> 2): the sixteen entries of a table then occupy sixteen
//     different bank positions for every slot number, so lanes with different indices never collide and lanes with equal indices
//     broadcast -- conflict-free for ANY indices, writes included;
//   * short inner loops put few workgroups on the chip, so the inner dimension may be split over workgroups that combine by no-return
//     atomic XOR (C zeroed by the caller when it is not an accumulation); rows of B beyond the inner dimension are staged as zero, so
//     bits of A beyond its last column never matter; whole words of C are written (callers mask excess columns, as for the other leaves).
//
// 1 / 32 byte of LDS traffic per bit operation against generation 4's 1 / 64: past a few thousand rows generation 4 wins again
// (engine.hip: small_leaf_wanted).
//
// Replaces (result-identical) _mzd_mul_m4rm, mzd_make_table and _mzd_combine_N of the reference
// (/root/reference m4ri/brilliantrussian.c:1032-1190, :163-211, m4ri/xor_template.h:12-227) on small operands.
#include <hip/hip_runtime.h>
#include "gf2_common.h"

namespace {

constexpr int SM_ROWS = 256, SM_TW = 8, SM_THREADS = 256;

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

template <bool XOR_OUT>
__global__ __launch_bounds__(SM_THREADS) void m4rm_small_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) word brows[64 * SM_TW];        // the 64 rows of B of the step, 8 words each
  __shared__ __attribute__((aligned(16))) word tab[16 * 16 * SM_TW];     // [table][entry][slot ^ (entry >> 2)][2 words]
  const int tid = threadIdx.x;
  uint32_t b    = blockIdx.x;
  const int ks     = (int)(b % (uint32_t)p.ksplit);  b /= (uint32_t)p.ksplit;
  const int tile_m = (int)(b % (uint32_t)p.tiles_m); b /= (uint32_t)p.tiles_m;
  const int tile_n = (int)(b % (uint32_t)p.tiles_n); b /= (uint32_t)p.tiles_n;
  const int64_t bat = b;
  const word *__restrict__ A = p.A + bat * p.a_bs;
  const word *__restrict__ B = p.B + bat * p.b_bs;
  word *C                    = p.C + bat * p.c_bs;

  const int wl      = (p.l + 63) >> 6;  // steps of the whole inner dimension
  const int q_begin = ks * p.chunks_per_split;
  int q_end         = q_begin + p.chunks_per_split;
  if (q_end > wl) q_end = wl;
  const int row   = tile_m * SM_ROWS + tid;
  const bool live = row < p.m;
  const int w0    = tile_n * SM_TW;
  const int tw    = (p.wn - w0) < SM_TW ? (p.wn - w0) : SM_TW;  // valid words of this tile's rows

  uint32_t acc[2 * SM_TW];
#pragma unroll
  for (int c = 0; c < 2 * SM_TW; ++c) acc[c] = 0u;
  const word *arow = A + (int64_t)row * p.a_stride;
  word a_next      = (live && q_begin < q_end) ? arow[q_begin] : 0;

  // build role: entry e of table t
  const int bt = tid >> 4, be = tid & 15, bh = be >> 2;
  for (int q = q_begin; q < q_end; ++q) {
    // 1. the step's 64 rows of B: consecutive threads take consecutive words of a row (64-byte runs)
#pragma unroll
    for (int k = tid; k < 64 * SM_TW; k += SM_THREADS) {
      const int j = k / SM_TW, c = k % SM_TW;
      const int64_t br = (int64_t)q * 64 + j;
      brows[k] = (br < p.l && c < tw) ? B[br * p.b_stride + w0 + c] : (word)0;
    }
    const word a = a_next;
    if (live && q + 1 < q_end) a_next = arow[q + 1];
    __syncthreads();  // rows staged; every wave is also past its lookups of the previous step, so the tables may be overwritten
    // 2. the tables
    {
      uint4 v[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const uint32_t keep = ((be >> bb) & 1) ? ~0u : 0u;
        const uint4 *r      = reinterpret_cast<const uint4 *>(&brows[(4 * bt + bb) * SM_TW]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 x = r[s];
          v[s].x ^= x.x & keep; v[s].y ^= x.y & keep; v[s].z ^= x.z & keep; v[s].w ^= x.w & keep;
        }
      }
      uint4 *e = reinterpret_cast<uint4 *>(&tab[(bt * 16 + be) * SM_TW]);
#pragma unroll
      for (int s = 0; s < 4; ++s) e[s ^ bh] = v[s];
    }
    __syncthreads();
    // 3. the lookups: sixteen entries per row, folded two at a time
    if (live && a) {
      const unsigned char *tb = reinterpret_cast<const unsigned char *>(tab);
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        const uint32_t i0 = (uint32_t)(a >> (4 * t)) & 15u, i1 = (uint32_t)(a >> (4 * t + 4)) & 15u;
        // byte address of physical slot 0 ^ h: entry base | (h << 4); slot s then is that ^ (s << 4)
        const uint32_t a0 = ((uint32_t)t * 1024u + i0 * 64u) | ((i0 >> 2) << 4);
        const uint32_t a1 = ((uint32_t)(t + 1) * 1024u + i1 * 64u) | ((i1 >> 2) << 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 x = *reinterpret_cast<const uint4 *>(tb + (a0 ^ (uint32_t)(s << 4)));
          const uint4 y = *reinterpret_cast<const uint4 *>(tb + (a1 ^ (uint32_t)(s << 4)));
          acc[4 * s + 0] = xor3(acc[4 * s + 0], x.x, y.x);
          acc[4 * s + 1] = xor3(acc[4 * s + 1], x.y, y.y);
          acc[4 * s + 2] = xor3(acc[4 * s + 2], x.z, y.z);
          acc[4 * s + 3] = xor3(acc[4 * s + 3], x.w, y.w);
        }
      }
    }
  }
  if (!live) return;
  word *crow = C + (int64_t)row * p.c_stride + w0;
#pragma unroll
  for (int c = 0; c < SM_TW; ++c) {
    if (c < tw) {
      const word x = (word)acc[2 * c] | ((word)acc[2 * c + 1] << 32);
      if constexpr (XOR_OUT) {
        if (p.ksplit == 1) crow[c] ^= x;  // one owner per word: a plain read-modify-write
        else atomicXor(reinterpret_cast<unsigned long long *>(crow + c), (unsigned long long)x);
      } else {
        crow[c] = x;
      }
    }
  }
}

}  // namespace

// Inner-dimension splits of a small product of `tiles` tiles, `wl` 64-bit steps and `c_words` words of C in all: the minimum of
//     rounds of workgroups over the CUs x steps per split x 1.9 us  +  38.5 ps per word of C and split (the atomics)
// -- both constants fitted on one MI355X (M4RI_AMD_SMALL_KS sweep of 2048^3: 1 / 2 / 4 / 8 splits = 65.9 / 39.0 / 28.9 / 31.4 us; the model's
// picks against the sweep: 512^3 8 (best measured 8), 1536^3 6 (4 ... 8 equal), 2560^3 4 (4), 4096 x 4096 x 256 14 (16), 8192 x 8192 x 200
// 8 (8)).  1 = unsplit (plain stores, no zeroing of C).
extern "C" int gf2_m4rm_small_ksplit(int64_t tiles, int64_t wl, int cus, int64_t c_words) {
  if (tiles <= 0 || wl < 2) return 1;
  if (cus < 1) cus = 1;
  int best = 1;
  double best_cost = 1e300;
  const int64_t cap = wl < 256 ? wl : 256;
  for (int64_t ks = 1; ks <= cap; ++ks) {
    const int64_t steps = (wl + ks - 1) / ks;
    if ((wl + steps - 1) / steps != ks) continue;  // the launcher would round this count to a smaller one
    const double rounds = (double)((tiles * ks + cus - 1) / cus);
    const double cost   = rounds * (double)steps * 1.9 + (ks > 1 ? 38.5e-6 * (double)c_words * (double)ks : 0.0);
    if (cost < best_cost * 0.97) { best_cost = cost; best = (int)ks; }
  }
  return best;
}

// C (^)= A * B for a batch of equal small products; a.mode 0 = store (a.ksplit must be 1), 1 = XOR into C (any split; the caller has
// zeroed C when the product is not an accumulation).
extern "C" hipError_t gf2_launch_m4rm_small(hipStream_t stream, LeafArgs a) {
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0 || a.l <= 0) return hipSuccess;
  a.wn      = (int32_t)words_of(a.n);
  a.tiles_m = (a.m + SM_ROWS - 1) / SM_ROWS;
  a.tiles_n = (a.wn + SM_TW - 1) / SM_TW;
  const int wl = (a.l + 63) / 64;
  if (a.ksplit < 1) a.ksplit = 1;
  if (a.ksplit > wl) a.ksplit = wl;
  a.chunks_per_split = (wl + a.ksplit - 1) / a.ksplit;
  a.ksplit           = (wl + a.chunks_per_split - 1) / a.chunks_per_split;
  if (a.ksplit > 1 && a.mode == 0) return hipErrorInvalidValue;
  const long long nwg = (long long)a.tiles_m * a.tiles_n * a.batch * a.ksplit;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nwg), block(SM_THREADS);
  if (a.mode == 0) hipLaunchKernelGGL((m4rm_small_kernel<false>), grid, block, 0, stream, a);
  else             hipLaunchKernelGGL((m4rm_small_kernel<true>), grid, block, 0, stream, a);
  return hipGetLastError();
}
