// m4rm_small.hip -- M4RM leaf for SMALL products: one launch, no packed A, no slabs.
//
// The generation-4 leaf (m4rm8q_leaf.hip) is built for throughput: 4096 x 512 tiles, A repacked by a pass of its own, short launches
// split along the inner dimension into slabs that a third kernel folds.  A product of a few hundred to a few thousand rows pays for
// that machinery with three dependent launches -- 512^3: 8 + 14 + 6 us of kernels (profiles/r06_small_gpu_path_zero_copy_experiment.log),
// 29 us resident; the depth model's "45 us per short launch".  This kernel is the opposite trade: lighter tables, smaller tiles, everything
// in ONE launch.
//
//   * a workgroup of 256 threads owns 256 rows x 512 columns of C, one ROW PER THREAD, the row's 8 words in registers;
//   * the inner dimension in steps of 64 bits = one word of A per row: SIXTEEN 4-bit tables of the step's 64 rows of B (16 entries of 64 bytes
//     each, 16 KiB) are built straight from registers -- a thread owns one 16-byte slot of four entries of one table and needs that slot of four
//     rows of B, loaded from global memory one step ahead, under the lookups -- into the OTHER of two table sets, and every row adds its sixteen
//     entries: 64 ds_read_b128 per row and step, ONE barrier per step (the first version staged the rows in LDS: two barriers, nothing in flight;
//     2048^3 33.0 -> 30.5 us, 1024 x 1024 x 16384 36.6 -> 30.2, 3 x 100000 x 5000 143 -> 100);
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

// PIPE: more than one step per workgroup -- two table sets (32 KiB), the next step's rows of B in flight under the lookups; a launch of
// single steps (inner dimension of at most 64 bits per split: the rank-64 updates of the solvers) keeps one set, 16 KiB, and its occupancy
template <bool XOR_OUT, bool PIPE>
__global__ __launch_bounds__(SM_THREADS) void m4rm_small_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) word tab[PIPE ? 2 : 1][16 * 16 * SM_TW];  // [step parity][table][entry][slot ^ (entry >> 2)][2 words]
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

  // Build role: table bt, 16-byte slot bs of the entries br, 4 + br, 8 + br, 12 + br (low two index bits br: rows 0 and 1 of the table's four
  // rows of B, the high two bits walk rows 2 and 3).  Sixteen consecutive lanes = the (br, bs) of ONE table: in write number h they hit the
  // sixteen bank positions (br, bs ^ h) -- conflict-free.  The thread needs slot bs of four rows of B: two words each, straight from global
  // memory into registers one step ahead (the sixteen threads that share a row hit the same lines), no staging copy, no second barrier.
  const int bt = tid >> 4, br = (tid >> 2) & 3, bs = tid & 3;
  const bool c0 = 2 * bs < tw, c1 = 2 * bs + 1 < tw;
  const word *bcol = B + w0 + 2 * bs;
  word rr[4][2];
  auto load_rows = [&](int q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t r = (int64_t)q * 64 + 4 * bt + j;
      const bool in   = r < p.l;
      rr[j][0] = (in && c0) ? bcol[r * p.b_stride] : (word)0;
      rr[j][1] = (in && c1) ? bcol[r * p.b_stride + 1] : (word)0;
    }
  };
  auto build = [&](int buf) {
    word e0 = (br & 1) ? rr[0][0] : 0, e1 = (br & 1) ? rr[0][1] : 0;
    e0 ^= (br & 2) ? rr[1][0] : 0;
    e1 ^= (br & 2) ? rr[1][1] : 0;
    unsigned char *tb = reinterpret_cast<unsigned char *>(tab[buf]) + bt * 1024 + br * 64;
#pragma unroll
    for (int h = 0; h < 4; ++h) {  // entry 4 h + br
      const word x0 = e0 ^ ((h & 1) ? rr[2][0] : 0) ^ ((h & 2) ? rr[3][0] : 0);
      const word x1 = e1 ^ ((h & 1) ? rr[2][1] : 0) ^ ((h & 2) ? rr[3][1] : 0);
      *reinterpret_cast<uint4 *>(tb + h * 256 + ((bs ^ h) << 4)) = make_uint4((uint32_t)x0, (uint32_t)(x0 >> 32), (uint32_t)x1, (uint32_t)(x1 >> 32));
    }
  };
  if (q_begin < q_end) {
    load_rows(q_begin);
    build(0);
  }
  __syncthreads();
  for (int q = q_begin; q < q_end; ++q) {
    const int buf = PIPE ? ((q - q_begin) & 1) : 0;
    const word a  = a_next;
    if (PIPE && q + 1 < q_end) {
      load_rows(q + 1);  // in flight under the lookups
      if (live) a_next = arow[q + 1];
    }
    // the lookups: sixteen entries per row, folded two at a time
    if (live && a) {
      const unsigned char *tb = reinterpret_cast<const unsigned char *>(tab[buf]);
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
    if (!PIPE) break;  // (a single step)
    if (q + 1 < q_end) build(buf ^ 1);
    __syncthreads();  // the next step's tables are complete, and nobody still reads the ones the step after it will overwrite
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
//     rounds of workgroups over the CUs (two per CU) x steps per split x 1.9 us  +  38.5 ps per word of C and split (the atomics)
// -- both constants fitted on one MI355X (M4RI_AMD_SMALL_KS sweep of 2048^3: 1 / 2 / 4 / 8 splits = 65.9 / 39.0 / 28.9 / 31.4 us; the model's
// picks against the sweep: 512^3 8 (best measured 8), 1536^3 6 (4 ... 8 equal), 2560^3 4 (4), 4096 x 4096 x 256 13 (16), 8192 x 8192 x 200
// 12 (16)).  1 = unsplit (plain stores, no zeroing of C).
extern "C" int gf2_m4rm_small_ksplit(int64_t tiles, int64_t wl, int cus, int64_t c_words) {
  if (tiles <= 0 || wl < 2) return 1;
  if (cus < 1) cus = 1;
  int best = 1;
  double best_cost = 1e300;
  const int64_t cap = wl < 256 ? wl : 256;
  for (int64_t ks = 1; ks <= cap; ++ks) {
    const int64_t steps = (wl + ks - 1) / ks;
    if ((wl + steps - 1) / steps != ks) continue;  // the launcher would round this count to a smaller one
    const double rounds = (double)((tiles * ks + 2 * cus - 1) / (2 * cus));  // (two workgroups per CU run side by side at no cost: 8192 x 8192 x 200 with 16 splits 31.4 us, with 8 36.3)
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
  const bool pipe = a.chunks_per_split > 1;
  if (a.mode == 0) {
    if (pipe) hipLaunchKernelGGL((m4rm_small_kernel<false, true>), grid, block, 0, stream, a);
    else      hipLaunchKernelGGL((m4rm_small_kernel<false, false>), grid, block, 0, stream, a);
  } else {
    if (pipe) hipLaunchKernelGGL((m4rm_small_kernel<true, true>), grid, block, 0, stream, a);
    else      hipLaunchKernelGGL((m4rm_small_kernel<true, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}
