// m4rm8_leaf.hip -- M4RM leaf, generation 3: 8-bit tables with 128-byte entries, two tables
// interleaved per LDS bank row, double-buffered, every wave symmetric.
//
// Generation 2 (k = 7, 256-byte entries; retired in round 2) was bound by LDS-array cycles: per workgroup and stage 512 gathers
// (4 clk) + 64 table writes (8 clk) = 2560 clk for 14 inner bits of a 1024 x 2048 tile.  The same
// 2560 clk buy 16 inner bits of a tile of the same area if the tile is 2048 rows x 1024 columns:
// twice the rows share every table entry, entries are 128 B, and two 256-entry tables (k = 8) of a
// stage are only 64 KiB, so double-buffering them fits in 128 KiB.  What made 128-byte entries
// unusable before -- two different rows of one ds_read_b128 service group landing in the same half
// of the 256-byte bank row -- is removed by construction:
//
//   * LDS bank row x holds [T0[x] | T1[x]]: the first table of the stage lives in the left 128
//     bytes of every bank row, the second in the right 128 bytes;
//   * lane = (row group lane>>3, 16-byte slot lane&7).  In the FIRST gather of a row pair of
//     instructions the row groups with (lane>>4)&1 == 0 read T0 and the others T1; the SECOND gather
//     swaps.  Each of ds_read_b128's four 16-lane service groups ({0-3,12-15,20-27}, ...) then holds
//     two row groups in the left half with slots {0-3} and {4-7} and two in the right half with
//     {4-7} and {0-3}: 16 distinct slots, conflict-free for ANY indices.
//   * k = 8 again: indices are the bytes of A, no bit repacking; A is only transposed to chunk-major
//     dwords (a4_pack_kernel) so that the four rows of a read group are one 16-byte load.
//
// Everything else is generation 2's: C-stationary tile in VGPRs (128 dwords per lane), one
// v_perm_b32 per lookup address (the selector is a per-lane constant now), one v_bitop3_b32 per dword
// folds two lookups, Gray-code table build (8 entries per thread and stage, one per row group),
// operands through range-checked buffer descriptors, one barrier per stage.
//
// Replaces (result-identical) _mzd_mul_m4rm, mzd_make_table and _mzd_combine_N of the reference
// (/root/reference m4ri/brilliantrussian.c:1032-1190, :163-211, m4ri/xor_template.h:12-227).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gf2_common.h"

namespace {

constexpr int K8_BITS  = 8;             // bits per table index
constexpr int K8_STAGE = 2 * K8_BITS;   // inner bits per stage (two tables)
constexpr int K8_CHUNK = 2 * K8_STAGE;  // inner bits per A dword (two stages)
constexpr int K8_TW    = 16;            // tile width in words (1024 columns, 128 B per entry)

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Raw buffer descriptor from wave-uniform inputs (readfirstlane makes the uniformity provable to
// hipcc; otherwise it may wrap every buffer_load in a waterfall loop, cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
  const uint64_t b  = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane(bytes);
  void *p           = reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)nb, 0x00020000);
}

// v_perm_b32(a, coloff, sel): byte j of a -> bits 8..15 (table index), coloff.byte0 -> bits 0..7
// (table half + column slot), buffer -> bit 16 (taken from coloff.byte1 == 0x01)
__device__ __forceinline__ uint32_t perm_sel(int j, int buf) {
  return 0x0c000000u | ((buf ? 0x01u : 0x0cu) << 16) | ((uint32_t)(4 + j) << 8) | 0x00u;
}

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
    // generation 4 reads table (rot + i) & 3 in a row's i-th gather, rot = (row >> 6) & 3 (its lane
    // geometry): store the four index bytes pre-rotated so that byte i IS the i-th gather's index
    // and the kernel's v_perm selectors are compile-time constants.  Generation 5 (rot == 2) does the
    // same with rot = (row >> 7) & 3 and additionally swaps the two chunks of a word for rows with
    // (row >> 5) & 1 (m4rm8o_leaf.hip)
    int64_t qo = q;
    if (rot == 1) v = __builtin_amdgcn_alignbyte(v, v, (uint32_t)(((r0 + r) >> 6) & 3));
    if (rot == 2) {
      v = __builtin_amdgcn_alignbyte(v, v, (uint32_t)(((r0 + r) >> 7) & 3));
      qo = q ^ (((r0 + r) >> 5) & 1);
    }
    A4[b * a4_bs + qo * m_pad + r0 + r] = v;  // rows m .. m_pad-1 come out 0 (index 0 = zero entries)
  }
}

template <int RG, int UG, bool PIPE, bool XOR_OUT>
__global__ __launch_bounds__(LEAF_THREADS) void m4rm8_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 65536];  // [buffer][256 bank rows][T0 | T1][128 B]
  constexpr int R  = 64 * RG;  // tile rows: 64 row groups (8 waves x 8) x RG rows
  constexpr int NG = RG / UG;  // row groups per stage (>= 8: a thread writes one table entry with each of the first 8)
  static_assert(RG % UG == 0 && NG >= 8, "need at least 8 row groups per stage");

  const int tid  = threadIdx.x;
  const int c    = tid & 7;          // 16-byte column slot of the 128-byte table entry
  const int rgrp = tid >> 3;         // row group 0..63
  const int par  = (tid >> 4) & 1;   // which table this lane reads FIRST in a pair of gathers
  // build role: 16 consecutive lanes = 8 slots x 2 tables of ONE entry index = one whole 256-byte
  // bank row per ds_write_b128 service group (conflict-free; see m4rm8q_leaf.hip)
  const int bz   = (tid >> 3) & 1;   // table 0/1 of the stage
  const int bhi  = tid >> 4;         // bits 3..7 of the entries this thread writes

  // block -> (batch, tile_n, ksplit, tile_m); consecutive logical ids share a B panel, and the XCD
  // remap keeps them on one XCD's L2 (blocks are dispatched round-robin over 8 XCDs)
  uint32_t lid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x;
    if ((nwg & 7u) == 0u) lid = (lid & 7u) * (nwg >> 3) + (lid >> 3);
  }
  const int tile_m = lid % p.tiles_m; lid /= p.tiles_m;
  const int ks     = lid % p.ksplit;  lid /= p.ksplit;
  const int tile_n = lid % p.tiles_n; lid /= p.tiles_n;
  const int64_t bat = lid;

  const uint32_t *Apkb = p.Apk + bat * p.apk_bs;
  const word *Bb      = p.B + bat * p.b_bs;
  word *__restrict__ Cb = p.C + bat * p.c_bs;

  const int nq = 2 * ((p.l + 63) / 64);
  // The packed A and B are read through raw buffer descriptors: per-lane 32-bit offsets from a wave-uniform
  // base, and the hardware range check returns 0 for rows >= m of the packed A and rows >= l of B -- exactly
  // the zero padding the algorithm wants, so the main loop has no edge branches.
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(Apkb, (uint32_t)((int64_t)nq * p.apk_stride * 4));  // apk_stride = m_pad
  const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(Bb, (uint32_t)(((int64_t)(p.l - 1) * p.b_stride + p.wn) * 8));

  const int w0   = tile_n * K8_TW + c * 2;  // this lane's two words of the row
  const bool v0  = w0 < p.wn;
  const bool v1  = (w0 + 1) < p.wn;
  const int row0 = tile_m * R + rgrp * RG;
  const uint32_t a_qs   = (uint32_t)p.apk_stride * 4u;  // bytes between chunks of the packed A (m_pad rows)
  const uint32_t b_rs   = (uint32_t)p.b_stride * 8u;
  const uint32_t a_lane = (uint32_t)row0 * 4u;
  // B offsets = wave-uniform part (table, tile column: SGPRs) + the lane's 16-byte slot; keeping the
  // uniform part out of VGPRs avoids spilled offsets (a scratch reload costs a vmcnt(0) drain)
  const uint32_t b_uni  = (uint32_t)__builtin_amdgcn_readfirstlane(tile_n) * (K8_TW * 8u);
  const uint32_t b_slot = (uint32_t)bz * K8_BITS * b_rs + (uint32_t)c * 16u;
  // per-lane perm operands: byte0 = table half (0 / 128) + column slot, byte1 = 0x01 (buffer bit)
  const uint32_t coloff1 = (uint32_t)(par * 128 + c * 16) | 0x0100u;        // first gather: table `par`
  const uint32_t coloff2 = (uint32_t)((par ^ 1) * 128 + c * 16) | 0x0100u;  // second gather: the other one
  unsigned char *const wr_base = lds + bhi * 8 * 256 + bz * 128 + c * 16;

  // perm selectors: stage J of a chunk uses index bytes 2J (table 0) and 2J+1 (table 1) and buffer J
  const uint32_t sel1_j0 = perm_sel(0 + par, 0), sel2_j0 = perm_sel(1 - par, 0);
  const uint32_t sel1_j1 = perm_sel(2 + par, 1), sel2_j1 = perm_sel(3 - par, 1);

  uint32_t acc[RG][4];
#pragma unroll
  for (int t = 0; t < RG; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0u; }

  const int q_begin = ks * p.chunks_per_split;
  int q_end         = q_begin + p.chunks_per_split;
  if (q_end > nq) q_end = nq;

  // B rows of the table this thread helps to build: rows 3..6 of the 7 (-> base) and rows 0..2
  // (-> Gray chain).  Columns outside the matrix may hold a neighbour's bits when B is a window;
  // they only reach C columns that are never stored.
  uint4 bhi_rows[5], blo_rows[3];
  auto load_hi = [&](int stage) {
    uint32_t off = (b_uni + ((uint32_t)stage * K8_STAGE + 3u) * b_rs) + b_slot;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      bhi_rows[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
      off += b_rs;
      asm volatile("" : "+v"(off));  // one running offset VGPR instead of hoisted per-row offsets
    }
  };
  auto load_lo = [&](int stage) {
    uint32_t off = (b_uni + (uint32_t)stage * K8_STAGE * b_rs) + b_slot;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      blo_rows[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
      off += b_rs;
      asm volatile("" : "+v"(off));
    }
  };
  uint32_t cur[4];
  auto make_base = [&]() {
    // the rows become visible to the optimiser only here (volatile asm stays behind the previous
    // barrier); un-pinned, hipcc hoists these XORs up to the loads and waits out their latency
#pragma unroll
    for (int j = 0; j < 5; ++j)
      asm volatile("" : "+v"(bhi_rows[j].x), "+v"(bhi_rows[j].y), "+v"(bhi_rows[j].z), "+v"(bhi_rows[j].w));
    cur[0] = cur[1] = cur[2] = cur[3] = 0u;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bool on = (bhi >> j) & 1;
      cur[0] ^= on ? bhi_rows[j].x : 0u;
      cur[1] ^= on ? bhi_rows[j].y : 0u;
      cur[2] ^= on ? bhi_rows[j].z : 0u;
      cur[3] ^= on ? bhi_rows[j].w : 0u;
    }
  };
  // entry number i (0..7) of the thread's 8: Gray step + one ds_write_b128 into buffer `buf`
  auto put_entry = [&](int i, int buf) {
    if (i > 0) {
      const int j = __builtin_ctz(i);
      if (i == 1) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
          asm volatile("" : "+v"(blo_rows[jj].x), "+v"(blo_rows[jj].y), "+v"(blo_rows[jj].z), "+v"(blo_rows[jj].w));
      }
      cur[0] ^= blo_rows[j].x;
      cur[1] ^= blo_rows[j].y;
      cur[2] ^= blo_rows[j].z;
      cur[3] ^= blo_rows[j].w;
    }
    // keep the Gray chain a chain (one XOR + one ds_write_b128 per entry)
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    const int gcode = i ^ (i >> 1);
    *reinterpret_cast<uint4 *>(wr_base + buf * 65536 + gcode * 256) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
  };

  static_assert(UG == 4, "the A refill is one 16-byte load per read group of 4 rows");
  uint32_t areg[RG];
  auto load_a4 = [&](int g, int q) {
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  a_rsrc, (int)(a_lane + (uint32_t)q * a_qs + (uint32_t)g * 16u), 0, 0));
    areg[g * 4 + 0] = v.x; areg[g * 4 + 1] = v.y; areg[g * 4 + 2] = v.z; areg[g * 4 + 3] = v.w;
  };
#pragma unroll
  for (int g = 0; g < RG / 4; ++g) load_a4(g, q_begin);

  // one stage: gather from the tables of stage s = 2q+J (buffer J) while building those of stage
  // s+1 into buffer J^1; in the second stage of a chunk the A registers are refilled on the way
  auto stage = [&](auto jtag, int q) {
    constexpr int J = decltype(jtag)::value;
    const int s     = 2 * q + J;
    // on entry: cur = base of the tables of stage s+1 (made late in the previous stage), blo_rows =
    // their chain rows, bhi_rows = the base rows of stage s+2.  Nothing but gathers happens right
    // behind the barrier: all 8 waves come out of it together, and whatever non-LDS work sits here
    // (VMEM issue, base XORs) would idle the LDS pipe for every one of them at once.
    uint4 t0[PIPE ? 2 : 1][UG], t1[PIPE ? 2 : 1][UG];
    auto issue = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const uint32_t a0 = __builtin_amdgcn_perm(areg[g * UG + u], coloff1, J ? sel1_j1 : sel1_j0);
        const uint32_t a1 = __builtin_amdgcn_perm(areg[g * UG + u], coloff2, J ? sel2_j1 : sel2_j0);
        t0[slot][u]       = *reinterpret_cast<const uint4 *>(lds + a0);
        t1[slot][u]       = *reinterpret_cast<const uint4 *>(lds + a1);
      }
      // second stage of a chunk: these four rows' last indices are out, refill their A registers
      // with the next chunk right away (one 16-byte load)
      if constexpr (J == 1) load_a4(g, q + 1);
    };
    auto fold = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        uint32_t *a = acc[g * UG + u];
        a[0] = xor3(a[0], t0[slot][u].x, t1[slot][u].x);
        a[1] = xor3(a[1], t0[slot][u].y, t1[slot][u].y);
        a[2] = xor3(a[2], t0[slot][u].z, t1[slot][u].z);
        a[3] = xor3(a[3], t0[slot][u].w, t1[slot][u].w);
        // pin the accumulation here (XOR is associative: un-pinned, hipcc re-associates the whole
        // stage into one late XOR tree and keeps every loaded table row live)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      }
    };
    if constexpr (PIPE) { issue(0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // software pipeline: group g+1's gathers go out before group g is folded, so the LDS queue
      // never drains while this wave is busy with XORs, table writes and VMEM issue
      if constexpr (PIPE) { if (g + 1 < NG) issue(g + 1, (g + 1) & 1); }
      else issue(g, 0);
      // the 8 table entries go out with the FIRST 8 groups, so the chain rows are dead early and
      // their successors (first needed one group into the next stage) get most of a stage to arrive
      if (g < 8) put_entry(g, J ^ 1);
      if (g == 8 || (NG == 8 && g == 7)) load_lo(s + 2);
      if (g == (NG > 10 ? 10 : NG - 1)) {
        make_base();     // base of stage s+2's tables (their entries are written during stage s+1)
        load_hi(s + 3);  // and the base rows after that: a whole stage of latency budget
      }
      __builtin_amdgcn_sched_barrier(0);
      fold(g, PIPE ? (g & 1) : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (q_begin < q_end) {
    // prologue: tables of the first stage (buffer 0), then the rows for building the second
    load_hi(2 * q_begin);
    load_lo(2 * q_begin);
    make_base();
#pragma unroll
    for (int i = 0; i < 8; ++i) put_entry(i, 0);
    load_hi(2 * q_begin + 1);
    load_lo(2 * q_begin + 1);
    make_base();
    load_hi(2 * q_begin + 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    for (int q = q_begin; q < q_end; ++q) {
      stage(std::integral_constant<int, 0>{}, q);
      stage(std::integral_constant<int, 1>{}, q);
    }
  }

  // epilogue: C tile out.  One running row pointer (pinned, so hipcc cannot hoist RG 64-bit row
  // addresses above the main loop); the column guards are loop-invariant per lane.
  if (v0) {
    word *cp       = Cb + (int64_t)row0 * p.c_stride + w0;
    const int rows = (p.m - row0) < RG ? (p.m - row0) : RG;  // may be <= 0
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      if (t < rows) {
        const word x0 = (word)acc[t][0] | ((word)acc[t][1] << 32);
        const word x1 = (word)acc[t][2] | ((word)acc[t][3] << 32);
        if constexpr (!XOR_OUT) {
          cp[0] = x0;
          if (v1) cp[1] = x1;
        } else {
          // C ^= tile: a no-return L2 atomic needs no destination registers and is what makes
          // inner-dimension splits (ksplit > 1) race-free; XOR is exact, so order is moot
          atomicXor(reinterpret_cast<unsigned long long *>(cp), (unsigned long long)x0);
          if (v1) atomicXor(reinterpret_cast<unsigned long long *>(cp + 1), (unsigned long long)x1);
        }
      }
      cp += p.c_stride;
      asm volatile("" : "+v"(cp));
    }
  }
}

}  // namespace

// words of workspace the packed copy of A needs for a launch (uint32 units rounded to 64-bit words)
extern "C" int64_t gf2_m4rm8_a4_words(int64_t m, int64_t l, int64_t batch) {
  const int64_t nq = 2 * ((l + 63) / 64);  // two 32-bit chunks per word of A (always even)
  return (batch * ((m + 3) & ~(int64_t)3) * nq + 1) / 2;
}

// Host launchers.  gf2_launch_a4_pack fills `a4_ws` (gf2_m4rm8_a4_words words) with the packed copy of
// A; gf2_launch_m4rm8 runs the leaf on it.  rg: tile height / 64 (rows = 64*rg).
static bool k8_geometry(LeafArgs &a, word *a4_ws, int rg, int64_t &nq, int64_t &m_pad) {
  const int R = 64 * rg;
  a.wn        = (int32_t)words_of(a.n);
  a.tiles_m   = (a.m + R - 1) / R;
  a.tiles_n   = (a.wn + K8_TW - 1) / K8_TW;
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
  if (!k8_geometry(a, a4_ws, 32, nq, m_pad)) return hipSuccess;
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

extern "C" hipError_t gf2_launch_m4rm8(hipStream_t stream, LeafArgs a, word *a4_ws, int rg, int ug, int pipe) {
  int64_t nq, m_pad;
  if (!k8_geometry(a, a4_ws, rg, nq, m_pad)) return hipSuccess;
  if ((uint64_t)m_pad * (uint64_t)nq * 4 >= (1ull << 32)) return hipErrorInvalidValue;
  if (a.ksplit < 1) a.ksplit = 1;
  int cps = (int)((nq + a.ksplit - 1) / a.ksplit);
  if (cps < 1) cps = 1;
  a.chunks_per_split = cps;
  a.ksplit           = (int)((nq + cps - 1) / cps);
  if (a.ksplit > 1 && a.mode == 0) return hipErrorInvalidValue;  // caller must pre-zero C and pass mode 1
  const long long nwg = (long long)a.tiles_m * a.tiles_n * a.ksplit * a.batch;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nwg), block(LEAF_THREADS);
#define K8_CASE(RGV, UGV, PV)                                                                        \
  if (rg == RGV && ug == UGV && pipe == PV) {                                                          \
    if (a.mode == 0) hipLaunchKernelGGL((m4rm8_kernel<RGV, UGV, PV != 0, false>), grid, block, 0, stream, a); \
    else             hipLaunchKernelGGL((m4rm8_kernel<RGV, UGV, PV != 0, true>), grid, block, 0, stream, a);  \
    return hipGetLastError();                                                                        \
  }
  K8_CASE(32, 4, 0)
#undef K8_CASE
  return hipErrorInvalidValue;
}
